// gm_api.hip -- implementation of the C ABI declared in include/genmap_amd.h (HIP, gfx950).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include "gm_internal.h"
#include "gm_host.h"
#include "gm_kernels.h"

namespace gm {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}

#define GM_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { gm::set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return (e_ == hipErrorOutOfMemory) ? GM_ERR_OOM : GM_ERR_HIP; } } while (0)

static inline unsigned grid_for(uint64_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// ---- rank block construction -------------------------------------------------------------------------------
// cnt[c * (nb + 1) + q] = letters c in block q; entry nb is zero so that the exclusive scan leaves the total there.
template <int WPP>
__global__ __launch_bounds__(256) void count_blocks_kernel(const uint8_t* __restrict__ bwt, uint64_t n, uint64_t nb, uint32_t* __restrict__ cnt)
{
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nb) return;
    uint32_t c[NLET] = {0, 0, 0, 0, 0};
    if (q < nb) {
        constexpr uint32_t SPB = BlockGeom<WPP>::SPB;
        for (uint32_t t = 0; t < SPB; ++t) {
            const uint64_t i = q * SPB + t;
            if (i < n) { const uint32_t s = bwt[i]; if (s < NLET) c[s]++; }
        }
    }
    for (uint32_t s = 0; s < NLET; ++s) cnt[s * (nb + 1) + q] = c[s];
}

template <int WPP>
__global__ __launch_bounds__(256) void pack_blocks_kernel(const uint8_t* __restrict__ bwt, uint64_t n, uint64_t nb, const uint32_t* __restrict__ cum, uint32_t* __restrict__ blk)
{
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nb) return;
    constexpr uint32_t WPB = BlockGeom<WPP>::WPB;
    uint32_t w[WPB];
    for (uint32_t i = 0; i < WPB; ++i) w[i] = 0;
    for (uint32_t s = 0; s < NLET; ++s) w[s] = cum[s * (nb + 1) + q];
    pack_planes<WPP>(bwt, n, q, w);
    uint32_t* dst = blk + q * WPB;
    for (uint32_t i = 0; i < WPB; ++i) dst[i] = w[i];
}

template <int WPP>
__global__ __launch_bounds__(256) void unpack_blocks_kernel(const uint32_t* __restrict__ blk, uint64_t n, uint8_t* __restrict__ bwt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr uint32_t SPB = BlockGeom<WPP>::SPB, WPB = BlockGeom<WPP>::WPB;
    const uint64_t q = i / SPB; const uint32_t off = (uint32_t)(i - q * SPB), w = off >> 5, t = off & 31u;
    const uint32_t* b = blk + q * WPB;
    bwt[i] = (uint8_t)(((b[5 + w] >> t) & 1u) | (((b[5 + WPP + w] >> t) & 1u) << 1) | (((b[5 + 2 * WPP + w] >> t) & 1u) << 2));
}

template <int WPP>
static int pack_direction(gm_index* ix, int d, const uint8_t* d_bwt)
{
    const uint64_t n = ix->nRows, nb = num_blocks<WPP>(n);
    uint32_t* d_cnt = nullptr; void* d_tmp = nullptr; size_t tmpBytes = 0;
    GM_HIP(hipMalloc(&d_cnt, (nb + 1) * NLET * sizeof(uint32_t)));
    hipLaunchKernelGGL(count_blocks_kernel<WPP>, dim3(grid_for(nb + 1)), dim3(256), 0, 0, d_bwt, n, nb, d_cnt);
    GM_HIP(rocprim::exclusive_scan(nullptr, tmpBytes, d_cnt, d_cnt, 0u, nb + 1, rocprim::plus<uint32_t>()));
    GM_HIP(hipMalloc(&d_tmp, tmpBytes ? tmpBytes : 16));
    for (uint32_t s = 0; s < NLET; ++s) {
        size_t tb = tmpBytes;
        GM_HIP(rocprim::exclusive_scan(d_tmp, tb, d_cnt + s * (nb + 1), d_cnt + s * (nb + 1), 0u, nb + 1, rocprim::plus<uint32_t>()));
    }
    ix->blkBytes = nb * BlockGeom<WPP>::BYTES;
    GM_HIP(hipMalloc(&ix->d_blk[d], ix->blkBytes));
    hipLaunchKernelGGL(pack_blocks_kernel<WPP>, dim3(grid_for(nb)), dim3(256), 0, 0, d_bwt, n, nb, d_cnt, ix->d_blk[d]);
    GM_HIP(hipGetLastError());
    if (d == 0) {
        uint32_t tot[NLET];
        for (uint32_t s = 0; s < NLET; ++s) GM_HIP(hipMemcpy(&tot[s], d_cnt + s * (nb + 1) + nb, 4, hipMemcpyDeviceToHost));
        uint32_t acc = ix->nSeq;   // sentinel suffixes occupy rows [0, nSeq)
        for (uint32_t s = 0; s < NLET; ++s) { ix->C[s] = acc; acc += tot[s]; }
        ix->C[NLET] = acc;
        ix->alphabet = tot[SYM_N] ? 5 : 4;
        if (acc != ix->nRows) { set_error("BWT letter counts (%u) do not add up to the row count (%llu)", acc, (unsigned long long)ix->nRows); return GM_ERR_BAD_ARG; }
    }
    GM_HIP(hipDeviceSynchronize());
    hipFree(d_cnt); hipFree(d_tmp);
    return GM_OK;
}

static int pack_dispatch(gm_index* ix, int d, const uint8_t* d_bwt)
{
    switch (ix->wpp) {
        case 1: return pack_direction<1>(ix, d, d_bwt);
        case 3: return pack_direction<3>(ix, d, d_bwt);
        case 9: return pack_direction<9>(ix, d, d_bwt);
    }
    return GM_ERR_BAD_ARG;
}

static int select_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device available (this library has no CPU fallback)"); return GM_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { set_error("device %d out of range (%d devices)", device, n); return GM_ERR_BAD_ARG; }
    GM_HIP(hipSetDevice(device));
    return GM_OK;
}

static uint32_t wpp_of_block_bytes(uint32_t bb)
{
    if (bb == 0) {
        const char* e = getenv("GM_BLOCK_BYTES");
        bb = e ? (uint32_t)atoi(e) : 64u;
    }
    return bb == 32 ? 1u : bb == 64 ? 3u : bb == 128 ? 9u : 0u;
}

static int index_common_setup(gm_index* ix, const uint8_t* codes, const uint64_t* seq_len, uint32_t n_seq, uint32_t sampling, uint32_t block_bytes, int device)
{
    ix->device = device; ix->nSeq = n_seq; ix->sampling = sampling;
    ix->wpp = wpp_of_block_bytes(block_bytes);
    if (!ix->wpp) { set_error("block_bytes must be 32, 64 or 128"); return GM_ERR_BAD_ARG; }
    ix->cum.assign((size_t)n_seq + 1, 0);
    for (uint32_t s = 0; s < n_seq; ++s) {
        if (seq_len[s] == 0) { set_error("empty sequences are not indexed (src/indexing.hpp:228-231)"); return GM_ERR_BAD_ARG; }
        ix->cum[s + 1] = ix->cum[s] + seq_len[s];
    }
    ix->textLen = ix->cum[n_seq];
    ix->nRows = ix->textLen + n_seq;
    if (ix->nRows >= 0xFFFFFFFFull) { set_error("index of %llu rows needs 64-bit positions (not in this build)", (unsigned long long)ix->nRows); return GM_ERR_TOO_LONG; }
    hipDeviceProp_t prop;
    GM_HIP(hipGetDeviceProperties(&prop, device));
    ix->numCU = prop.multiProcessorCount;
    GM_HIP(hipMalloc(&ix->d_text, ix->textLen + 16));
    GM_HIP(hipMemcpy(ix->d_text, codes, ix->textLen, hipMemcpyHostToDevice));
    GM_HIP(hipMalloc(&ix->d_cum, ((size_t)n_seq + 1) * 8));
    GM_HIP(hipMemcpy(ix->d_cum, ix->cum.data(), ((size_t)n_seq + 1) * 8, hipMemcpyHostToDevice));
    GM_HIP(hipMalloc(&ix->d_small, 64));
    for (int i = 0; i < 4; ++i) GM_HIP(hipEventCreate(&ix->ev[i]));
    return GM_OK;
}

}  // namespace gm

using namespace gm;

extern "C" {

const char* gm_status_string(int s)
{
    switch (s) {
        case GM_OK: return "ok";
        case GM_ERR_NO_DEVICE: return "no HIP device (no CPU fallback exists)";
        case GM_ERR_BAD_ERRORS: return "E > 4 not yet supported.";
        case GM_ERR_BAD_VALUE_BITS: return "value_bits must be 8 or 16";
        case GM_ERR_NEED_LOCATE: return "csv / --exclude-pseudo need an index with SA samples";
        case GM_ERR_BAD_OVERLAP: return "overlap cannot be larger than min(K - 1, K - E - 2)";
        case GM_ERR_BAD_K: return "K out of range (1..128)";
        case GM_ERR_TOO_LONG: return "index too long for 32-bit positions";
        case GM_ERR_BAD_ARG: return "bad argument";
        case GM_ERR_HIP: return "HIP runtime error";
        case GM_ERR_IO: return "I/O error";
        case GM_ERR_OOM: return "out of device memory";
        case GM_ERR_INTERNAL: return "internal invariant violated";
    }
    return "unknown status";
}
const char* gm_last_error(void) { return g_err; }

int gm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

uint32_t gm_default_infix_length(uint32_t K, uint32_t E, int32_t xo) { return default_infix_length(K, E, xo); }

void gm_index_free(gm_index* ix)
{
    if (!ix) return;
    hipSetDevice(ix->device);
    hipFree(ix->d_blk[0]); hipFree(ix->d_blk[1]); hipFree(ix->d_text); hipFree(ix->d_cum);
    hipFree(ix->d_acc); hipFree(ix->d_stack); hipFree(ix->d_small); hipFree(ix->d_table); hipFree(ix->d_blocks); hipFree(ix->d_cumLocal);
    for (int i = 0; i < 4; ++i) if (ix->ev[i]) hipEventDestroy(ix->ev[i]);
    delete ix;
}

int gm_index_build(const uint8_t* codes, const uint64_t* seq_len, uint32_t n_seq, uint32_t sampling, uint32_t block_bytes, int device, gm_index** out)
{
    if (!codes || !seq_len || !n_seq || !out) { set_error("null argument"); return GM_ERR_BAD_ARG; }
    int rc = select_device(device);
    if (rc) return rc;
    gm_index* ix = new (std::nothrow) gm_index();
    if (!ix) return GM_ERR_OOM;
    rc = index_common_setup(ix, codes, seq_len, n_seq, sampling, block_bytes, device);
    uint32_t* d_sa = nullptr; uint8_t* d_bwt = nullptr;
    if (!rc && hipMalloc(&d_sa, ix->nRows * 4) != hipSuccess) rc = GM_ERR_OOM;
    if (!rc && hipMalloc(&d_bwt, ix->nRows) != hipSuccess) rc = GM_ERR_OOM;
    for (int d = 0; d < 2 && !rc; ++d) {
        rc = build_sa_bwt(ix->d_text, ix->d_cum, n_seq, ix->textLen, d, d_sa, d_bwt, &ix->buildRounds[d]);
        if (!rc) rc = pack_dispatch(ix, d, d_bwt);
    }
    hipFree(d_sa); hipFree(d_bwt);
    if (rc) { gm_index_free(ix); return rc; }
    *out = ix;
    return GM_OK;
}

int gm_index_import(const uint8_t* bwt_fwd, const uint8_t* bwt_rev, const uint32_t* sa_fwd, const uint8_t* codes, const uint64_t* seq_len,
                    uint32_t n_seq, uint32_t sampling, uint32_t block_bytes, int device, gm_index** out)
{
    (void)sa_fwd;
    if (!bwt_fwd || !bwt_rev || !codes || !seq_len || !n_seq || !out) { set_error("null argument"); return GM_ERR_BAD_ARG; }
    int rc = select_device(device);
    if (rc) return rc;
    gm_index* ix = new (std::nothrow) gm_index();
    if (!ix) return GM_ERR_OOM;
    rc = index_common_setup(ix, codes, seq_len, n_seq, sampling, block_bytes, device);
    uint8_t* d_bwt = nullptr;
    if (!rc && hipMalloc(&d_bwt, ix->nRows) != hipSuccess) rc = GM_ERR_OOM;
    for (int d = 0; d < 2 && !rc; ++d) {
        if (hipMemcpy(d_bwt, d ? bwt_rev : bwt_fwd, ix->nRows, hipMemcpyHostToDevice) != hipSuccess) { rc = GM_ERR_HIP; break; }
        rc = pack_dispatch(ix, d, d_bwt);
    }
    hipFree(d_bwt);
    if (rc) { gm_index_free(ix); return rc; }
    *out = ix;
    return GM_OK;
}

int gm_index_export_bwt(const gm_index* ix, uint8_t* bwt_fwd, uint8_t* bwt_rev)
{
    if (!ix || !bwt_fwd || !bwt_rev) return GM_ERR_BAD_ARG;
    GM_HIP(hipSetDevice(ix->device));
    uint8_t* d_bwt = nullptr;
    GM_HIP(hipMalloc(&d_bwt, ix->nRows));
    for (int d = 0; d < 2; ++d) {
        switch (ix->wpp) {
            case 1: hipLaunchKernelGGL(unpack_blocks_kernel<1>, dim3(grid_for(ix->nRows)), dim3(256), 0, 0, ix->d_blk[d], ix->nRows, d_bwt); break;
            case 3: hipLaunchKernelGGL(unpack_blocks_kernel<3>, dim3(grid_for(ix->nRows)), dim3(256), 0, 0, ix->d_blk[d], ix->nRows, d_bwt); break;
            default: hipLaunchKernelGGL(unpack_blocks_kernel<9>, dim3(grid_for(ix->nRows)), dim3(256), 0, 0, ix->d_blk[d], ix->nRows, d_bwt); break;
        }
        GM_HIP(hipMemcpy(d ? bwt_rev : bwt_fwd, d_bwt, ix->nRows, hipMemcpyDeviceToHost));
    }
    hipFree(d_bwt);
    return GM_OK;
}

int gm_index_get_info(const gm_index* ix, gm_index_info* info)
{
    if (!ix || !info) return GM_ERR_BAD_ARG;
    info->n_rows = ix->nRows; info->text_len = ix->textLen; info->n_seq = ix->nSeq; info->sampling = ix->sampling;
    info->alphabet_size = ix->alphabet;
    info->block_bytes = ix->wpp == 1 ? 32 : ix->wpp == 3 ? 64 : 128;
    info->device_bytes = 2 * ix->blkBytes + ix->textLen + (ix->nSeq + 1) * 8ull;
    info->device = ix->device;
    return GM_OK;
}

}  // extern "C"

// ---- gm_map ---------------------------------------------------------------------------------------------------
namespace gm {

template <typename T> static int grow(T** p, uint64_t* cap, uint64_t need)
{
    if (*cap >= need) return GM_OK;
    if (*p) { hipFree(*p); *p = nullptr; *cap = 0; }
    hipError_t e = hipMalloc(p, need * sizeof(T));
    if (e != hipSuccess) { set_error("hipMalloc of %llu bytes failed: %s", (unsigned long long)(need * sizeof(T)), hipGetErrorString(e)); return GM_ERR_OOM; }
    *cap = need;
    return GM_OK;
}

template <int WPP>
static int launch_search(gm_index* ix, const SearchArgs& A, unsigned blocks, hipStream_t st)
{
    hipLaunchKernelGGL(search_kernel<WPP>, dim3(blocks), dim3(256), 0, st, A);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

template <int WPP> static int occupancy_blocks(int* out)
{
    int nb = 0;
    GM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, search_kernel<WPP>, 256, 0));
    *out = nb;
    return GM_OK;
}

static int map_impl(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
                    const uint64_t* intervals, uint64_t n_intervals, const uint32_t* seq_file_id, void* d_out, hipStream_t st)
{
    (void)seq_file_id;
    if (!ix || !p || !d_out) { set_error("null argument"); return GM_ERR_BAD_ARG; }
    if (p->value_bits != 8 && p->value_bits != 16) return GM_ERR_BAD_VALUE_BITS;
    if (p->exclude_pseudo) { set_error("--exclude-pseudo needs the locate path"); return GM_ERR_NEED_LOCATE; }
    if (text_begin + text_len > ix->textLen || (uint64_t)first_seq + n_seq > ix->nSeq || n_seq == 0) { set_error("slice outside the index"); return GM_ERR_BAD_ARG; }
    if (ix->cum[first_seq] != text_begin || ix->cum[first_seq + n_seq] != text_begin + text_len) { set_error("slice does not match its sequences"); return GM_ERR_BAD_ARG; }
    if (p->E > MAX_ERRORS) return GM_ERR_BAD_ERRORS;
    GM_HIP(hipSetDevice(ix->device));

    const uint32_t infix = p->infix > 0 ? (uint32_t)p->infix : default_infix_length(p->K, p->E, p->overlap);
    if (infix == 0) return GM_ERR_BAD_OVERLAP;
    MapPlan plan;
    int rc = make_map_plan(p->K, p->E, infix, p->revcompl, text_len, intervals, n_intervals, &plan);
    if (rc) return rc;   // PlanError values coincide with gm_status

    // shard [kmer_begin, kmer_end): blocks whose first k-mer lies inside
    uint64_t blockBegin = 0, blockEnd = plan.numBlocks;
    if (p->kmer_begin != 0 || p->kmer_end != 0) {
        if (plan.useList) {
            auto lo = std::lower_bound(plan.blocks.begin(), plan.blocks.end(), p->kmer_begin, [](const std::pair<uint32_t, uint32_t>& b, uint64_t v) { return b.first < v; });
            auto hi = std::lower_bound(plan.blocks.begin(), plan.blocks.end(), p->kmer_end, [](const std::pair<uint32_t, uint32_t>& b, uint64_t v) { return b.first < v; });
            blockBegin = lo - plan.blocks.begin(); blockEnd = hi - plan.blocks.begin();
        } else {
            blockBegin = std::min<uint64_t>((p->kmer_begin + plan.stepSize - 1) / plan.stepSize, plan.numBlocks);
            blockEnd = std::min<uint64_t>((p->kmer_end + plan.stepSize - 1) / plan.stepSize, plan.numBlocks);
        }
        if (blockEnd < blockBegin) blockEnd = blockBegin;
    }
    const uint32_t rpb = plan.nSearches * plan.nStrands;
    const uint64_t numRoots = (blockEnd - blockBegin) * rpb;

    // ---- workspace ----
    rc = grow(&ix->d_acc, &ix->accCap, text_len + 4); if (rc) return rc;
    rc = grow(&ix->d_table, &ix->tableCap, (uint64_t)plan.table.size()); if (rc) return rc;
    rc = grow(&ix->d_cumLocal, &ix->cumLocalCap, (uint64_t)n_seq + 1); if (rc) return rc;
    if (plan.useList) { rc = grow(&ix->d_blocks, &ix->blocksCap, std::max<uint64_t>(plan.blocks.size(), 1)); if (rc) return rc; }

    int perCU = 0;
    switch (ix->wpp) { case 1: rc = occupancy_blocks<1>(&perCU); break; case 3: rc = occupancy_blocks<3>(&perCU); break; default: rc = occupancy_blocks<9>(&perCU); break; }
    if (rc) return rc;
    if (const char* e = getenv("GM_BLOCKS_PER_CU")) { int v = atoi(e); if (v > 0) perCU = std::min(perCU, v); }
    if (perCU < 1) perCU = 1;
    uint64_t blocks = (uint64_t)ix->numCU * perCU;
    const uint64_t useful = (numRoots + 255) / 256;
    if (blocks > useful) blocks = std::max<uint64_t>(useful, 1);
    const uint32_t depth = stack_bound(p->E, plan.stepSize);
    rc = grow(&ix->d_stack, &ix->stackCap, blocks * 256ull * depth); if (rc) return rc;

    GM_HIP(hipMemcpyAsync(ix->d_table, plan.table.data(), plan.table.size() * sizeof(OssRecord), hipMemcpyHostToDevice, st));
    if (plan.useList && !plan.blocks.empty())
        GM_HIP(hipMemcpyAsync(ix->d_blocks, plan.blocks.data(), plan.blocks.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
    std::vector<uint64_t> cumLocal((size_t)n_seq + 1);
    for (uint32_t s = 0; s <= n_seq; ++s) cumLocal[s] = ix->cum[first_seq + s] - text_begin;
    GM_HIP(hipMemcpyAsync(ix->d_cumLocal, cumLocal.data(), cumLocal.size() * 8, hipMemcpyHostToDevice, st));
    GM_HIP(hipStreamSynchronize(st));   // host staging buffers go out of scope; also keeps the timed region device-only

    GM_HIP(hipEventRecord(ix->ev[0], st));
    GM_HIP(hipMemsetAsync(ix->d_acc, 0, (text_len + 4) * sizeof(uint32_t), st));
    GM_HIP(hipMemsetAsync(ix->d_small, 0, 64, st));

    SearchArgs A;
    A.blk[0] = ix->d_blk[0]; A.blk[1] = ix->d_blk[1];
    for (uint32_t c = 0; c <= NLET; ++c) A.C[c] = ix->C[c];
    A.nRows = (uint32_t)ix->nRows;
    A.text = ix->d_text + text_begin;
    A.acc = ix->d_acc;
    A.K = p->K; A.E = p->E;
    A.stepSize = plan.stepSize; A.nSearches = plan.nSearches; A.rootsPerBlock = rpb;
    A.numKmers = (uint32_t)plan.numKmers;
    A.blockBegin = blockBegin; A.numRoots = numRoots;
    A.blockList = plan.useList ? ix->d_blocks : nullptr;
    A.table = ix->d_table;
    A.stack = ix->d_stack; A.stackDepth = depth;
    A.workCounter = reinterpret_cast<unsigned long long*>(ix->d_small);
    A.errorFlag = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ix->d_small) + 8);
    A.counters = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ix->d_small) + 16);

    GM_HIP(hipEventRecord(ix->ev[1], st));
    if (numRoots > 0) {
        switch (ix->wpp) {
            case 1: rc = launch_search<1>(ix, A, (unsigned)blocks, st); break;
            case 3: rc = launch_search<3>(ix, A, (unsigned)blocks, st); break;
            default: rc = launch_search<9>(ix, A, (unsigned)blocks, st); break;
        }
        if (rc) return rc;
    }
    GM_HIP(hipEventRecord(ix->ev[2], st));
    if (text_len > 0) {
        if (p->value_bits == 8) {
            hipLaunchKernelGGL(finalize_kernel<uint8_t>, dim3(grid_for((text_len + 3) / 4)), dim3(256), 0, st, ix->d_acc, (uint8_t*)d_out, text_len, 255u);
            hipLaunchKernelGGL(reset_limits_kernel<uint8_t>, dim3(n_seq), dim3(64), 0, st, (uint8_t*)d_out, ix->d_cumLocal, n_seq, p->K);
        } else {
            hipLaunchKernelGGL(finalize_kernel<uint16_t>, dim3(grid_for((text_len + 3) / 4)), dim3(256), 0, st, ix->d_acc, (uint16_t*)d_out, text_len, 65535u);
            hipLaunchKernelGGL(reset_limits_kernel<uint16_t>, dim3(n_seq), dim3(64), 0, st, (uint16_t*)d_out, ix->d_cumLocal, n_seq, p->K);
        }
        GM_HIP(hipGetLastError());
    }
    GM_HIP(hipEventRecord(ix->ev[3], st));
    ix->evValid = true;
    ix->stats = gm_map_stats{};
    uint64_t kmers = 0;
    if (plan.useList) { for (uint64_t b = blockBegin; b < blockEnd; ++b) kmers += plan.blocks[b].second; }
    else if (blockEnd > blockBegin) kmers = std::min<uint64_t>(blockEnd * plan.stepSize, plan.numKmers) - blockBegin * plan.stepSize;
    ix->stats.kmers = kmers; ix->stats.roots = numRoots;
    return GM_OK;
}

static int check_device_error(gm_index* ix)
{
    uint32_t flag = 0;
    GM_HIP(hipMemcpy(&flag, reinterpret_cast<char*>(ix->d_small) + 8, 4, hipMemcpyDeviceToHost));
    if (flag) { set_error("device-side invariant violated (lane stack overflow)"); return GM_ERR_INTERNAL; }
    return GM_OK;
}

}  // namespace gm

extern "C" {

int gm_map_device(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
                  const uint64_t* intervals, uint64_t n_intervals, const uint32_t* seq_file_id, void* out_device, void* stream)
{
    return map_impl(ix, text_begin, text_len, first_seq, n_seq, p, intervals, n_intervals, seq_file_id, out_device, (hipStream_t)stream);
}

int gm_map(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
           const uint64_t* intervals, uint64_t n_intervals, const uint32_t* seq_file_id, void* out_host)
{
    if (!ix || !p || !out_host) { set_error("null argument"); return GM_ERR_BAD_ARG; }
    if (p->value_bits != 8 && p->value_bits != 16) return GM_ERR_BAD_VALUE_BITS;
    GM_HIP(hipSetDevice(ix->device));
    void* d_out = nullptr;
    const size_t bytes = (size_t)text_len * (p->value_bits / 8);
    GM_HIP(hipMalloc(&d_out, bytes + 16));
    int rc = map_impl(ix, text_begin, text_len, first_seq, n_seq, p, intervals, n_intervals, seq_file_id, d_out, nullptr);
    if (!rc) { hipError_t e = hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost); if (e != hipSuccess) { set_error("copy back failed: %s", hipGetErrorString(e)); rc = GM_ERR_HIP; } }
    if (!rc) rc = check_device_error(ix);
    hipFree(d_out);
    return rc;
}

int gm_last_map_stats(const gm_index* cix, gm_map_stats* out)
{
    gm_index* ix = const_cast<gm_index*>(cix);
    if (!ix || !out) return GM_ERR_BAD_ARG;
    if (!ix->evValid) { *out = gm_map_stats{}; return GM_OK; }
    GM_HIP(hipSetDevice(ix->device));
    GM_HIP(hipEventSynchronize(ix->ev[3]));
    float a = 0, b = 0;
    GM_HIP(hipEventElapsedTime(&a, ix->ev[1], ix->ev[2]));
    GM_HIP(hipEventElapsedTime(&b, ix->ev[0], ix->ev[3]));
    ix->stats.search_ms = a; ix->stats.total_ms = b;
    unsigned long long cnt[2] = {0, 0};
    GM_HIP(hipMemcpy(cnt, reinterpret_cast<char*>(ix->d_small) + 16, 16, hipMemcpyDeviceToHost));
    ix->stats.node_steps = cnt[0]; ix->stats.rank_lines = cnt[1];
    int rc = check_device_error(ix);
    *out = ix->stats;
    return rc;
}

}  // extern "C"
