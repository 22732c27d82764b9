// gm_api.hip -- implementation of the C ABI declared in include/genmap_amd.h (HIP, gfx950).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include "gm_internal.h"
#include "gm_host.h"
#include "gm_kernels.h"
#include "gm_longk.h"

namespace gm {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}

#define GM_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { gm::set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return (e_ == hipErrorOutOfMemory) ? GM_ERR_OOM : GM_ERR_HIP; } } while (0)

// every kernel launched over rows or text positions is a grid-stride loop: a dispatch holds fewer than 2^32 work-items per
// dimension and wide indexes have more rows than that
static inline unsigned grid_for(uint64_t n, unsigned bs = 256) { return (unsigned)std::min<uint64_t>((n + bs - 1) / bs, 1u << 22); }

// d_small: work counter (8 B) | pad | statistics counters (50 x 8 B at +16), zeroed by every call  ||  +512: sticky error flag
// (set by a device-side invariant check, surfaced and cleared by the next host-side check: gm_map, gm_index_sync, gm_last_map_stats)
constexpr size_t SMALL_BYTES = 1024, SMALL_ZEROED = 512, SMALL_ERR_OFF = 512;   // [0,16) work counter, [16,512) statistics, 512: sticky error flag

// ---- rank block construction -------------------------------------------------------------------------------
// cnt[c * (nb + 1) + q] = letters c in block q; entry nb is zero so that the exclusive scan leaves the total there.
template <int WPP>
__global__ __launch_bounds__(256) void count_blocks_kernel(const uint8_t* __restrict__ bwt, uint64_t n, uint64_t nb, typename BlockGeom<WPP>::row_t* __restrict__ cnt)
{
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q <= nb; q += (uint64_t)gridDim.x * blockDim.x) {
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
}

template <int WPP>
__global__ __launch_bounds__(256) void pack_blocks_kernel(const uint8_t* __restrict__ bwt, uint64_t n, uint64_t nb, const typename BlockGeom<WPP>::row_t* __restrict__ cum, uint32_t* __restrict__ blk)
{
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nb; q += (uint64_t)gridDim.x * blockDim.x) {
    constexpr uint32_t WPB = BlockGeom<WPP>::WPB;
    uint32_t w[WPB];
    for (uint32_t i = 0; i < WPB; ++i) w[i] = 0;
    for (uint32_t s = 0; s < NLET; ++s) {
        const uint64_t c = cum[s * (nb + 1) + q];
        if (BlockGeom<WPP>::HDRW == 5u) w[s] = (uint32_t)c; else { w[2 * s] = (uint32_t)c; w[2 * s + 1] = (uint32_t)(c >> 32); }
    }
    pack_planes<WPP>(bwt, n, q, w);
    uint32_t* dst = blk + q * WPB;
    for (uint32_t i = 0; i < WPB; ++i) dst[i] = w[i];
    }
}

template <int WPP>
__global__ __launch_bounds__(256) void unpack_blocks_kernel(const uint32_t* __restrict__ blk, uint64_t n, uint8_t* __restrict__ bwt)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    constexpr uint32_t SPB = BlockGeom<WPP>::SPB, WPB = BlockGeom<WPP>::WPB;
    const uint64_t q = i / SPB; const uint32_t off = (uint32_t)(i - q * SPB), w = off >> 5, t = off & 31u;
    const uint32_t* b = blk + q * WPB;
    constexpr uint32_t H = BlockGeom<WPP>::HDRW;
    bwt[i] = (uint8_t)(((b[H + w] >> t) & 1u) | (((b[H + WPP + w] >> t) & 1u) << 1) | (((b[H + 2 * WPP + w] >> t) & 1u) << 2));
    }
}

// 4-bit packed copy of the text: chunk c holds symbols [32c, 32c + 32), symbol i in nibble (i & 1) of byte (i >> 1)
__global__ __launch_bounds__(256) void pack_text4_kernel(const uint8_t* __restrict__ codes, uint64_t textLen, uint32_t* __restrict__ out, uint64_t nWords)
{
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nWords; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        for (uint32_t j = 0; j < 8; ++j) { const uint64_t i = w * 8 + j; const uint32_t c = i < textLen ? codes[i] : 0u; v |= (c & 15u) << (4u * j); }
        out[w] = v;
    }
}

// 2-bit packed copy of the text for needle windows (SearchArgs::win2): chunk c holds symbols [64c, 64c + 64), symbol i at bits 2 (i & 63) of the chunk, an N as A;
// flags[b] bit j: chunk 8b + j holds an N (16 bits: a window's chunks are one aligned load whatever chunk it starts in).  One thread per flag entry.
__global__ __launch_bounds__(256) void pack_text2_kernel(const uint4* __restrict__ text4, uint64_t nChunks4, uint4* __restrict__ text2, uint64_t nChunks2, uint16_t* __restrict__ flags, uint64_t nFlags)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nFlags; b += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t f = 0;
        for (uint32_t j = 0; j < 16; ++j) {
            const uint64_t c = 8 * b + j;
            uint32_t o[4], anyN = 0;
            for (uint32_t h = 0; h < 2; ++h) {
                const uint4 v = 2 * c + h < nChunks4 ? text4[2 * c + h] : make_uint4(0u, 0u, 0u, 0u);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                anyN |= (v.x | v.y | v.z | v.w) & 0x44444444u;
                for (uint32_t k = 0; k < 2; ++k) {
                    unsigned long long t = ((unsigned long long)w[2 * k + 1] << 32 | w[2 * k]) & 0x3333333333333333ull;
                    t = (t | (t >> 2)) & 0x0F0F0F0F0F0F0F0Full;
                    t = (t | (t >> 4)) & 0x00FF00FF00FF00FFull;
                    t = (t | (t >> 8)) & 0x0000FFFF0000FFFFull;
                    t = (t | (t >> 16)) & 0x00000000FFFFFFFFull;
                    o[2 * h + k] = (uint32_t)t;
                }
            }
            if (j < 8 && c < nChunks2) text2[c] = make_uint4(o[0], o[1], o[2], o[3]);
            f |= (anyN ? 1u : 0u) << j;
        }
        flags[b] = (uint16_t)f;
    }
}

// sentinel text: sequence s occupies [cum[s] + s, cum[s+1] + s), its sentinel follows
__global__ __launch_bounds__(256) void sentinel_text_kernel(const uint8_t* __restrict__ codes, const uint64_t* __restrict__ cum, uint32_t nSeq, uint64_t textLen,
                                                            uint8_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < textLen + nSeq; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i < textLen) {
            uint32_t lo = 0, hi = nSeq;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cum[mid] <= i) lo = mid; else hi = mid; }
            out[i + lo] = codes[i];
        } else {
            const uint32_t s = (uint32_t)(i - textLen);
            out[cum[s + 1] + s] = (uint8_t)SYM_SENT;
        }
    }
}

// verification records (gm_kernels.h: CTX_*): one 32-byte record per forward SA row = {SA[row], 56 text symbols around it}
__global__ __launch_bounds__(256) void ctx_build_kernel(const uint32_t* __restrict__ sa, const uint8_t* __restrict__ textS, uint64_t nRows, uint4* __restrict__ ctx)
{
    for (uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nRows; row += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t p0 = sa[row];
    const uint8_t* p = textS + ((long long)p0 - CTX_LEFT);   // 512 sentinel bytes of padding on both sides of textS
    uint32_t w[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const uintptr_t u = reinterpret_cast<uintptr_t>(p + 8 * k);
        const uint64_t* b = reinterpret_cast<const uint64_t*>(u & ~static_cast<uintptr_t>(7));
        const uint32_t sh = (uint32_t)(u & 7u) * 8u;
        const uint64_t lo = b[0], hi = b[1];
        uint64_t v = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;   // 8 symbols, one per byte
        v = (v | (v >> 4)) & 0x00FF00FF00FF00FFull;              // bytes -> nibbles
        v = (v | (v >> 8)) & 0x0000FFFF0000FFFFull;
        v = (v | (v >> 16)) & 0x00000000FFFFFFFFull;
        w[k] = (uint32_t)v;
    }
    ctx[row * 2] = make_uint4(p0, w[0], w[1], w[2]);
    ctx[row * 2 + 1] = make_uint4(w[3], w[4], w[5], w[6]);
    }
}

static int make_ctx(gm_index* ix)
{
    // 32 B per row (99 GB for a 3.1 Gbp index): only when it leaves the device at least half empty -- the q-mer tables (up to
    // 17 GB), the per-call workspaces and the caller's own buffers come later.  Without it verification reads SA + text.
    if (ix->wide) return GM_OK;   // the record holds a 32-bit position
    size_t freeB = 0, totalB = 0;
    const uint64_t bytes = ix->nRows * 32ull;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess || bytes > freeB || freeB - bytes < totalB / 2) return GM_OK;
    if (hipMalloc(&ix->d_ctx, bytes) != hipSuccess) { (void)hipGetLastError(); ix->d_ctx = nullptr; return GM_OK; }
    hipLaunchKernelGGL(ctx_build_kernel, dim3(grid_for(ix->nRows)), dim3(256), 0, 0, (const uint32_t*)ix->d_sa, ix->d_textS, ix->nRows, ix->d_ctx);
    GM_HIP(hipGetLastError());
    GM_HIP(hipDeviceSynchronize());
    return GM_OK;
}

static int make_sentinel_text(gm_index* ix)
{
    GM_HIP(hipMalloc(&ix->d_textSAlloc, ix->nRows + 1024));
    GM_HIP(hipMemset(ix->d_textSAlloc, (int)SYM_SENT, ix->nRows + 1024));   // padding reads as sentinels
    ix->d_textS = ix->d_textSAlloc + 512;
    hipLaunchKernelGGL(sentinel_text_kernel, dim3(grid_for(ix->nRows)), dim3(256), 0, 0, ix->d_text, ix->d_cum, ix->nSeq, ix->textLen, ix->d_textS);
    GM_HIP(hipGetLastError());
    GM_HIP(hipDeviceSynchronize());
    return GM_OK;
}

template <int WPP>
static int pack_direction(gm_index* ix, int d, const uint8_t* d_bwt)
{
    typedef typename BlockGeom<WPP>::row_t row_t;
    const uint64_t n = ix->nRows, nb = num_blocks<WPP>(n);
    row_t* d_cnt = nullptr; void* d_tmp = nullptr; size_t tmpBytes = 0;
    GM_HIP(hipMalloc(&d_cnt, (nb + 1) * NLET * sizeof(row_t)));
    hipLaunchKernelGGL(count_blocks_kernel<WPP>, dim3(grid_for(nb + 1)), dim3(256), 0, 0, d_bwt, n, nb, d_cnt);
    GM_HIP(rocprim::exclusive_scan(nullptr, tmpBytes, d_cnt, d_cnt, (row_t)0, nb + 1, rocprim::plus<row_t>()));
    GM_HIP(hipMalloc(&d_tmp, tmpBytes ? tmpBytes : 16));
    for (uint32_t s = 0; s < NLET; ++s) {
        size_t tb = tmpBytes;
        GM_HIP(rocprim::exclusive_scan(d_tmp, tb, d_cnt + s * (nb + 1), d_cnt + s * (nb + 1), (row_t)0, nb + 1, rocprim::plus<row_t>()));
    }
    ix->blkBytes = nb * BlockGeom<WPP>::BYTES;
    GM_HIP(hipMalloc(&ix->d_blk[d], ix->blkBytes));
    hipLaunchKernelGGL(pack_blocks_kernel<WPP>, dim3(grid_for(nb)), dim3(256), 0, 0, d_bwt, n, nb, d_cnt, ix->d_blk[d]);
    GM_HIP(hipGetLastError());
    if (d == 0) {
        row_t tot[NLET];
        for (uint32_t s = 0; s < NLET; ++s) GM_HIP(hipMemcpy(&tot[s], d_cnt + s * (nb + 1) + nb, sizeof(row_t), hipMemcpyDeviceToHost));
        uint64_t acc = ix->nSeq;   // sentinel suffixes occupy rows [0, nSeq)
        for (uint32_t s = 0; s < NLET; ++s) { ix->C[s] = acc; acc += tot[s]; }
        ix->C[NLET] = acc;
        ix->alphabet = tot[SYM_N] ? 5 : 4;
        if (acc != ix->nRows) { set_error("BWT letter counts (%llu) do not add up to the row count (%llu)", (unsigned long long)acc, (unsigned long long)ix->nRows); return GM_ERR_BAD_ARG; }
    }
    GM_HIP(hipDeviceSynchronize());
    hipFree(d_cnt); hipFree(d_tmp);
    return GM_OK;
}

static int pack_dispatch(gm_index* ix, int d, const uint8_t* d_bwt)
{
    switch (ix->wpp) {
        case 1: return pack_direction<1>(ix, d, d_bwt);
        case 2: return pack_direction<2>(ix, d, d_bwt);
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
    if (bb == 0) bb = 32u;   // measured: 32-B blocks sustain the highest random-read rate (profiles/r01_gather_*)
    return bb == 32 ? 1u : bb == 64 ? 3u : bb == 128 ? 9u : 0u;
}

static int index_common_setup(gm_index* ix, const uint8_t* codes, const uint64_t* seq_len, uint32_t n_seq, uint32_t sampling, uint32_t block_bytes, int device)
{
    ix->device = device; ix->nSeq = n_seq; ix->sampling = sampling;
    if (sampling > 64) { set_error("sampling rate %u: 0 (no suffix array), 1 (all of it) or 2..64", sampling); return GM_ERR_BAD_ARG; }
    const bool forceWide = (block_bytes & GM_BLOCK_WIDE_ROWS) != 0;
    block_bytes &= ~(uint32_t)GM_BLOCK_WIDE_ROWS;
    ix->wpp = wpp_of_block_bytes(block_bytes);
    if (!ix->wpp) { set_error("block_bytes must be 32, 64 or 128"); return GM_ERR_BAD_ARG; }
    ix->cum.assign((size_t)n_seq + 1, 0);
    for (uint32_t s = 0; s < n_seq; ++s) {
        if (seq_len[s] == 0) { set_error("empty sequences are not indexed (src/indexing.hpp:228-231)"); return GM_ERR_BAD_ARG; }
        ix->cum[s + 1] = ix->cum[s] + seq_len[s];
    }
    ix->textLen = ix->cum[n_seq];
    ix->nRows = ix->textLen + n_seq;
    // 2^32 - 1 rows or more: 64-bit rows, ranges and text positions (the reference's 64-bit BWT variants, src/indexing.hpp:158-169)
    ix->wide = forceWide || ix->nRows >= 0xFFFFFFFFull;
    if (ix->wide) ix->wpp = WPP_WIDE;
    if (ix->nRows >= (1ull << 40)) { set_error("index of %llu rows is beyond this build (2^40 rows)", (unsigned long long)ix->nRows); return GM_ERR_TOO_LONG; }
    hipDeviceProp_t prop;
    GM_HIP(hipGetDeviceProperties(&prop, device));
    ix->numCU = prop.multiProcessorCount;
    GM_HIP(hipMalloc(&ix->d_textAlloc, ix->textLen + 64));
    GM_HIP(hipMemset(ix->d_textAlloc, 0, ix->textLen + 64));
    ix->d_text = ix->d_textAlloc + 16;
    GM_HIP(hipMemcpy(ix->d_text, codes, ix->textLen, hipMemcpyHostToDevice));
    {
        const uint64_t nChunks = ix->textLen / 32 + 20, nWords = nChunks * 4;   // +20 chunks: window staging (up to 2K-1 = 509 symbols) may read past the end
        GM_HIP(hipMalloc(&ix->d_text4, nChunks * 16));
        hipLaunchKernelGGL(pack_text4_kernel, dim3(grid_for(nWords)), dim3(256), 0, 0, ix->d_text, ix->textLen, reinterpret_cast<uint32_t*>(ix->d_text4), nWords);
        GM_HIP(hipGetLastError());
    }
    GM_HIP(hipMalloc(&ix->d_cum, ((size_t)n_seq + 1) * 8));
    GM_HIP(hipMemcpy(ix->d_cum, ix->cum.data(), ((size_t)n_seq + 1) * 8, hipMemcpyHostToDevice));
    GM_HIP(hipMalloc(&ix->d_small, SMALL_BYTES));
    GM_HIP(hipMemset(ix->d_small, 0, SMALL_BYTES));
    for (int i = 0; i < 4; ++i) GM_HIP(hipEventCreate(&ix->ev[i]));
    for (uint32_t i = 0; i < gm_index::EV_RING; ++i) for (int j = 0; j < 2; ++j) GM_HIP(hipEventCreate(&ix->evRing[i][j]));
    GM_HIP(hipEventCreateWithFlags(&ix->evDone, hipEventDisableTiming));
    return GM_OK;
}

// number of suffix array rows a sampling rate keeps: the offsets 0, s, 2s, .. of every sequence
static uint64_t expected_samples(const gm_index* ix)
{
    uint64_t t = 0;
    const uint64_t s = std::max<uint32_t>(ix->sampling, 1u);
    for (uint32_t i = 0; i < ix->nSeq; ++i) t += (ix->cum[i + 1] - ix->cum[i] + s - 1) / s;
    return t;
}

// the sampled form of a full forward suffix array (device): marks + "samples before this word" per 32 rows, and the kept values
static int sample_sa(gm_index* ix, const void* d_saFull)   // entries as wide as the index's rows
{
    const uint64_t n = ix->nRows, words = (n + 31) / 32;
    // The marked rows are the in-sequence offsets that are multiples of the rate: their number is known exactly on the host, in 64
    // bits -- the device-side counts below are 32-bit and would wrap silently
    if (expected_samples(ix) >= (1ull << 32)) { set_error("more than 2^32 suffix array samples: use a larger sampling rate"); return GM_ERR_TOO_LONG; }
    GM_HIP(hipMalloc(&ix->d_saMark, words * sizeof(uint2)));   // (owned by the index: gm_index_free releases it on every path)
    if (ix->wide) hipLaunchKernelGGL(sa_mark_kernel<uint64_t>, dim3(grid_for(words)), dim3(256), 0, 0, (const uint64_t*)d_saFull, ix->d_cum, ix->nSeq, n, ix->sampling, ix->d_saMark);
    else hipLaunchKernelGGL(sa_mark_kernel<uint32_t>, dim3(grid_for(words)), dim3(256), 0, 0, (const uint32_t*)d_saFull, ix->d_cum, ix->nSeq, n, ix->sampling, ix->d_saMark);
    uint32_t *d_cnt = nullptr, *d_before = nullptr; void* d_tmp = nullptr; size_t tmpBytes = 0;
    int rc = hipGetLastError() == hipSuccess ? GM_OK : GM_ERR_HIP;
    if (!rc && (hipMalloc(&d_cnt, words * 4) != hipSuccess || hipMalloc(&d_before, words * 4) != hipSuccess)) rc = GM_ERR_OOM;
    if (!rc) {
        hipLaunchKernelGGL(sa_mark_counts_kernel, dim3(grid_for(words)), dim3(256), 0, 0, ix->d_saMark, words, d_cnt);
        if (rocprim::exclusive_scan(nullptr, tmpBytes, d_cnt, d_before, 0u, words, rocprim::plus<uint32_t>()) != hipSuccess) rc = GM_ERR_HIP;
    }
    if (!rc && hipMalloc(&d_tmp, tmpBytes ? tmpBytes : 16) != hipSuccess) rc = GM_ERR_OOM;
    if (!rc && rocprim::exclusive_scan(d_tmp, tmpBytes, d_cnt, d_before, 0u, words, rocprim::plus<uint32_t>()) != hipSuccess) rc = GM_ERR_HIP;
    uint32_t last[2] = {0, 0};
    if (!rc) {
        hipLaunchKernelGGL(sa_mark_offsets_kernel, dim3(grid_for(words)), dim3(256), 0, 0, ix->d_saMark, words, d_before);
        if (hipMemcpy(&last[0], d_before + (words - 1), 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&last[1], d_cnt + (words - 1), 4, hipMemcpyDeviceToHost) != hipSuccess) rc = GM_ERR_HIP;
    }
    hipFree(d_cnt); hipFree(d_before); hipFree(d_tmp);
    if (rc) return rc;
    ix->nSamples = (uint64_t)last[0] + last[1];
    if (ix->nSamples != expected_samples(ix)) { set_error("suffix array sampling: %llu rows marked, %llu expected", (unsigned long long)ix->nSamples, (unsigned long long)expected_samples(ix)); return GM_ERR_HIP; }
    GM_HIP(hipMalloc(&ix->d_saSamples, std::max<uint64_t>(ix->nSamples, 1) * (ix->wide ? 8 : 4)));
    if (ix->wide) hipLaunchKernelGGL(sa_compact_kernel<uint64_t>, dim3(grid_for(n)), dim3(256), 0, 0, (const uint64_t*)d_saFull, ix->d_saMark, n, (uint64_t*)ix->d_saSamples);
    else hipLaunchKernelGGL(sa_compact_kernel<uint32_t>, dim3(grid_for(n)), dim3(256), 0, 0, (const uint32_t*)d_saFull, ix->d_saMark, n, (uint32_t*)ix->d_saSamples);
    GM_HIP(hipGetLastError());
    GM_HIP(hipDeviceSynchronize());
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
        case GM_ERR_BAD_K: return "K out of range (1..32768)";
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
uint32_t gm_tuned_infix_length(uint32_t K, uint32_t E) { return (K < 1 || K > MAX_K_LONG || E > MAX_ERRORS) ? 0u : long_k_infix(K, tuned_infix_length(K, E)); }
uint32_t gm_tuned_infix_length_locating(uint32_t K, uint32_t E) { return (K < 1 || K > MAX_K_LONG || E > MAX_ERRORS) ? 0u : long_k_infix(K, tuned_infix_length(K, E, true)); }

void gm_index_free(gm_index* ix)
{
    if (!ix) return;
    hipSetDevice(ix->device);
    hipFree(ix->d_blk[0]); hipFree(ix->d_blk[1]); hipFree(ix->d_textAlloc); hipFree(ix->d_text4); hipFree(ix->d_text2); hipFree(ix->d_nflag); hipFree(ix->d_cum); hipFree(ix->d_sa); hipFree(ix->d_textSAlloc); hipFree(ix->d_C); hipFree(ix->d_ctx);
    hipFree(ix->d_saMark); hipFree(ix->d_saSamples);
    for (auto& kv : ix->qtables) hipFree(kv.second);
    for (auto& kv : ix->jbits) hipFree(kv.second);
    hipFree(ix->d_jinfo2); hipFree(ix->d_seqFile); hipFree(ix->d_rowFile);
    hipFree(ix->d_locCnt); hipFree(ix->d_locOffs); hipFree(ix->d_locEmit); hipFree(ix->d_locSorted); hipFree(ix->d_locTmp); hipFree(ix->d_locSeg); hipFree(ix->d_bits);
    hipFree(ix->d_acc); hipFree(ix->d_stack); hipFree(ix->d_small); hipFree(ix->d_table); hipFree(ix->d_tableL); hipFree(ix->d_blocks); hipFree(ix->d_cumLocal);
    for (int i = 0; i < 4; ++i) if (ix->ev[i]) hipEventDestroy(ix->ev[i]);
    for (uint32_t i = 0; i < gm_index::EV_RING; ++i) for (int j = 0; j < 2; ++j) if (ix->evRing[i][j]) hipEventDestroy(ix->evRing[i][j]);
    if (ix->evDone) hipEventDestroy(ix->evDone);
    hipFree(ix->d_shardOut); hipFree(ix->d_patterns); hipFree(ix->d_jinfo); hipFree(ix->d_cblocks);
    if (ix->h_stage) hipHostFree(ix->h_stage);
    hipFree(ix->d_pkt); hipFree(ix->d_xctl); hipFree(ix->d_xprog); hipFree(ix->d_wmap);
    if (ix->h_xprog) hipHostFree(ix->h_xprog);
    for (auto& e : ix->evX) if (e) hipEventDestroy(e);
    if (ix->stXA) hipStreamDestroy(ix->stXA);
    for (auto& q : ix->stXB) if (q) hipStreamDestroy(q);
    for (auto& e : ix->evXA) if (e) hipEventDestroy(e);
    for (auto& e : ix->evXB) if (e) hipEventDestroy(e);
    if (ix->evXGo) hipEventDestroy(ix->evXGo);
    for (auto& e : ix->evStage) if (e) hipEventDestroy(e);
    if (ix->stCorr) hipStreamDestroy(ix->stCorr);
    if (ix->evCorrGo) hipEventDestroy(ix->evCorrGo);
    if (ix->evCorrDone) hipEventDestroy(ix->evCorrDone);
    if (ix->evCorrStart) hipEventDestroy(ix->evCorrStart);
    if (ix->stCompute) hipStreamDestroy(ix->stCompute);
    if (ix->stCopy) hipStreamDestroy(ix->stCopy);
    for (auto& e : ix->evShard) if (e) hipEventDestroy(e);
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
    void* d_sa = nullptr; uint8_t* d_bwt = nullptr;
    const size_t rb = ix->wide ? 8 : 4;   // bytes per suffix-array entry
    if (!rc && hipMalloc(&d_sa, ix->nRows * rb) != hipSuccess) rc = GM_ERR_OOM;
    if (!rc && hipMalloc(&d_bwt, ix->nRows) != hipSuccess) rc = GM_ERR_OOM;
    // the reverse direction first: the forward suffix array, built last, is the one that stays -- no second array beside the sort's own six
    // (4.32 G rows with 64-bit rows: 35 GB each; with the forward direction first a resident suffix array did not fit beside the second sort)
    for (int k = 0; k < 2 && !rc; ++k) {
        const int d = 1 - k;
        rc = ix->wide ? build_sa_bwt_wide(ix->d_text, ix->d_cum, n_seq, ix->textLen, d, (uint64_t*)d_sa, d_bwt, &ix->buildRounds[d])
                      : build_sa_bwt(ix->d_text, ix->d_cum, n_seq, ix->textLen, d, (uint32_t*)d_sa, d_bwt, &ix->buildRounds[d]);
        if (!rc) rc = pack_dispatch(ix, d, d_bwt);
        if (!rc && d == 0 && sampling == 1) { ix->d_sa = d_sa; d_sa = nullptr; }   // the forward SA stays resident: 4 (8) B/row of 288 GB buys locate without LF walks
        if (!rc && d == 0 && sampling > 1) rc = sample_sa(ix, d_sa);   // -S: keep 1/s of it, locate walks the LF mapping
    }
    hipFree(d_sa); hipFree(d_bwt);
    if (!rc && ix->d_sa) rc = make_sentinel_text(ix);
    if (!rc && ix->d_sa) rc = make_ctx(ix);
    if (rc) { gm_index_free(ix); return rc; }
    *out = ix;
    return GM_OK;
}

int gm_index_import(const uint8_t* bwt_fwd, const uint8_t* bwt_rev, const void* sa_fwd, uint32_t sa_entry_bytes, const uint8_t* codes, const uint64_t* seq_len,
                    uint32_t n_seq, uint32_t sampling, uint32_t block_bytes, int device, gm_index** out)
{
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
    if (!rc && sa_fwd && sa_entry_bytes != (ix->wide ? 8u : 4u)) {
        set_error("sa_fwd entries of %u bytes, but this index has %s rows", sa_entry_bytes, ix->wide ? "64-bit" : "32-bit"); rc = GM_ERR_BAD_ARG;
    }
    if (!rc && sa_fwd && sampling == 1) {
        const size_t rb = ix->wide ? 8 : 4;
        if (hipMalloc(&ix->d_sa, ix->nRows * rb) != hipSuccess) rc = GM_ERR_OOM;
        else if (hipMemcpy(ix->d_sa, sa_fwd, ix->nRows * rb, hipMemcpyHostToDevice) != hipSuccess) rc = GM_ERR_HIP;
        if (!rc) rc = make_sentinel_text(ix);
        if (!rc) rc = make_ctx(ix);
    }
    if (!rc && sa_fwd && sampling > 1) {   // the caller holds the full array: sample it on the device
        void* d_full = nullptr;
        const size_t rb = ix->wide ? 8 : 4;
        if (hipMalloc(&d_full, ix->nRows * rb) != hipSuccess) rc = GM_ERR_OOM;
        else if (hipMemcpy(d_full, sa_fwd, ix->nRows * rb, hipMemcpyHostToDevice) != hipSuccess) rc = GM_ERR_HIP;
        if (!rc) rc = sample_sa(ix, d_full);
        hipFree(d_full);
    }
    if (rc) { gm_index_free(ix); return rc; }
    *out = ix;
    return GM_OK;
}

int gm_index_import_sampled(const uint8_t* bwt_fwd, const uint8_t* bwt_rev, const uint32_t* mark_words, const void* samples, uint32_t sa_entry_bytes, uint64_t n_samples,
                            const uint8_t* codes, const uint64_t* seq_len, uint32_t n_seq, uint32_t sampling, uint32_t block_bytes, int device, gm_index** out)
{
    if (!mark_words || !samples || sampling < 2) { set_error("gm_index_import_sampled: marks, samples and a sampling rate of 2..64"); return GM_ERR_BAD_ARG; }
    gm_index* ix = nullptr;
    int rc = gm_index_import(bwt_fwd, bwt_rev, nullptr, 0, codes, seq_len, n_seq, sampling, block_bytes, device, &ix);
    if (rc) return rc;
    if (sa_entry_bytes != (ix->wide ? 8u : 4u)) { set_error("samples of %u bytes, but this index has %s rows", sa_entry_bytes, ix->wide ? "64-bit" : "32-bit"); gm_index_free(ix); return GM_ERR_BAD_ARG; }
    const uint64_t words = (ix->nRows + 31) / 32;
    // "samples before this word" is recomputed here; a file whose marks and sample count disagree is rejected
    std::vector<uint2> mk(words);
    uint64_t run = 0;
    for (uint64_t w = 0; w < words; ++w) { mk[w] = make_uint2(mark_words[w], (uint32_t)run); run += (uint64_t)__builtin_popcount(mark_words[w]); }
    if (run != n_samples || ((ix->nRows & 31u) && (mark_words[words - 1] >> (ix->nRows & 31u)))) {
        set_error("sampled suffix array: %llu marks for %llu samples", (unsigned long long)run, (unsigned long long)n_samples);
        gm_index_free(ix); return GM_ERR_BAD_ARG;
    }
    if (n_samples >= (1ull << 32)) { set_error("more than 2^32 suffix array samples: use a larger sampling rate"); gm_index_free(ix); return GM_ERR_TOO_LONG; }   // (the mark words count them in 32 bits)
    if (hipMalloc(&ix->d_saMark, words * sizeof(uint2)) != hipSuccess || hipMalloc(&ix->d_saSamples, std::max<uint64_t>(n_samples, 1) * sa_entry_bytes) != hipSuccess) rc = GM_ERR_OOM;
    if (!rc && (hipMemcpy(ix->d_saMark, mk.data(), words * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(ix->d_saSamples, samples, n_samples * sa_entry_bytes, hipMemcpyHostToDevice) != hipSuccess)) rc = GM_ERR_HIP;
    if (rc) { gm_index_free(ix); return rc; }
    ix->nSamples = n_samples;
    *out = ix;
    return GM_OK;
}

int gm_index_export_sa_sampled(const gm_index* ix, uint32_t* mark_words, void* samples, uint32_t sa_entry_bytes, uint64_t* n_samples)
{
    if (!ix || !n_samples) return GM_ERR_BAD_ARG;
    if (!ix->d_saMark) { set_error("index holds no sampled suffix array (sampling %u)", ix->sampling); return GM_ERR_NEED_LOCATE; }
    *n_samples = ix->nSamples;
    if (!samples && !mark_words) return GM_OK;
    if (!samples || !mark_words) return GM_ERR_BAD_ARG;
    if (sa_entry_bytes != (ix->wide ? 8u : 4u)) { set_error("samples of %u bytes, but this index has %s rows", sa_entry_bytes, ix->wide ? "64-bit" : "32-bit"); return GM_ERR_BAD_ARG; }
    GM_HIP(hipSetDevice(ix->device));
    const uint64_t words = (ix->nRows + 31) / 32;
    std::vector<uint2> mk(words);
    GM_HIP(hipMemcpy(mk.data(), ix->d_saMark, words * sizeof(uint2), hipMemcpyDeviceToHost));
    for (uint64_t w = 0; w < words; ++w) mark_words[w] = mk[w].x;
    GM_HIP(hipMemcpy(samples, ix->d_saSamples, ix->nSamples * sa_entry_bytes, hipMemcpyDeviceToHost));
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
            case 2: hipLaunchKernelGGL(unpack_blocks_kernel<2>, dim3(grid_for(ix->nRows)), dim3(256), 0, 0, ix->d_blk[d], ix->nRows, d_bwt); break;
            case 3: hipLaunchKernelGGL(unpack_blocks_kernel<3>, dim3(grid_for(ix->nRows)), dim3(256), 0, 0, ix->d_blk[d], ix->nRows, d_bwt); break;
            default: hipLaunchKernelGGL(unpack_blocks_kernel<9>, dim3(grid_for(ix->nRows)), dim3(256), 0, 0, ix->d_blk[d], ix->nRows, d_bwt); break;
        }
        GM_HIP(hipMemcpy(d ? bwt_rev : bwt_fwd, d_bwt, ix->nRows, hipMemcpyDeviceToHost));
    }
    hipFree(d_bwt);
    return GM_OK;
}

int gm_index_export_sa(const gm_index* ix, void* sa, uint32_t sa_entry_bytes)
{
    if (!ix || !sa) return GM_ERR_BAD_ARG;
    if (sa_entry_bytes != (ix->wide ? 8u : 4u)) { set_error("sa entries of %u bytes, but this index has %s rows", sa_entry_bytes, ix->wide ? "64-bit" : "32-bit"); return GM_ERR_BAD_ARG; }
    if (!ix->d_sa) { set_error("index holds no full suffix array (sampling %u)", ix->d_saMark ? ix->sampling : 0u); return GM_ERR_NEED_LOCATE; }
    GM_HIP(hipSetDevice(ix->device));
    GM_HIP(hipMemcpy(sa, ix->d_sa, ix->nRows * (ix->wide ? 8 : 4), hipMemcpyDeviceToHost));
    return GM_OK;
}

int gm_index_get_info(const gm_index* ix, gm_index_info* info)
{
    if (!ix || !info) return GM_ERR_BAD_ARG;
    info->n_rows = ix->nRows; info->text_len = ix->textLen; info->n_seq = ix->nSeq; info->sampling = ix->sampling;
    info->alphabet_size = ix->alphabet;
    info->block_bytes = ix->wpp == 1 ? 32 : (ix->wpp == 3 || ix->wpp == 2) ? 64 : 128;
    info->row_bits = ix->wide ? 64 : 32;
    info->device_bytes = 2 * ix->blkBytes + ix->textLen + (ix->nSeq + 1) * 8ull + (ix->d_sa ? ix->nRows * (ix->wide ? 9ull : 5ull) : 0ull) + (ix->d_ctx ? ix->nRows * 32ull : 0ull) + ix->qtableBytes
                       + (ix->d_saMark ? (ix->nRows + 31) / 32 * 8ull + ix->nSamples * (ix->wide ? 8ull : 4ull) : 0ull)
                       + ix->shardOutCap;   // the result buffer gm_map / gm_map_shard keep between calls
    if (!ix->d_sa && !ix->d_saMark) info->sampling = 0;
    info->device = ix->device;
    info->verify_records = ix->d_ctx ? 1u : 0u;
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

enum LeafMode { LEAF_COUNT = 0, LEAF_FILESET = 1, LEAF_OCC_COUNT = 2, LEAF_OCC_EMIT = 3, LEAF_STORE = 4, LEAF_STORE8 = 5, LEAF_COUNT_JUMP = 6, LEAF_SCATTER = 7, LEAF_COUNT_NODES = 8 };

// nu = 16-byte units per stored node / queue entry: 1, or 2 with 64-bit rows (gm_kernels.h: NodeIO)
static inline size_t search_lds_bytes(const SearchArgs& A, uint32_t nu) { return (size_t)A.ldsPad + (size_t)(4u * A.vqCap * nu + 4u * 64u * (A.ldsDepth * nu + A.winChunks)) * 16u + 4u * 80u * 4u + 448u + (A.entrySlots ? 4096u : 0u) + (A.lqCap ? 4u * (A.lqCap * 16u + 80u * 4u) : 0u); }

template <int WPP, class EnvT>
static int launch_one(const SearchArgs& A, unsigned blocks, hipStream_t st)
{
    const size_t lds = search_lds_bytes(A, NodeIO<typename BlockGeom<WPP>::row_t>::NU);
    // 32- and 64-byte blocks with 32-bit rows are compiled for 4 waves per SIMD (at most 128 VGPRs: the kernels sit within a
    // register or two of that limit); the 128-byte and the wide geometries take what they need.  Cooperative reads of the rank
    // blocks (groups of 2 / 4 lanes) exist for the 32- and 64-byte blocks and for the wide geometry's 64-byte blocks.
    constexpr bool W4 = WPP == 1 || WPP == 3 || (WPP == 2 && EnvT::EXACT_ONLY);   // (wide rows: the e = 0 kernels gain 5 % capped at 128 VGPRs, 3 spilled; the counting kernels lose 5 %: profiles/r04/final/wide_rows_smoke_w4.txt)
    constexpr bool CAN_COOP = W4 || WPP == 2;
    const void* fn;
    if constexpr (W4) fn = A.coop ? reinterpret_cast<const void*>(&search_kernel_w4<WPP, EnvT, true>) : reinterpret_cast<const void*>(&search_kernel_w4<WPP, EnvT, false>);
    else if constexpr (CAN_COOP) fn = A.coop ? reinterpret_cast<const void*>(&search_kernel<WPP, EnvT, true>) : reinterpret_cast<const void*>(&search_kernel<WPP, EnvT, false>);
    else fn = reinterpret_cast<const void*>(&search_kernel<WPP, EnvT, false>);
    if (lds > 65536)   // long needle windows (K + n - 1 up to 509 symbols per lane): ask for more than the default 64 KB of LDS
        GM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if constexpr (W4) {
        if (A.coop) hipLaunchKernelGGL((search_kernel_w4<WPP, EnvT, true>), dim3(blocks), dim3(256), lds, st, A);
        else hipLaunchKernelGGL((search_kernel_w4<WPP, EnvT, false>), dim3(blocks), dim3(256), lds, st, A);
    } else if constexpr (CAN_COOP) {
        if (A.coop) hipLaunchKernelGGL((search_kernel<WPP, EnvT, true>), dim3(blocks), dim3(256), lds, st, A);
        else hipLaunchKernelGGL((search_kernel<WPP, EnvT, false>), dim3(blocks), dim3(256), lds, st, A);
    } else hipLaunchKernelGGL((search_kernel<WPP, EnvT, false>), dim3(blocks), dim3(256), lds, st, A);
    GM_HIP(hipGetLastError());
    return GM_OK;
}
template <int WPP>
static int launch_mode(int mode, const SearchArgs& A, unsigned blocks, hipStream_t st)
{
    switch (mode) {
        case LEAF_COUNT: return launch_one<WPP, CountEnv<WPP>>(A, blocks, st);
        case LEAF_COUNT_JUMP: return launch_one<WPP, CountEnv<WPP, 1>>(A, blocks, st);
        case LEAF_COUNT_NODES:   // the walker of the split search (32-bit rows: a packet holds a 16-byte node)
            if constexpr (WPP != 2) return launch_one<WPP, CountEnv<WPP, 2>>(A, blocks, st);
            else { set_error("internal: the split search has no 64-bit-row walker"); return GM_ERR_INTERNAL; }
        case LEAF_SCATTER: return launch_one<WPP, ScatterEnv<WPP>>(A, blocks, st);
        case LEAF_FILESET: return launch_one<WPP, FileSetEnv<WPP>>(A, blocks, st);
        case LEAF_OCC_COUNT: return launch_one<WPP, OccCountEnv<WPP>>(A, blocks, st);
        case LEAF_STORE: return launch_one<WPP, StoreEnv<WPP, uint16_t>>(A, blocks, st);
        case LEAF_STORE8: return launch_one<WPP, StoreEnv<WPP, uint8_t>>(A, blocks, st);
        default: return launch_one<WPP, OccEmitEnv<WPP>>(A, blocks, st);
    }
}
// k-mers longer than MAX_K: gm_longk.h (one leaf policy each for frequency, --exclude-pseudo and the two csv passes)
template <int WPP>
static int launch_long(int mode, const SearchArgs& A, unsigned blocks, hipStream_t st)
{
    constexpr size_t LDS = 4u * 80u * 4u + 4u * 64u + 64u;   // control words of the (empty) leaf queues, pairing bytes of the work sharing
    switch (mode) {
        case LEAF_FILESET: hipLaunchKernelGGL((longk_kernel<WPP, FileSetEnv<WPP>>), dim3(blocks), dim3(256), LDS, st, A); break;
        case LEAF_OCC_COUNT: hipLaunchKernelGGL((longk_kernel<WPP, OccCountEnv<WPP>>), dim3(blocks), dim3(256), LDS, st, A); break;
        case LEAF_OCC_EMIT: hipLaunchKernelGGL((longk_kernel<WPP, OccEmitEnv<WPP>>), dim3(blocks), dim3(256), LDS, st, A); break;
        case LEAF_COUNT: hipLaunchKernelGGL((longk_kernel<WPP, CountEnv<WPP>>), dim3(blocks), dim3(256), LDS, st, A); break;
        default: set_error("internal: leaf policy %d has no long k-mer kernel", mode); return GM_ERR_INTERNAL;
    }
    GM_HIP(hipGetLastError());
    return GM_OK;
}
static int launch_search(const gm_index* ix, int mode, const SearchArgs& A, unsigned blocks, hipStream_t st)
{
    if (A.K > MAX_K) {
        switch (ix->wpp) {
            case 1: return launch_long<1>(mode, A, blocks, st);
            case 2: return launch_long<2>(mode, A, blocks, st);
            case 3: return launch_long<3>(mode, A, blocks, st);
            default: return launch_long<9>(mode, A, blocks, st);
        }
    }
    switch (ix->wpp) {
        case 1: return launch_mode<1>(mode, A, blocks, st);
        case 2: return launch_mode<2>(mode, A, blocks, st);
        case 3: return launch_mode<3>(mode, A, blocks, st);
        default: return launch_mode<9>(mode, A, blocks, st);
    }
}

template <int WPP> static int occupancy_blocks(int* out, size_t ldsBytes)
{
    int nb = 0;
    if constexpr (WPP == 1 || WPP == 3) GM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (search_kernel_w4<WPP, CountEnv<WPP>, false>), 256, ldsBytes));
    else GM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (search_kernel<WPP, CountEnv<WPP>, false>), 256, ldsBytes));
    *out = nb;
    return GM_OK;
}

static int check_device_error(gm_index* ix);

// table of all q-mers for this index (cached)
// the 2-bit text and its N flags (pack_text2_kernel), made by the first call that stages windows from them
static int ensure_text2(gm_index* ix, hipStream_t st)
{
    if (ix->d_text2) return GM_OK;
    const uint64_t nChunks4 = ix->textLen / 32 + 20, nChunks2 = ix->textLen / 64 + 24, nFlags = nChunks2 / 8 + 2;   // (padding as behind the 4-bit text: a window may be staged past the end)
    if (hipMalloc(&ix->d_text2, nChunks2 * 16) != hipSuccess) { ix->d_text2 = nullptr; (void)hipGetLastError(); return GM_ERR_HIP; }
    if (hipMalloc(&ix->d_nflag, nFlags * 2) != hipSuccess) { hipFree(ix->d_text2); ix->d_text2 = nullptr; ix->d_nflag = nullptr; (void)hipGetLastError(); return GM_ERR_HIP; }
    hipLaunchKernelGGL(pack_text2_kernel, dim3(grid_for(nFlags)), dim3(256), 0, st, ix->d_text4, nChunks4, ix->d_text2, nChunks2, ix->d_nflag, nFlags);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

static int get_qtable(gm_index* ix, uint32_t* qio, const uint4** out)
{
    uint32_t q = *qio;
    {   // an existing table of this or (after an earlier out-of-memory) the next shorter length
        auto it = ix->qtables.find(q);
        if (it != ix->qtables.end()) { *out = it->second; return GM_OK; }
        if (ix->qtableCap && q > ix->qtableCap) { q = ix->qtableCap; it = ix->qtables.find(q); if (it != ix->qtables.end()) { *qio = q; *out = it->second; return GM_OK; } }
    }
    uint4* d = nullptr;
    for (;; --q) {   // shorter prefixes when the device is short of memory (the table is an accelerator, not a requirement)
        if (q == 0) { *qio = 0; *out = nullptr; return GM_OK; }
        size_t freeB = 0, totalB = 0;
        const uint64_t bytes = (1ull << (2 * q)) * sizeof(uint4);
        // (tables up to 4^15 entries leave half of the free memory alone; the 69 GB of 4^16 ask for 16 GiB of slack instead)
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && (q >= 16 ? bytes + (16ull << 30) > freeB : bytes > freeB / 2)) { ix->qtableCap = q - 1; continue; }
        if (hipMalloc(&d, bytes) == hipSuccess) break;
        (void)hipGetLastError();
        ix->qtableCap = q - 1;
    }
    const uint64_t n = 1ull << (2 * q);
    const uint64_t qblocks = (n + 255) / 256;   // one thread per string: up to 2^24 blocks of 256 (HIP refuses 2^32 threads in one grid dimension)
    const dim3 qgrid((unsigned)std::min<uint64_t>(qblocks, 1u << 22), (unsigned)((qblocks + (1u << 22) - 1) >> 22));
    // one-row entries carry the text next to their occurrence when the suffix array is at hand (gm_kernels.h: qmer_table_kernel)
    const uint32_t* nbSa = (!ix->wide && ix->d_sa && ix->d_textS) ? reinterpret_cast<const uint32_t*>(ix->d_sa) : nullptr;
    if (!ix->d_C) { GM_HIP(hipMalloc(&ix->d_C, sizeof(ix->C))); GM_HIP(hipMemcpy(ix->d_C, ix->C, sizeof(ix->C), hipMemcpyHostToDevice)); }
    switch (ix->wpp) {
        case 1: hipLaunchKernelGGL(qmer_table_kernel<1>, qgrid, dim3(256), 0, 0, ix->d_blk[1], ix->d_C, ix->nRows, q, d, nbSa, ix->d_textS); break;
        case 2: hipLaunchKernelGGL(qmer_table_kernel<2>, qgrid, dim3(256), 0, 0, ix->d_blk[1], ix->d_C, ix->nRows, q, d, nbSa, ix->d_textS); break;
        case 3: hipLaunchKernelGGL(qmer_table_kernel<3>, qgrid, dim3(256), 0, 0, ix->d_blk[1], ix->d_C, ix->nRows, q, d, nbSa, ix->d_textS); break;
        default: hipLaunchKernelGGL(qmer_table_kernel<9>, qgrid, dim3(256), 0, 0, ix->d_blk[1], ix->d_C, ix->nRows, q, d, nbSa, ix->d_textS); break;
    }
    {   // (a failed build must not leave 69 GB behind)
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { hipFree(d); GM_HIP(e); }
    }
    ix->qtables[q] = d;
    ix->qtableBytes += n * sizeof(uint4);
    *qio = q;
    *out = d;
    return GM_OK;
}

// The bitmaps of the groups of jump patterns (gm_oss.h), one array of (1 + 16 + 16) x 4^q / 64 words (cached per q):
//   kind 0 "the q-mer occurs" (from the table of all q-mers) | kind 1 "... followed by letters x y", LOW layout, 16 pairs | the same, MID layout
// (kind 1 from the sentinel text; q = 16: 0.5 + 8.6 + 8.6 GB).  *level: 0 = kind 0 only (device short of memory, or no sentinel text),
// 1 = + LOW, 2 = + MID.  *out stays null when not even kind 0 fits: the call then keeps plain pattern lists.
// one thread per word of the kind-0 bitmap of the layout at bit offset sh: its 64 J-mers differ in the layout's three characters
__global__ __launch_bounds__(256) void jbits_layout_kernel(const uint4* __restrict__ tab, uint64_t words, uint32_t sh, unsigned long long* __restrict__ bits)
{
    const uint64_t w = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= words) return;
    const uint64_t idx0 = ((w >> sh) << (sh + 6u)) | (w & ((1ull << sh) - 1ull));
    unsigned long long m = 0ull;
    for (uint32_t v = 0; v < 64u; ++v) { const uint4 t = tab[idx0 | (uint64_t)v << sh]; if ((t.z | (t.w & 0x00FF0000u)) != 0u) m |= 1ull << v; }   // (64-bit rows keep bits 32..39 of the width in byte 2 of .w; 32-bit rows: zero there whenever .z is)
    bits[w] = m;
}

// *extra: kind-0 planes of the layouts 1, 2, .. (gm_oss.h: group_layout_shifts) behind the 1 + 16 x level planes; 0 when the device is short of memory
static int get_jbits(gm_index* ix, uint32_t q, const uint4* tab, const unsigned long long** out, int* level, uint32_t* extra)
{
    *out = nullptr; *level = 0; *extra = 0;
    if (q <= GROUP_SYMS || !tab) return GM_OK;
    auto it = ix->jbits.find(q);
    if (it != ix->jbits.end()) { *out = it->second; *level = ix->jbitsLevel[q]; *extra = ix->jbitsExtra[q]; return GM_OK; }
    const uint64_t n = 1ull << (2 * q), words = n / 64;
    int lv = !ix->d_textS ? 0 : (q >= 2 * GROUP_SYMS ? 2 : 1);
    const std::vector<uint32_t> shifts = group_layout_shifts(q);
    uint32_t ex = shifts.size() > 1 ? (uint32_t)shifts.size() - 1u : 0u;   // kind-0 planes of the layouts beyond LOW: the first thing to go when memory is short
    unsigned long long* d = nullptr;
    for (;;) {
        if (lv < 0) return GM_OK;
        const uint64_t bytes = (1 + 16 * (uint64_t)lv + ex) * words * 8;
        size_t freeB = 0, totalB = 0;
        const bool fits = !(hipMemGetInfo(&freeB, &totalB) == hipSuccess && bytes + (8ull << 30) > freeB && bytes > (1ull << 24));
        if (fits && hipMalloc(&d, bytes) == hipSuccess) break;
        (void)hipGetLastError();
        if (ex) ex = 0; else --lv;
    }
    const uint64_t blocks = (n + 255) / 256;
    const dim3 grid((unsigned)std::min<uint64_t>(blocks, 1u << 22), (unsigned)((blocks + (1u << 22) - 1) >> 22));
    hipLaunchKernelGGL(jbits_kernel, grid, dim3(256), 0, 0, tab, n, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && lv >= 1) {
        e = hipMemset(d + words, 0, 16 * (uint64_t)lv * words * 8);
        if (e == hipSuccess) {
            const uint64_t threads = (ix->nRows + 63) / 64, tb = (threads + 255) / 256;
            const dim3 g2((unsigned)std::min<uint64_t>(tb, 1u << 22), (unsigned)((tb + (1u << 22) - 1) >> 22));
            hipLaunchKernelGGL(jbits1_kernel, g2, dim3(256), 0, 0, ix->d_textS, ix->nRows, q, d + words, lv >= 2 ? d + 17 * words : nullptr, words);
            e = hipGetLastError();
        }
    }
    for (uint32_t k = 0; k < ex && e == hipSuccess; ++k) {
        const uint64_t tb = (words + 255) / 256;
        const dim3 g3((unsigned)std::min<uint64_t>(tb, 1u << 22), (unsigned)((tb + (1u << 22) - 1) >> 22));
        hipLaunchKernelGGL(jbits_layout_kernel, g3, dim3(256), 0, 0, tab, words, shifts[k + 1], d + (1 + 16 * (uint64_t)lv + k) * words);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { hipFree(d); GM_HIP(e); }
    ix->jbits[q] = d; ix->jbitsLevel[q] = lv; ix->jbitsExtra[q] = ex;
    ix->qtableBytes += (1 + 16 * (uint64_t)lv + ex) * words * 8;
    *out = d; *level = lv; *extra = ex;
    return GM_OK;
}

// maximal runs of the letter N in the whole text (needed once per index, by the correction pass of N-less calls)
__global__ __launch_bounds__(256) void n_run_bounds_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t* __restrict__ starts, uint64_t* __restrict__ ends,
                                                            unsigned long long* __restrict__ counts, uint64_t cap)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (text[i] != SYM_N) continue;
        if (i == 0 || text[i - 1] != SYM_N) { const unsigned long long k = atomicAdd(&counts[0], 1ull); if (k < cap) starts[k] = i; }
        if (i + 1 == n || text[i + 1] != SYM_N) { const unsigned long long k = atomicAdd(&counts[1], 1ull); if (k < cap) ends[k] = i + 1; }
    }
}
static int ensure_n_runs(gm_index* ix)
{
    if (ix->nRunsValid) return GM_OK;
    ix->nRuns.clear();
    if (ix->alphabet == 5) {
        unsigned long long* d_cnt = nullptr; uint64_t *d_s = nullptr, *d_e = nullptr;
        int rc = GM_OK;
        unsigned long long cnt[2] = {0, 0};
        if (hipMalloc(&d_cnt, 16) != hipSuccess) return GM_ERR_OOM;
        for (int pass = 0; pass < 2 && !rc; ++pass) {
            const uint64_t cap = pass ? cnt[0] : 0;
            if (pass) { if (cap == 0) break; if (hipMalloc(&d_s, cap * 8) != hipSuccess || hipMalloc(&d_e, cap * 8) != hipSuccess) { rc = GM_ERR_OOM; break; } }
            if (hipMemset(d_cnt, 0, 16) != hipSuccess) { rc = GM_ERR_HIP; break; }
            hipLaunchKernelGGL(n_run_bounds_kernel, dim3(grid_for(ix->textLen)), dim3(256), 0, 0, ix->d_text, ix->textLen, d_s, d_e, d_cnt, cap);
            if (hipGetLastError() != hipSuccess || hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost) != hipSuccess) rc = GM_ERR_HIP;
        }
        if (!rc && cnt[0]) {
            std::vector<uint64_t> hs(cnt[0]), he(cnt[0]);
            if (cnt[0] != cnt[1] || hipMemcpy(hs.data(), d_s, cnt[0] * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(he.data(), d_e, cnt[0] * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = GM_ERR_HIP;
            else {
                std::sort(hs.begin(), hs.end()); std::sort(he.begin(), he.end());
                for (size_t i = 0; i < hs.size(); ++i) ix->nRuns.emplace_back(hs[i], he[i]);
            }
        }
        hipFree(d_cnt); hipFree(d_s); hipFree(d_e);
        if (rc) { set_error("cannot list the runs of N of the text"); return rc; }
    }
    ix->nRunsValid = true;
    return GM_OK;
}
// block list of the correction pass: the text windows with 1..E letters N of a run (gm_host.h: n_window_intervals), cached per (K, E, infix)
static int ensure_correction_blocks(gm_index* ix, uint32_t K, uint32_t E, uint32_t infix, hipStream_t st)
{
    if (ix->corrValid && ix->corrK == K && ix->corrE == E && ix->corrInfix == infix) return GM_OK;
    int rc = ensure_n_runs(ix); if (rc) return rc;
    ix->corrValid = false; ix->nCBlocks = 0;
    std::vector<uint64_t> iv;
    n_window_intervals(ix->nRuns, ix->cum, K, E, iv);
    if (!iv.empty()) {
        MapPlan cp;
        rc = make_map_plan(K, E, infix, 1, ix->textLen, iv.data(), iv.size() / 2, &cp);
        if (rc) return rc;
        if (!cp.blocks.empty()) {
            rc = grow(&ix->d_cblocks, &ix->cblocksCap, (uint64_t)cp.blocks.size()); if (rc) return rc;
            // on the call's stream, which already waits for the end of the previous call on this index (prepare_search): a correction
            // kernel of that call, on another stream, may still be reading the list this one replaces
            GM_HIP(hipMemcpyAsync(ix->d_cblocks, cp.blocks.data(), cp.blocks.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
            GM_HIP(hipStreamSynchronize(st));   // the host list goes out of scope
            ix->nCBlocks = cp.blocks.size();
        }
    }
    ix->corrK = K; ix->corrE = E; ix->corrInfix = infix; ix->corrValid = true;
    return GM_OK;
}

// relative lengths of the OSS blocks (gm_host.h: make_map_plan).  The scheme is exact for any positive lengths; the reference
// splits evenly.  e = 2: searches 2 and 3 start with ONE exact block (the third / the fourth): blocks in the proportion 5,4,7,8
// instead of 6,6,6,6 make their free-branching part shorter at the price of a shorter exact start of search 1 --
// 3.09 Gbp, K = 30: 5,5,7,7 +17 % without and +20 % with jump patterns, 5,4,7,8 another +8 % (profiles/r03/sweep_oss_weights_e2*.txt, sweep_shapes.txt).
static uint32_t oss_weights_for(const gm_index* ix, uint32_t E)
{
    if (ix->tune.ossWeights >= 0) return (uint32_t)ix->tune.ossWeights;
    // e = 2, four blocks left to right: 5,4,8,8 of an infix of 25 (K=30: 432 ms against 445 with 6,4,7,8 and 482 with 6,6,6,7 on
    // 3 % of the 3.09 Gbp text, profiles/r03/sweep_shapes_j16.txt)
    return E == 2 ? 0x8845u : 0u;
}

struct SearchSetup {
    MapPlan plan;
    ChunkSel sel{0, 0, 0};   // interleaved chunks: which positions of [posBase, posEnd) this call owns
    uint64_t blockBegin = 0, blockEnd = 0, numRoots = 0, kmers = 0;
    unsigned blocks = 1;
    uint64_t posBase = 0, posEnd = 0;   // slice positions covered by the selected blocks: [posBase, posEnd)
    bool jump = false;                  // the call runs the N-less kernel (with jump patterns where they apply) + the correction pass
    // the split search (gm_expand.h): phase A enumerates the jump patterns of the call's blocks into node packets, the walker draws packets
    bool expand = false, overlap = false;
    uint32_t exactItem = 0, rootsPerBlock = 0;   // two passes of phase A: the first takes one item per root (gm_expand.h: expand_strip_exact)
    uint32_t itemsPerBlock = 0, expandBlocks = 0, pktChunks = 0;
    uint32_t rootWinChunks = 0;         // LDS chunks per lane of a kernel that stages the windows of ROOTS (any alignment): the correction pass beside a walker
    uint64_t numBlocksCall = 0, totalChunks = 0;
};

// validation, planning, workspace, uploads; fills every SearchArgs field that does not depend on the leaf policy
static int prepare_search(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
                          const uint64_t* intervals, uint64_t n_intervals, hipStream_t st, SearchSetup* S, SearchArgs* Aout, bool wantJump = false, uint32_t lqCap = 0)
{
    if (!ix || !p) { set_error("null argument"); return GM_ERR_BAD_ARG; }
    if (p->value_bits != 8 && p->value_bits != 16) return GM_ERR_BAD_VALUE_BITS;
    if (text_begin + text_len > ix->textLen || (uint64_t)first_seq + n_seq > ix->nSeq || n_seq == 0) { set_error("slice outside the index"); return GM_ERR_BAD_ARG; }
    if (ix->cum[first_seq] != text_begin || ix->cum[first_seq + n_seq] != text_begin + text_len) { set_error("slice does not match its sequences"); return GM_ERR_BAD_ARG; }
    if (p->E > MAX_ERRORS) return GM_ERR_BAD_ERRORS;
    if (n_intervals > 0 && !intervals) { set_error("n_intervals > 0 with a null interval list"); return GM_ERR_BAD_ARG; }
    GM_HIP(hipSetDevice(ix->device));
    // the workspaces of the index (work counter, accumulators, stacks, per-call tables) are shared by all calls: a call
    // issued on another stream than its predecessor waits for the predecessor's end-of-call event
    if (ix->doneValid) GM_HIP(hipStreamWaitEvent(st, ix->evDone, 0));

    if (p->K < 1 || p->K > MAX_K_LONG) return GM_ERR_BAD_K;
    // k-mers longer than MAX_K: the plain tree walk of gm_longk.h (no tables, no verification, no jump patterns, no LDS staging)
    const bool longK = p->K > MAX_K;
    if (longK) wantJump = false;
    // (frequency calls ask for jump patterns; --exclude-pseudo and gm_locate do not: gm_tuned_infix_length_locating)
    const uint32_t infix = long_k_infix(p->K, p->infix > 0 ? (uint32_t)p->infix : (p->overlap >= 0 ? default_infix_length(p->K, p->E, p->overlap) : tuned_infix_length(p->K, p->E, !wantJump)));
    if (infix == 0) return GM_ERR_BAD_OVERLAP;
    MapPlan& plan = S->plan;
    int rc = make_map_plan(p->K, p->E, infix, p->revcompl, text_len, intervals, n_intervals, &plan, ix->tune.partBias, oss_weights_for(ix, p->E));
    if (rc) return rc;   // PlanError values coincide with gm_status

    // shard [kmer_begin, kmer_end): blocks whose first k-mer lies inside
    uint64_t blockBegin = 0, blockEnd = plan.numBlocks;
    if ((p->flags & (GM_MAP_FLAG_RANGE | GM_MAP_FLAG_PIECE)) || p->kmer_begin != 0 || p->kmer_end != 0) {
        if (plan.useList) {
            auto lo = std::lower_bound(plan.blocks.begin(), plan.blocks.end(), p->kmer_begin, [](const std::pair<uint32_t, uint32_t>& b, uint64_t v) { return MapPlan::block_pos(b) < v; });
            auto hi = std::lower_bound(plan.blocks.begin(), plan.blocks.end(), p->kmer_end, [](const std::pair<uint32_t, uint32_t>& b, uint64_t v) { return MapPlan::block_pos(b) < v; });
            blockBegin = lo - plan.blocks.begin(); blockEnd = hi - plan.blocks.begin();
        } else {
            blockBegin = std::min<uint64_t>((p->kmer_begin + plan.stepSize - 1) / plan.stepSize, plan.numBlocks);
            blockEnd = std::min<uint64_t>((p->kmer_end + plan.stepSize - 1) / plan.stepSize, plan.numBlocks);
        }
        if (blockEnd < blockBegin) blockEnd = blockBegin;
    }
    const uint32_t rpb = plan.nSearches * plan.nStrands;
    // interleaved chunks of whole blocks: this call owns the chunks c = chunk_index (mod chunk_stride) of the range
    const bool chunked = p->chunk_blocks > 0 && p->chunk_stride > 1;
    uint64_t myBlocks = blockEnd - blockBegin;
    if (chunked) {
        if (p->chunk_index >= p->chunk_stride) { set_error("chunk_index >= chunk_stride"); return GM_ERR_BAD_ARG; }
        if (plan.useList) { set_error("interleaved chunks and a selection cannot be combined (shard a selection with kmer_begin/kmer_end)"); return GM_ERR_BAD_ARG; }
        const uint64_t T = blockEnd - blockBegin, cb = p->chunk_blocks, full = T / cb, rem = T % cb;
        myBlocks = (full > p->chunk_index ? (full - p->chunk_index + p->chunk_stride - 1) / p->chunk_stride : 0) * cb;
        if (rem && full % p->chunk_stride == p->chunk_index) myBlocks += rem;
        if ((uint64_t)cb * plan.stepSize >= (1ull << 32)) { set_error("chunk too long"); return GM_ERR_BAD_ARG; }
        S->sel = ChunkSel{(uint32_t)(cb * plan.stepSize), p->chunk_stride, p->chunk_index};
    }
    S->blockBegin = blockBegin; S->blockEnd = blockEnd; S->numRoots = myBlocks * rpb;
    uint64_t kmers = 0;
    S->posBase = S->posEnd = 0;
    if (blockEnd > blockBegin) {
        if (plan.useList) {
            for (uint64_t b = blockBegin; b < blockEnd; ++b) kmers += MapPlan::block_n(plan.blocks[b]);
            S->posBase = MapPlan::block_pos(plan.blocks[blockBegin]); S->posEnd = MapPlan::block_pos(plan.blocks[blockEnd - 1]) + MapPlan::block_n(plan.blocks[blockEnd - 1]);
        } else {
            S->posBase = blockBegin * plan.stepSize; S->posEnd = std::min<uint64_t>(blockEnd * plan.stepSize, plan.numKmers);
            kmers = S->posEnd - S->posBase;
            if (chunked) {   // k-mers of the own chunks (only the very last block of the text can be short)
                kmers = myBlocks * plan.stepSize;
                const uint64_t lastBlock = plan.numBlocks - 1, cb = p->chunk_blocks;
                if (blockEnd == plan.numBlocks && ((lastBlock - blockBegin) / cb) % p->chunk_stride == p->chunk_index)
                    kmers -= (uint64_t)plan.numBlocks * plan.stepSize - plan.numKmers;
            }
        }
    }
    S->kmers = kmers;

    // ---- workspace ----
    {
        const void *t0 = ix->d_table, *c0 = ix->d_cumLocal, *b0 = ix->d_blocks;
        rc = grow(&ix->d_table, &ix->tableCap, (uint64_t)plan.table.size()); if (rc) return rc;
        if (longK) { const void* l0 = ix->d_tableL; rc = grow(&ix->d_tableL, &ix->tableLCap, (uint64_t)plan.tableL.size()); if (rc) return rc; if (l0 != ix->d_tableL) ix->sigValid = false; }
        rc = grow(&ix->d_cumLocal, &ix->cumLocalCap, (uint64_t)n_seq + 1); if (rc) return rc;
        if (plan.useList) { rc = grow(&ix->d_blocks, &ix->blocksCap, std::max<uint64_t>(plan.blocks.size(), 1)); if (rc) return rc; }
        if (t0 != ix->d_table || c0 != ix->d_cumLocal || b0 != ix->d_blocks) ix->sigValid = false;   // reallocated: contents are gone
    }
    // (calls that may jump keep their table entries in flight in LDS: one 16-byte slot per lane)
    // (64-bit rows jump too since round 5: plain pattern lists -- no bitmaps, no neighbour filter, the table entry travels in registers)
    const bool mayJump = wantJump && p->E >= 1 && (ix->d_sa || ix->d_saMark) && ix->tune.jump != 0;
    const uint32_t nu = ix->wide ? 2u : 1u;
    // ---- jump patterns (frequency calls with errors on an index that can locate): one J for every search ----
    std::vector<uint32_t> patHost; std::vector<uint4> jinfoHost; uint32_t jumpJ = 0, jumpAPacked[2] = {0, 0};
    uint32_t layShift[8] = {0, 0, 0, 0, 0, 0, 0, 0}, layPlane0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, layPlane1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long* jbitsCall = nullptr; unsigned long long gmaskCall[GROUP_MAX_MASKS] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t jbitsWords = 0; std::vector<uint4> jinfo2Host(8, make_uint4(0, 0, 0, 0));
    const uint4* jtab = nullptr;
    uint32_t firstItem[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nItems[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool hasExact[8] = {false, false, false, false, false, false, false, false};   // the search's patterns include the one without a substitution
    std::vector<uint32_t> wmapHost;
    S->jump = mayJump;
    if (S->jump) {
        const uint32_t L = plan.infix;
        uint32_t J = 1;   // longest tabulated string: as for the q-mer tables below
        // (3.09 Gbp, 16 against 15 characters: K=30 e=1 -10 %, e=2 -4.7 %, K=100 e=1 -6.5 % kernel time, profiles/r03/sweep_qtable16.txt)
        while (J < (ix->nRows > (1ull << 30) ? 16u : 15u) && (1ull << (2 * J)) < 4ull * ix->nRows) ++J;
        if (ix->tune.jump > 0) J = (uint32_t)ix->tune.jump;   // forced (sweeps, tests)
        if (ix->tune.qtable >= 0) J = std::min<uint32_t>(J, (uint32_t)ix->tune.qtable);
        J = L >= 2 ? std::min(J, L - 1u) : 0u;
        if (J) { rc = get_qtable(ix, &J, &jtab); if (rc) return rc; }   // (shorter when the device is short of memory)
        std::vector<JumpSearch> js(plan.nSearches);
        for (; J >= 1; --J) {
            bool ok = true; size_t total = 0;
            for (uint32_t s2 = 0; s2 < plan.nSearches && ok; ++s2) {
                ok = oss_jump_patterns(p->E, plan.table[(size_t)(plan.stepSize - 1) * 8 + s2], L, J, 4096, &js[s2]) && !js[s2].pat.empty();
                total += js[s2].pat.size();
            }
            if (ok && total < 65000) break;
        }
        if (J >= 1) { uint32_t J2 = J; rc = get_qtable(ix, &J2, &jtab); if (rc) return rc; if (J2 != J) J = 0; }
        if (J >= 1) {
            jumpJ = J;
            jinfoHost.assign(8, make_uint4(0, 0, 0, 0));
            // Groups of patterns (gm_oss.h): patterns that differ in the last three characters only share one word of the existence bitmap.
            // A search is grouped when that saves table reads: groups + (share of J-mers that occur) x their patterns against one read
            // per pattern (3.09 Gbp, J = 16: 51 % occur; K = 30 e = 2, search 1: 211 reads -> 13 words + 54 + ~80 reads).
            const unsigned long long* jbits = nullptr; int jbLevel = 0; uint32_t jbExtra = 0;
            if (ix->tune.jumpGroups != 0) { rc = get_jbits(ix, J, jtab, &jbits, &jbLevel, &jbExtra); if (rc) return rc; }
            // the call's layouts: bit offsets, the plane of each layout's kind-0 bitmap, the first plane of its kind-1 bitmaps
            const std::vector<uint32_t> shifts = group_layout_shifts(J);
            uint32_t kind0Layouts = jbits ? 1u : 0u;
            for (uint32_t L = 0; L < GROUP_MAX_LAYOUTS; ++L) { layShift[L] = L < shifts.size() ? shifts[L] : 0u; layPlane0[L] = 0u; layPlane1[L] = 0u; }
            if (jbLevel >= 1) layPlane1[0] = 1u;
            if (jbLevel >= 2) layPlane1[1] = 17u;
            for (uint32_t L = 1; L < shifts.size() && L - 1u < jbExtra && ix->tune.jumpLayouts != 0; ++L) { layPlane0[L] = 1u + 16u * (uint32_t)jbLevel + (L - 1u); kind0Layouts |= 1u << L; }
            const double occur = 1.0 - std::exp(-(double)ix->nRows / std::ldexp(1.0, 2 * (int)J));
            const double occur1 = 1.0 - std::exp(-(double)ix->nRows / std::ldexp(1.0, 2 * (int)J + 4));
            std::vector<uint64_t> masks;
            std::vector<SearchItems> items(plan.nSearches);
            jinfo2Host.assign(8, make_uint4(0, 0, 0, 0));
            for (uint32_t s2 = 0; s2 < plan.nSearches; ++s2) {
                // kind 1 needs two more infix characters to the right of the J-mer (and, for MID groups, its second bitmap family)
                const bool ext = jbLevel >= (J >= 2 * GROUP_SYMS ? 2 : 1) && js[s2].regionA + J + 2u <= L;
                // which groups are formed: where the occurrence estimate expects fewer table reads (e <= 1), wherever two patterns share a word (e >= 2:
                // a dead table read also costs its lane a turn of the loop -- 3.09 Gbp K=30 e=2: 254.6 ms against 273.0 with the estimate,
                // e=1 287.8 / 288.8, K=100 e=1 226.9 / 224.9; profiles/r05/sweep_layouts.txt)
                const int gmode = !jbits ? 0 : ix->tune.jumpGroups > 0 ? 1 : (p->E >= 2 ? 1 : 2);
                oss_make_items(js[s2], p->E, gmode, ext, occur, occur1, &masks, &items[s2], kind0Layouts);
            }
            jbitsCall = masks.empty() ? nullptr : jbits; jbitsWords = (1ull << (2 * J)) / 64;
            for (size_t k = 0; k < masks.size(); ++k) gmaskCall[k] = masks[k];
            for (uint32_t s2 = 0; s2 < plan.nSearches; ++s2) {
                // neighbour filter (gm_kernels.h): how many infix characters right / left of the J-mer a one-row table entry is compared with
                uint32_t nbWord = 0;
                if (ix->tune.jumpFilter != 0 && !ix->wide && ix->d_sa && ix->d_textS) {
                    const uint32_t nr = std::min<uint32_t>(NB_SYMS, L - js[s2].regionA - J), nl = std::min<uint32_t>(NB_SYMS, js[s2].regionA);
                    for (uint32_t i = 0; i < nr; ++i) nbWord |= 1u << (2u * i);
                    for (uint32_t i = 0; i < nl; ++i) nbWord |= 1u << (16u + 2u * i);
                    nbWord |= nr << 12 | nl << 28 | 1u << 31;
                }
                jinfoHost[s2] = make_uint4((uint32_t)patHost.size() | (uint32_t)items[s2].items.size() << 16, js[s2].meta0, items[s2].items[0], nbWord);
                {   // where the groups of each layout end among the call's items (16 bits each)
                    uint32_t e16[GROUP_MAX_LAYOUTS], run = (uint32_t)patHost.size();
                    for (uint32_t L = 0; L < GROUP_MAX_LAYOUTS; ++L) { run += L < items[s2].seg.size() ? items[s2].seg[L] : 0u; e16[L] = run; }
                    jinfo2Host[s2] = make_uint4(e16[0] | e16[1] << 16, e16[2] | e16[3] << 16, e16[4] | e16[5] << 16, items[s2].ext ? 1u : 0u);
                }
                jumpAPacked[s2 >> 2] |= js[s2].regionA << (8u * (s2 & 3u));
                firstItem[s2] = (uint32_t)patHost.size(); nItems[s2] = (uint32_t)items[s2].items.size();
                for (uint32_t d : js[s2].pat) hasExact[s2] = hasExact[s2] || (d & 7u) == 0u;
                patHost.insert(patHost.end(), items[s2].items.begin(), items[s2].items.end());
            }
            const void *p0 = ix->d_patterns, *j0 = ix->d_jinfo;
            rc = grow(&ix->d_patterns, &ix->patternsCap, (uint64_t)patHost.size()); if (rc) return rc;
            rc = grow(&ix->d_jinfo, &ix->jinfoCap, 8); if (rc) return rc;
            if (!ix->d_jinfo2) { GM_HIP(hipMalloc(&ix->d_jinfo2, 8 * sizeof(uint4))); ix->sigValid = false; }
            if (p0 != ix->d_patterns || j0 != ix->d_jinfo) ix->sigValid = false;   // reallocated: contents are gone
        }
    }
    // ---- the split search (gm_expand.h): one lane per (root, item) enumerates the patterns into node packets, the walker draws packets.
    // 32-bit rows (a packet holds a 16-byte node); calls large enough that the extra launches do not show.
    S->expand = false;
    S->numBlocksCall = rpb ? S->numRoots / rpb : 0;
    if (jumpJ >= 1 && !ix->wide && !longK && S->numBlocksCall > 0 && patHost.size() < (1u << 24) &&
        (ix->tune.expand > 0 || (ix->tune.expand < 0 && S->numRoots >= (1ull << 20) && p->K < 64u))) {   // (K=100 e=1 has 0.16 table reads per k-mer: nothing to split off -- its walker measured 150 ms against the 120 ms of the one-loop kernel on 30 % of 3.09 Gbp)
        wmapHost = make_wmap(plan.nStrands, plan.nSearches, firstItem, nItems);
        S->itemsPerBlock = (uint32_t)wmapHost.size();
        // two passes: the work items of the first -- one per root, the pattern without a substitution: a plain item (rotation word 0) behind the
        // items of every search, so that no layout claims it (gm_oss.h: item_layout) -- follow the work map of the second
        S->exactItem = (uint32_t)patHost.size();
        patHost.push_back(0u);
        for (uint32_t st2 = 0; st2 < plan.nStrands; ++st2) for (uint32_t s2 = 0; s2 < plan.nSearches; ++s2) wmapHost.push_back(wmap_pack(s2, st2, S->exactItem) | (hasExact[s2] ? 0u : WMAP_ROOT_ONLY));
        S->rootsPerBlock = plan.nStrands * plan.nSearches;
        { const void* p0 = ix->d_patterns; rc = grow(&ix->d_patterns, &ix->patternsCap, (uint64_t)patHost.size()); if (rc) return rc; if (p0 != ix->d_patterns) ix->sigValid = false; }
        S->expandBlocks = ix->tune.expandChunk > 0 ? (uint32_t)ix->tune.expandChunk : std::max<uint32_t>(1u, 8192u / std::max<uint32_t>(S->itemsPerBlock, 1u));   // (about 8192 work items: the chunk counter is ONE address, good for ~15 M returning atomics a second -- with 1024 items per chunk it was the limit of phase A; the packets of a chunk are neighbours in the lists, and long runs of neighbours make the walker's pools uneven)
        S->expandBlocks = std::min<uint32_t>(S->expandBlocks, (1u << 22) / std::max<uint32_t>(S->itemsPerBlock, 1u));   // (work items of a chunk are numbered in 32 bits)
        S->totalChunks = (S->numBlocksCall + S->expandBlocks - 1) / S->expandBlocks;
        S->pktChunks = pkt_chunks_for(p->K, plan.stepSize);
        S->expand = S->itemsPerBlock > 0 && S->expandBlocks > 0 && S->totalChunks < 0xFFFFFFF0ull;
        if (S->expand) {
            const void* w0 = ix->d_wmap;
            rc = grow(&ix->d_wmap, &ix->wmapCap, (uint64_t)wmapHost.size()); if (rc) return rc;
            if (w0 != ix->d_wmap) ix->sigValid = false;
        }
    }
    const bool expand = S->expand;
    // LDS staging per block of 4 wavefronts: verification queue, top of the lane stacks, packed needle windows
    uint32_t verifyT = 0;
    if (ix->d_sa && ix->d_textS && longK) verifyT = (uint32_t)std::max(0, std::min(ix->tune.verifyT >= 0 ? ix->tune.verifyT : 1, (int)VERIFY_TMAX));   // gm_longk.h: a row costs one suffix-array read and a scan of the text by its lane alone (3.09 Gbp K=300: one row e=0 -53 %, e=1 -7 % over none; four rows +10 % / +28 % over one, profiles/r05/longk_scale.txt)
    if (ix->d_sa && ix->d_textS && !longK) {   // narrow nodes are resolved against the text when the SA is resident
        int t = 1;
        // long k-mers with errors: a two-row node has a long way to go by rank steps; with the 32-byte row records two reads
        // settle it (K=100 e=1: +8 % on 3.09 Gbp, +12 % on 249 Mbp; K=30: -20 %, profiles/r02/sweep_*_steal_verify.txt)
        // (only up to two errors: K=101 e=3 -5.5 %, e=4 -16 % kernel time with one row, profiles/r04/sweep_e3_e4.txt)
        if (p->K >= 64 && p->E >= 1 && p->E <= 2 && ix->d_ctx && ix->tune.useCtx) t = 2;
        else if (plan.stepSize >= 32) t = 4;   // long blocks: a narrow node still covers many k-mers (profiles/r01c)
        if (ix->tune.verifyT >= 0) t = ix->tune.verifyT;
        verifyT = (uint32_t)std::max(0, std::min(t, (int)VERIFY_TMAX));
    }
    // extension-phase nodes (infix complete) may be wider: each row costs one record and two short scans against ~n log n steps
    // (3.09 Gbp K=100 e=1: 4 rows -11 % kernel time over 2, 8 and 16 the same; profiles/r04/sweep_verify_t_ext.txt)
    uint32_t verifyTExt = verifyT;
    if (verifyT && p->E >= 1) {
        int t = 4;   // K=30: e=1 -2 %, e=2 -0.7 % kernel time over 1 (2 rows: no gain)
        if (ix->tune.verifyTExt >= 0) t = ix->tune.verifyTExt;
        verifyTExt = (uint32_t)std::max((int)verifyT, std::min(t, (int)VERIFY_TMAX));
    }
    const uint32_t depth = stack_bound(p->E, plan.stepSize) + STEAL_LEVELS;   // room for the levels work sharing may vacate at the bottom
    uint32_t verifyRows = std::min(verifyTExt, VERIFY_ROWS);   // rows of one node queued per iteration (search_body); one instead of two where that keeps a block per CU (below)
    // queue entries per wavefront: up to 63 left from the last iteration + one row of every lane + (two rows) 32 second rows; the rest waits (search_body)
    auto vq_cap = [](uint32_t rows) { return rows >= 2u ? 160u : 128u; };
    uint32_t vqCap = verifyT ? vq_cap(verifyRows) : 1u;
    // (long k-mers read their needle from the text; the walker of the split search stages the window of its packet, which starts at nibble 0)
    S->rootWinChunks = longK ? 1u : (31u + p->K + plan.stepSize - 1u + 31u) / 32u;
    // windows at 2 bits per symbol: where the chunks they save give LDS stack levels back (K=100 e=1: five chunks -> three, no stack level in LDS -> two);
    // the correction pass, the walker of the split search and long k-mers keep theirs.  3.09 Gbp, kernel time with them against without
    // (profiles/r06/dev/win2_sweep.txt): K=100 e=1 353 / 366 ms, e=0 31.2 / 32.9, K=64 e=1 585 / 596, K=150 e=1 326 / 348, K=250 e=1 (half the k-mers)
    // 206 / 244, K=101 e=4 (0.4 %) 645-672 / 690-716; K=101 e=2 and e=3 the same.  Short windows (K < 64) save a chunk at most and were not measured.
    const uint32_t win2Chunks = (63u + p->K + plan.stepSize - 1u + 63u) / 64u;
    bool win2 = !expand && !longK && (ix->tune.win2 > 0 || (ix->tune.win2 < 0 && p->K >= 64u && S->rootWinChunks > win2Chunks));
    if (win2 && ensure_text2(ix, st) != GM_OK) { if (ix->tune.win2 > 0) { set_error("no device memory for the 2-bit text"); return GM_ERR_HIP; } win2 = false; }   // (no memory for it: the 4-bit windows)
    const uint32_t winChunks = expand ? S->pktChunks : win2 ? win2Chunks : S->rootWinChunks;
    // (the split search runs phase A of the next slice BESIDE the walker: three walker blocks per CU leave the fourth slot -- registers and LDS -- to a block of phase A)
    const bool overlap = expand && ix->tune.expandOverlap > 0;   // (off by default: phase A is not light enough on the vector ALU yet -- 3.09 Gbp K=30 e=2 236 ms side by side against 215 one after the other, profiles/r06)
    S->overlap = overlap;
    const int wantPerCU = std::max(1, overlap ? std::min(ix->tune.blocksPerCU, 3) : ix->tune.blocksPerCU);   // default 4 = 4 waves/SIMD, what the kernel's VGPR count allows
    const bool entrySlots = mayJump && !ix->wide && !expand;
    const uint32_t ldsPad = (uint32_t)std::max(0, ix->tune.ldsPad);   // (measurement: where is the occupancy cliff?)
    auto lds_bytes_for = [&](uint32_t d) { return (size_t)ldsPad + (size_t)(4u * vqCap * nu + 4u * 64u * (d * nu + winChunks)) * 16u + 4u * 80u * 4u + 448u + (entrySlots ? 4096u : 0u) + (lqCap ? 4u * (lqCap * 16u + 80u * 4u) : 0u); };   // == search_lds_bytes
    // Blocks per CU that REALLY become resident: the occupancy query says four blocks of up to 40,960 B fit the 160 KB of a CU, the device
    // runs three of them beyond ~38.6 KB per block (measured with padded launches, 3.09 Gbp: K=30 e=2 244 ms at 36,544 and 37,568 B per
    // block, 250 at 38,592, 279 at 39,616 and 40,640 = the time of three blocks per CU; K=100 e=1 192 / 193 / 220 ms at 36,544 / 38,592 /
    // 40,640 B; profiles/r05/sweep_lds_occupancy_cliff.txt).  Round 4's last change had put K=30 e>=1 and K=100 at 40,640 B.
    constexpr size_t LDS_USABLE_PER_CU = 154624;   // 151 KB
    auto blocks_for = [&](uint32_t d, int* nb) {
        int rc2;
        switch (ix->wpp) { case 1: rc2 = occupancy_blocks<1>(nb, lds_bytes_for(d)); break; case 2: rc2 = occupancy_blocks<2>(nb, lds_bytes_for(d)); break; case 3: rc2 = occupancy_blocks<3>(nb, lds_bytes_for(d)); break; default: rc2 = occupancy_blocks<9>(nb, lds_bytes_for(d)); break; }
        if (!rc2) *nb = std::min<int>(*nb, (int)(LDS_USABLE_PER_CU / std::max<size_t>(lds_bytes_for(d), 1)));
        return rc2;
    };
    // stack levels kept in LDS: four when they fit beside the needle windows and the verification queue at full occupancy;
    // long windows (K >= ~60) trade levels for resident blocks -- a fourth block per CU is worth more than the levels
    // (3.09 Gbp e=1: K=100 676 -> 600 ms, K=150 819 -> 642 ms; at equal occupancy deeper is better; profiles/r02/sweep_grch38_ldsstack.txt)
    uint32_t ldsDepth = 0; int perCU = 0;
    if (verifyRows > 1u && ix->tune.ldsStack < 0) {   // long windows (K >= ~64): a smaller verification queue where it buys the fourth block per CU (K=100 e=1: 220 -> 200 ms)
        int nb2 = 0, nb1 = 0;
        const uint32_t d0 = (p->E >= 3 && nu == 1u) ? std::min(2u, depth) : 0u;   // the fewest LDS stack levels the choice below may end at
        rc = blocks_for(d0, &nb2); if (rc) return rc;
        vqCap = vq_cap(1u); rc = blocks_for(d0, &nb1); if (rc) return rc;
        if (std::min(nb1, wantPerCU) > std::min(nb2, wantPerCU)) verifyRows = 1u; else vqCap = vq_cap(verifyRows);
    }
    if (ix->tune.ldsStack >= 0) {
        ldsDepth = std::min((uint32_t)ix->tune.ldsStack / nu, depth);   // the same LDS for the stack tops of wide nodes
        rc = blocks_for(ldsDepth, &perCU); if (rc) return rc;
        perCU = std::max(1, std::min(perCU, wantPerCU));
    } else {
        // (three and four errors stack deep: two levels in LDS are worth more than the block per CU they may cost -- K=101 e=4 -12 %, e=3 -1.6 %)
        // (no level at all where that is what keeps the fourth block: K=100 e=1 203 ms against 226 with one level and three blocks)
        const uint32_t dmin = (p->E >= 3 && nu == 1u) ? std::min(2u, depth) : 0u;
        for (int d = (int)std::min(4u / nu, depth); d >= (int)dmin; --d) {
            int nb = 0;
            rc = blocks_for((uint32_t)d, &nb); if (rc) return rc;
            nb = std::min(nb, wantPerCU);
            if (nb > perCU) { perCU = nb; ldsDepth = (uint32_t)d; }
        }
        perCU = std::max(1, perCU);
    }
    uint64_t blocks = (uint64_t)ix->numCU * (longK ? 8u : (uint32_t)perCU);   // (gm_longk.h: no LDS to speak of, eight blocks of 256 lanes per CU)
    const uint64_t useful = (S->numRoots + 255) / 256;
    if (blocks > useful) blocks = std::max<uint64_t>(useful, 1);
    S->blocks = (unsigned)blocks;
    if (longK) { ldsDepth = 0; rc = grow(&ix->d_stack, &ix->stackCap, blocks * 256ull * depth * 2ull); if (rc) return rc; }   // LNodeT: at most 32 bytes per entry
    // (twice: the correction pass of an N-less call runs beside the main search with the same geometry and at most as many blocks: the upper half is its)
    rc = grow(&ix->d_stack, &ix->stackCap, 2ull * blocks * 256ull * std::max<uint32_t>(depth - ldsDepth, 1u) * nu); if (rc) return rc;

    {   // the call's small device-side tables (OSS records, block list, local sequence limits) are uploaded only when the
        // call differs from the previous one on this index: a loop over shards or repeated passes launches without any
        // host-device synchronisation
        uint64_t h = 1469598103934665603ull;
        auto mix = [&h](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
        mix(p->K); mix(p->E); mix(plan.infix); mix((uint64_t)(int64_t)ix->tune.partBias); mix((uint64_t)(int64_t)ix->tune.ossWeights); mix(jumpJ); mix((uint64_t)ix->tune.jumpFilter); mix(patHost.size()); for (uint32_t v : patHost) mix(v); mix(wmapHost.size()); for (uint32_t v : wmapHost) mix(v); for (const uint4& v : jinfo2Host) { mix(v.x); mix(v.y); mix(v.z); mix(v.w); } mix(text_begin); mix(text_len); mix(first_seq); mix(n_seq); mix(n_intervals);
        for (uint64_t k = 0; k < 2 * n_intervals; ++k) mix(intervals[k]);
        if (!ix->sigValid || ix->sig != h) {
            GM_HIP(hipMemcpyAsync(ix->d_table, plan.table.data(), plan.table.size() * sizeof(OssRecord), hipMemcpyHostToDevice, st));
            if (longK) GM_HIP(hipMemcpyAsync(ix->d_tableL, plan.tableL.data(), plan.tableL.size() * sizeof(OssRecordL), hipMemcpyHostToDevice, st));
            if (plan.useList && !plan.blocks.empty())
                GM_HIP(hipMemcpyAsync(ix->d_blocks, plan.blocks.data(), plan.blocks.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
            if (jumpJ) {
                GM_HIP(hipMemcpyAsync(ix->d_patterns, patHost.data(), patHost.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                GM_HIP(hipMemcpyAsync(ix->d_jinfo, jinfoHost.data(), 8 * sizeof(uint4), hipMemcpyHostToDevice, st));
                GM_HIP(hipMemcpyAsync(ix->d_jinfo2, jinfo2Host.data(), 8 * sizeof(uint4), hipMemcpyHostToDevice, st));
            }
            if (expand) GM_HIP(hipMemcpyAsync(ix->d_wmap, wmapHost.data(), wmapHost.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            std::vector<uint64_t> cumLocal((size_t)n_seq + 1);
            for (uint32_t s = 0; s <= n_seq; ++s) cumLocal[s] = ix->cum[first_seq + s] - text_begin;
            GM_HIP(hipMemcpyAsync(ix->d_cumLocal, cumLocal.data(), cumLocal.size() * 8, hipMemcpyHostToDevice, st));
            GM_HIP(hipStreamSynchronize(st));   // host staging buffers go out of scope
            ix->sig = h; ix->sigValid = true;
        }
    }

    SearchArgs A; memset(&A, 0, sizeof(A));
    A.blk[0] = ix->d_blk[0]; A.blk[1] = ix->d_blk[1];
    for (uint32_t c = 0; c <= NLET; ++c) A.C[c] = ix->C[c];
    A.nRows = ix->nRows;
    A.text = ix->d_text + text_begin;
    A.K = p->K; A.E = p->E;
    A.stepSize = plan.stepSize; A.nSearches = plan.nSearches; A.rootsPerBlock = rpb;
    A.numKmers = plan.numKmers;
    A.blockBegin = blockBegin; A.numRoots = S->numRoots;
    A.blockList = plan.useList ? ix->d_blocks : nullptr;
    A.table = ix->d_table;
    A.stack = ix->d_stack; A.stackDepth = depth; A.spillDepth = std::max<uint32_t>(depth - ldsDepth, 1u);
    {   // q-mer tables for the first block of every search: q = min(Qmax, length of that block - 1) for the regular block shape
        // Longest tabulated prefix: one more symbol than it takes a random string to become unique in this text
        // (ceil(log4 rows) + 1), at most 15: 4^15 entries x 16 B = 17 GB of the 288 GB -- every tabulated symbol
        // replaces a bidirectional step (1-2 random rank reads) by a share of ONE table read
        // (profiles/r01h_qtable_sweep.txt: e=0 +27 % on 249 Mbp, +30 % on 3.1 Gbp going from 12 to 15).
        // Beyond 2^30 rows the average 15-mer still has ~3 occurrences: 16 symbols (4^16 entries x 16 B = 69 GB) leave most k-mers of a
        // genome with their single row straight from the table (profiles/r03/sweep_qtable16.txt)
        uint32_t qmax = 1;
        const uint32_t qcap = ix->tune.qtable >= 0 ? (uint32_t)ix->tune.qtable : (ix->nRows > (1ull << 30) ? 16u : 15u);
        while (qmax < qcap && (1ull << (2 * qmax)) < 4ull * ix->nRows) ++qmax;
        if (ix->tune.qtable >= 0) qmax = (uint32_t)std::min(ix->tune.qtable, 16);
        A.qtabA = A.qtabB = nullptr; A.qlenPacked[0] = A.qlenPacked[1] = 0; A.qselMask = 0; A.startPacked[0] = A.startPacked[1] = 0;
        uint32_t qA = 0, qB = 0;
        for (uint32_t s = 0; s < plan.nSearches; ++s) {
            const OssRecord& r = plan.table[(size_t)(plan.stepSize - 1) * 8 + s];
            if (!longK) A.startPacked[s >> 2] |= oss_start(r) << (8u * (s & 3u));
            if (qmax == 0) continue;
            // (long k-mers, gm_longk.h: the first block of the REGULAR shape bounds the prefix; shorter blocks at the end of the text have longer infixes)
            const uint32_t bl0 = longK ? oss_bl(plan.tableL[(size_t)(plan.stepSize - 1) * 8 + s], 0) : oss_bl(r, 0);
            uint32_t q = std::min(qmax, bl0 > 0 ? bl0 - 1u : 0u);
            if (longK) for (uint32_t n2 = 1; n2 <= plan.stepSize; ++n2) {   // ... and no shape's first block may be shorter than the prefix + 1 (never seen: checked, not assumed)
                const uint32_t b2 = oss_bl(plan.tableL[(size_t)(n2 - 1) * 8 + s], 0);
                q = std::min(q, b2 > 0 ? b2 - 1u : 0u);
            }
            if (q == 0) continue;
            if (qA == 0 || qA == q) { if (qA == 0) { rc = get_qtable(ix, &q, &A.qtabA); if (rc) return rc; qA = q; } }
            else if (qB == 0 || qB == q) { if (qB == 0) { rc = get_qtable(ix, &q, &A.qtabB); if (rc) return rc; qB = q; } A.qselMask |= 1u << s; }
            else continue;   // a third distinct prefix length: this search starts from the root
            if (q == 0) continue;
            A.qlenPacked[s >> 2] |= q << (8u * (s & 3u));
        }
        ix->lastQ = std::max(qA, qB) | jumpJ << 8;
    }
    A.entrySlots = entrySlots ? 1u : 0u;
    A.text4 = ix->d_text4; A.text2 = win2 ? ix->d_text2 : nullptr; A.nflag = win2 ? reinterpret_cast<const uint8_t*>(ix->d_nflag) : nullptr; A.win2 = win2 ? 1u : 0u; A.textBegin = text_begin; A.vqCap = vqCap; A.verifyRows = verifyRows ? verifyRows : 1u; A.ldsDepth = ldsDepth; A.winChunks = winChunks; A.lqCap = lqCap;
    A.workCounter = reinterpret_cast<unsigned long long*>(ix->d_small);
    A.errorFlag = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ix->d_small) + SMALL_ERR_OFF);
    // (a wavefront without a node for seconds on end is spinning: a root has a few hundred patterns at most.  iter_cap bounds ALL iterations: tests)
    if (ix->tune.iterCap > 0) { A.guardCap = (uint32_t)ix->tune.iterCap; A.guardKeep = 0xFFFFFFFFu; }
    else { A.guardCap = ix->tune.stallCap > 0 ? (uint32_t)ix->tune.stallCap : (1u << 22); A.guardKeep = 0u; }
    A.counters = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ix->d_small) + 16);
    A.sa = ix->d_sa; A.saMark = ix->d_saMark; A.saSamples = ix->d_saSamples; A.cumGlobal = ix->d_cum; A.nSeqGlobal = ix->nSeq;
    A.posBase = S->posBase; A.windowLen = S->posEnd - S->posBase;
    A.textS = ix->d_textS;
    A.ctx = ix->tune.useCtx ? ix->d_ctx : nullptr;
    A.verifyT = verifyT;
    {   // fast verification (gm_engine.h: fv_masks): every text symbol an item may look at must lie inside its row's record, i.e. K <= 32
        // symbols to the right of the anchor and at most CTX_LEFT to its left -- the anchor is the node's window coordinate `a`, which never
        // exceeds the start of its search (n - 1 + startPos, any block shape of the call) -- and the window must fit the masks
        uint32_t maxA0 = 0;
        for (uint32_t n = 1; n <= plan.stepSize; ++n)
            for (uint32_t s = 0; s < plan.nSearches; ++s) maxA0 = std::max(maxA0, n - 1u + oss_start(plan.table[(size_t)(n - 1) * 8 + s]));
        A.fastVerify = (verifyT && A.ctx && !ix->wide && ix->tune.fastVerify != 0 && p->K <= 32u && p->K + plan.stepSize - 1u <= FV_MAXW && maxA0 <= (uint32_t)CTX_LEFT) ? 1u : 0u;
    }
    A.verifyTExt = verifyTExt;   // (== verifyT at e = 0: the plain stores rely on one row per k-mer and strand)
    A.satMinW = (uint32_t)std::max(1, ix->tune.satMinW);   // default 256: narrow nodes finish sooner than the lookup takes (r01h sweep: 128-256 best)
    // profiles/r01e_infix_sweeps.txt (r01h): 16 / 8 / 4 on the 249 Mbp index; beyond 2^30 rows the fetch loads are HBM
    // misses and larger batches pay (3.09 Gbp: e=0 82.6 vs 90.2 ms, K100 e=1 861 vs 900 ms with 32)
    const bool huge = ix->nRows >= (1ull << 30);
    // (K=100 e=1 on 3.09 Gbp: 48 -> -6 % over 32, 64 the same, 16 +8 %; profiles/r05/sweep_k100_knobs.txt, sweep_k100_lds_fetch_batch_steal.txt)
    A.fetchBatch = p->E == 0 ? (huge ? 32u : 16u) : p->E == 1 ? (huge ? (p->K >= 64 ? 48u : 32u) : 8u) : (huge ? 8u : 4u);   // (e=2 on 3.09 Gbp: 8 -> -1.7 % over 4, 16 +3 %; profiles/r04/sweep_retune_e2.txt)
    if (ix->tune.fetchBatch > 0) A.fetchBatch = (uint32_t)std::min(ix->tune.fetchBatch, 64);
    // (3.09 Gbp: 8 -> K=30 e=2 -2.0 %, e=1 -2.0 % over 1, 4 the same, 16 none; K=100 e=1 +1.3 % with any batch; profiles/r05/sweep_pat_batch.txt)
    A.patBatch = ix->tune.patBatch > 0 ? (uint32_t)std::min(ix->tune.patBatch, 64) : (p->K < 64 ? 8u : 1u);
    // e = 0: a single row is almost always the k-mer's own location.  Beyond ~1 G rows a lone-row step is an HBM miss
    // like the verification reads it postpones, and no longer pays (3.09 Gbp e = 2: 90.7 vs 86.7 M k-mers/s without).
    // 3.09 Gbp: e=1 +8.5 %, e=2 +4 % over 0 (profiles/r02/sweep_grch38_retune.txt); K=100: no difference.  With the neighbour filter the
    // chance hits the two steps used to kill are mostly gone before they become nodes: e=2 one step 371-373 ms against 376-379 with
    // two, e=1 no difference (profiles/r03/sweep_neighbour_filter.txt, sweep_knobs_after_filter.txt)
    // r04, with the two-row filter and the groups (chance hits are mostly gone before they become nodes): e=2 0 -> -3.2 % over 1, e=1 1 -> -1.2 %
    // over 2 (profiles/r04/sweep_retune_e*.txt); K=100 e=1: 0 -> -1.3 %
    A.probation = p->E == 1 && p->K < 64 ? 1u : 0u;
    if (ix->tune.probation >= 0) A.probation = (uint32_t)ix->tune.probation;
    A.verifyCost = (uint32_t)std::max(0, ix->tune.verifyCost);
    A.selfHit = ix->tune.selfHit != 0 ? 1u : 0u;
    A.nbFilter = (uint32_t)std::max(0, ix->tune.jumpFilter);   // 0 off, 1 one- and two-row entries, 2 one-row entries only
    A.chunkBlocks = chunked ? p->chunk_blocks : 0u; A.chunkStride = p->chunk_stride; A.chunkIndex = p->chunk_index;
    A.skipDup = ix->tune.skipDup >= 0 ? (uint32_t)(ix->tune.skipDup != 0) : 1u;   // profiles/r02: +3..8 % on 3.09 Gbp, +1..5 % on 249 Mbp
    // groups of lanes read the rank blocks (rank2_coop): +4..12 % with 32-byte blocks on 249 Mbp and 3.09 Gbp (profiles/r02)
    // work sharing inside the wavefront (value = idle lanes that trigger an exchange): every idle lane at e <= 1 on large
    // indexes (3.09 Gbp: e=1 +8 %, e=0 +1 %) and in small calls (tail of the kernel).  At e = 2 that costs 6..30 % -- stolen
    // subtrees are searched before the counters that would have pruned them saturate -- but an exchange only when a quarter of
    // the wavefront is idle gains 1.5 % (3.09 Gbp) to 5 % (249 Mbp) (profiles/r02/sweep_steal_e2.txt, sweep_chr1_steal_*.txt)
    uint32_t stealDefault = 0u;
    if ((p->E <= 1 && p->K < 64 && ix->nRows >= (1ull << 30)) || S->numRoots < 64ull * 4ull * 1024ull) stealDefault = 1u;
    else if (p->E >= 2) stealDefault = p->K < 64 ? (expand ? 8u : 16u) : (p->E >= 3 ? 4u : 8u);   // (the walker of the split search: 8 -> -3.3 % over 16, 4 the same; profiles/r06)   // K=100 e=2: 8 -> +5.7 %, 16 -> 0 (sweep_steal_longk.txt); K=101 e=3 / e=4: 4 -> -2 / -7 % over 8 (r04)
    // e=1 at K >= 64: sharing used to lose 5..9 % (r02); with verified runs added in two atomics the balance turned: an exchange when a
    // quarter of the wavefront is idle gains 2-6 % on 3.09 Gbp (profiles/r04/sweep_k100_knobs.txt, sweep_verify_t_ext.txt)
    // (round 5, with four blocks per CU again: an exchange at half a wavefront idle, -3.4 % over a quarter)
    else if (p->E == 1 && p->K >= 64) stealDefault = 32u;
    A.steal = ix->tune.steal >= 0 ? (uint32_t)std::min(ix->tune.steal, 64) : stealDefault;
    A.coop = ix->tune.coop >= 0 ? (uint32_t)(ix->tune.coop != 0) : ((ix->wpp == 1 || ix->wide) ? 1u : 0u);   // (wide: groups of four lanes per 64-byte block, r04)
    if (ix->wpp == 9) A.coop = 0u;
    A.jumpJ = jumpJ; A.jumpAPacked[0] = jumpAPacked[0]; A.jumpAPacked[1] = jumpAPacked[1];
    A.patterns = ix->d_patterns; A.jinfo = ix->d_jinfo; A.jinfo2 = ix->d_jinfo2; A.jtab = jtab; A.jbits = jbitsCall; A.jbitsWords = jbitsWords;
    for (uint32_t k = 0; k < GROUP_MAX_MASKS; ++k) A.gmask[k] = gmaskCall[k];
    for (uint32_t k = 0; k < 8u; ++k) { A.layShift[k] = layShift[k]; A.layPlane0[k] = layPlane0[k]; A.layPlane1[k] = layPlane1[k]; }
    A.tableL = longK ? ix->d_tableL : nullptr;
    A.pktChunks = S->pktChunks; A.wmap = ix->d_wmap; A.itemsPerBlock = S->itemsPerBlock; A.expandBlocks = S->expandBlocks; A.numBlocksCall = S->numBlocksCall;
    A.ldsPad = ldsPad;
    A.satDrawW = (uint32_t)(ix->tune.satDrawW >= 0 ? ix->tune.satDrawW : 1);   // (3.09 Gbp K=30 e=2: 1 -> 267 ms, 4 -> 269, 16 -> 285, never -> 326 with the test's two counters requested at the draw and looked at an iteration later)
    if (longK) { A.lqCap = 0u; A.entrySlots = 0u; A.selfHit = 0u; A.spillDepth = depth; A.steal = ix->tune.steal >= 0 ? (uint32_t)std::min(ix->tune.steal, 2) : 2u; }   // (2: lanes share before every root draw, 3.09 Gbp K=300 e=1 +19 %, K=1000 e=1 +14 %, e=0 +4 % over sharing at the end only; profiles/r05/longk_scale.txt)
    A.sliceBegin = text_begin; A.sliceLen = text_len; A.ownBegin = 0; A.ownEnd = text_len; A.ownChunkLen = 0; A.selBlocks = nullptr; A.nSelBlocks = 0;
    *Aout = A;
    return GM_OK;
}

// Can an accumulator (u32) of a counting call reach 2^32?  What is added to the accumulator of ONE (position, strand):
//   * one clamped add (<= addCap) per leaf step, i.e. at most one per distinct string within Hamming distance E of the k-mer, the text
//     letter N included as a substitute: L = sum_{e <= E} C(K, e) 4^e strings per strand (every error configuration is searched exactly
//     once, src/find2_index_approx.hpp:67-134);
//   * ones from verified rows and self hits when the call has no difference plane: at most VERIFY_TMAX rows per narrow node, and a narrow
//     node is a prefix of one of those strings: <= VERIFY_TMAX * K * L;
//   * ones from the correction pass: at most one per (text window with N, strand): 2 * corrKmers.
// If their sum stays below 2^32 no wrap-around can happen and the adds need not return anything (gm_kernels.h: CountEnv::add_acc).
static bool acc_cannot_wrap(uint32_t K, uint32_t E, uint32_t strands, uint32_t addCap, uint64_t corrKmers)
{
    long double L = 0, c = 1, p4 = 1;
    for (uint32_t e = 0; e <= E; ++e) { L += c * p4; c = c * (long double)(K - e) / (long double)(e + 1); p4 *= 4; }
    L *= strands;
    const long double bound = L * addCap + (long double)VERIFY_TMAX * K * L + 2.0L * (long double)corrKmers;
    return bound < 4.0e9L;
}

template <typename TValue>
static int launch_reset_limits(gm_index* ix, TValue* d_out, uint32_t n_seq, uint32_t K, hipStream_t st)
{
    hipLaunchKernelGGL(reset_limits_kernel<TValue>, dim3(n_seq), dim3(64), 0, st, d_out, ix->d_cumLocal, n_seq, K);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

// ---- the split search (gm_expand.h): slices of "phase A writes node packets, the walker draws them" ------------------------------------------
__global__ void expand_reset_kernel(ExpandProgress* prog, unsigned long long totalChunks)
{
    if (threadIdx.x != 0u || blockIdx.x != 0u) return;
    prog->committed = 0ull; prog->totalChunks = totalChunks; prog->slices = 0u; prog->lastChunks = 0u; prog->lastX = prog->lastY = 0u;   // (the stamp counter runs on: packets of earlier calls stay stale)
}

// The main search of a call whose roots have jump patterns.  Everything is queued on the call's stream; the host only reads, two slices
// behind, how far the call has got (a slice takes as many chunks as its packet buffers hold -- decided on the device from what the slice
// before it produced), so the device always has the next slice in its queue.
static int run_expand(gm_index* ix, const SearchSetup& S, SearchArgs A, const gm_map_params* p, hipStream_t st)
{
    const uint32_t U = PKT_HEADER_UNITS + S.pktChunks;            // 16-byte units per packet
    if (!ix->d_xctl) {
        GM_HIP(hipMalloc(&ix->d_xctl, 2 * sizeof(ExpandCtl)));
        GM_HIP(hipStreamCreateWithFlags(&ix->stXA, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            GM_HIP(hipStreamCreateWithFlags(&ix->stXB[i], hipStreamNonBlocking));
            GM_HIP(hipEventCreateWithFlags(&ix->evXA[i], hipEventDisableTiming));
            GM_HIP(hipEventCreateWithFlags(&ix->evXB[i], hipEventDisableTiming));
        }
        GM_HIP(hipEventCreateWithFlags(&ix->evXGo, hipEventDisableTiming));
        GM_HIP(hipMalloc(&ix->d_xprog, sizeof(ExpandProgress)));
        GM_HIP(hipMemset(ix->d_xprog, 0, sizeof(ExpandProgress)));
        GM_HIP(hipHostMalloc(&ix->h_xprog, 4 * sizeof(ExpandProgress)));
        for (int i = 0; i < 4; ++i) GM_HIP(hipEventCreateWithFlags(&ix->evX[i], hipEventDisableTiming));
    }
    // packet buffers: X (patterns without / with one substitution, from its two ends) and Y (two or more) are the two parts of ONE allocation
    // that is kept between calls and only ever grows
    const bool twoPlus = p->E >= 2;
    const uint64_t have = ix->pktCap * 16ull;
    uint64_t budget;
    if (ix->tune.expandMB > 0) budget = (uint64_t)ix->tune.expandMB << 20;
    else {
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) freeB = 0;
        // (a third of what is free, 32 GiB at most: 3.09 Gbp K=30 e=2 on 10 % of the text 1056 / 651 / 577 / 576 ms with 6 / 12 / 24 / 40 GiB -- a slice ends
        //  with the wavefronts waiting for its heaviest packets, and the longer the lists the more of a root's exact hits are counted before its patterns with errors are drawn)
        budget = std::max<uint64_t>(std::min<uint64_t>((freeB + have) / 3, 32ull << 30), have);   // (never below what is there)
    }
    // what the call can use at all: every (work item, rotation) a packet (64 per item at most), rounded up generously
    const uint64_t worstPackets = std::min<uint64_t>(S.numBlocksCall * (uint64_t)S.itemsPerBlock * 64ull + (1u << 20), 1ull << 31);
    uint64_t pkts = std::min<uint64_t>(budget / (16ull * U), worstPackets + (twoPlus ? worstPackets : 0));
    // (a call over a small share of the text -- one of eight ranks on one device, a range of a sweep -- does not take gigabytes it cannot fill: eight packets per k-mer is
    //  twice what a genome makes at K=30 e=2; a repeat-rich share just takes more slices)
    if (ix->tune.expandMB <= 0) pkts = std::min<uint64_t>(pkts, S.kmers * 8ull + (1ull << 22));
    // phase A: as many wavefronts as the device holds, but no more than leave three quarters of a buffer to packets (a wavefront keeps one
    // open region per class)
    int perCU = 0;
    GM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, expand_kernel, 256, 0));
    perCU = std::max(1, std::min(perCU, 8));
    if (S.overlap) perCU = 1;                                  // beside the walker: ONE block per CU takes the slot the walker's three blocks leave
    if (ix->tune.expandOcc > 0) perCU = ix->tune.expandOcc;   // (measurement: blocks of phase A per CU, resident or not)
    uint64_t blocksA = std::min<uint64_t>((uint64_t)ix->numCU * (uint64_t)perCU, std::max<uint64_t>(1, (S.totalChunks + 3) / 4));
    blocksA = std::max<uint64_t>(1, std::min<uint64_t>(blocksA, (twoPlus ? pkts * 2 / 5 : pkts) / (S.overlap ? 2u : 1u) / (4ull * 4ull * XREGION)));
    // one block of work at its worst (64 rotations per item) must fit a quarter of either buffer beside the open regions: tiny budgets are raised
    const uint64_t floorCap = 4ull * (uint64_t)S.itemsPerBlock * 64ull + 2ull * blocksA * 4ull * XREGION;
    pkts = std::max<uint64_t>(pkts, twoPlus ? floorCap * 5 / 2 + 16 : floorCap);
    pkts = std::min<uint64_t>(pkts, (1ull << 31) - 1);
    if (ix->pktCap < pkts * U) {
        uint4* d = nullptr;
        if (hipMalloc(&d, pkts * U * 16) != hipSuccess) {
            (void)hipGetLastError();
            if (ix->pktCap / U < (twoPlus ? floorCap * 5 / 2 + 16 : floorCap)) { set_error("no device memory for %llu node packets", (unsigned long long)pkts); return GM_ERR_OOM; }
            pkts = ix->pktCap / U;       // (what is there will do: more, smaller slices)
        } else {
            if (ix->d_pkt) GM_HIP(hipFree(ix->d_pkt));   // (synchronises with the device: nothing reads the old buffer any more)
            GM_HIP(hipMemsetAsync(d, 0, pkts * U * 16, st));
            ix->d_pkt = d; ix->pktCap = pkts * U; ix->pktUnits = U;
        }
    } else if (ix->tune.expandMB <= 0) pkts = ix->pktCap / U;        // (everything that is there; a forced budget -- tests -- uses its share of it)
    if (ix->pktUnits != U) {   // packets of another size lie in the buffer: a slot's stamp word would be somebody's window symbols
        GM_HIP(hipMemsetAsync(ix->d_pkt, 0, ix->pktCap * 16, st));
        ix->pktUnits = U;
    }
    pkts = std::min<uint64_t>(pkts, (1ull << 31) - 1);
    // overlap: two sets of buffers -- phase A fills one while the walker empties the other
    const bool overlap = S.overlap && pkts / 2 >= (twoPlus ? floorCap * 5 / 2 + 16 : floorCap);
    const uint64_t setPkts = overlap ? pkts / 2 : pkts;
    uint64_t capY = twoPlus ? setPkts * 3 / 5 : 0, capX = setPkts - capY;
    uint4* const setBase[2] = {ix->d_pkt, ix->d_pkt + (overlap ? setPkts * U : 0)};
    auto use_set = [&](SearchArgs& a, uint32_t q) {
        a.pktX = setBase[q]; a.pktY = twoPlus ? setBase[q] + capX * U : setBase[q];
        a.capX = (uint32_t)capX; a.capY = (uint32_t)std::max<uint64_t>(capY, XREGION);   // (capY is never reached with one substitution at most: no class-2 packet exists)
        a.xctl = reinterpret_cast<ExpandCtl*>(ix->d_xctl) + q;
    };
    use_set(A, 0);
    A.xctl = reinterpret_cast<ExpandCtl*>(ix->d_xctl);
    ExpandProgress* prog = reinterpret_cast<ExpandProgress*>(ix->d_xprog);
    ExpandProgress* hprog = reinterpret_cast<ExpandProgress*>(ix->h_xprog);
    const uint64_t slack = blocksA * 4ull * XREGION;
    const uint32_t usableX = (uint32_t)(capX - std::min<uint64_t>(slack, capX / 2)), usableY = twoPlus ? (uint32_t)(capY - std::min<uint64_t>(slack, capY / 2)) : 0xFFFFFFFFu;   // (no packet of the third class exists with one substitution at most)
    uint32_t slicesTotal = 0;
    auto run_pass = [&](uint32_t mode, uint32_t ipb, const uint32_t* wmapPtr, uint32_t wantBlocks) -> int {
    // A chunk is the unit that is redone when its packets do not fit: whatever a chunk can produce at most (64 rotations per work item) must fit
    // half a buffer, or a slice could fail on its first chunk for ever.  (Only the tiny buffers of the tests ever shorten a chunk.)
    const uint64_t usableMin = std::min<uint64_t>(usableX, twoPlus ? usableY : usableX);
    const uint64_t chunkWorst = (uint64_t)ipb * 64ull;
    const uint32_t G = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(wantBlocks, (usableMin / 2) / std::max<uint64_t>(chunkWorst, 1)));
    const uint64_t totalChunks = (S.numBlocksCall + G - 1) / G;
    A.expandBlocks = G; A.itemsPerBlock = ipb; A.wmap = wmapPtr; A.xmode = mode;
    A.wmapRoots = ix->d_wmap + S.itemsPerBlock; A.rootsPerBlockA = S.rootsPerBlock; A.xshare = ix->tune.expandShare != 0 ? 1u : 0u;   // (3.09 Gbp, 10 % / 30 % of the k-mers: K=30 e=2 540 -> 527 ms, e=1 323 -> 320: phase A is bound by its fetches, not by its instructions)
    // first slice: a guess at the packets a chunk makes (8 per k-mer); the slices behind it follow what their predecessor measured
    const uint64_t guess = (uint64_t)G * S.plan.stepSize * (mode == 1u ? 2ull : 8ull);
    const uint32_t firstChunks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(usableMin / std::max<uint64_t>(guess, 1), 0x7FFFFFFFull));
    hipLaunchKernelGGL(expand_reset_kernel, dim3(1), dim3(1), 0, st, prog, (unsigned long long)totalChunks);
    unsigned long long lastSeen = 0; bool seen = false;
    uint32_t i = 0;
    // Overlap: phase A of every slice runs in slice order on a stream of its own; the walkers of even and odd slices have a stream each (the
    // tail of one walker -- wavefronts that have run out of packets -- is filled by the head of the next); phase A of slice i waits for the
    // walker of slice i - 2, which read the buffers it fills.  Without overlap everything is queued on the call's stream.
    hipStream_t sa = overlap ? ix->stXA : st;
    if (overlap) {
        GM_HIP(hipEventRecord(ix->evXGo, st));                       // accumulators cleared, counters zeroed, buffers zeroed
        GM_HIP(hipStreamWaitEvent(ix->stXA, ix->evXGo, 0));
        for (int q = 0; q < 2; ++q) GM_HIP(hipStreamWaitEvent(ix->stXB[q], ix->evXGo, 0));
    }
    for (;; ++i) {
        const uint32_t q = overlap ? (i & 1u) : 0u;
        use_set(A, q);
        hipStream_t sb = overlap ? ix->stXB[q] : st;
        if (overlap && i >= 2u) GM_HIP(hipStreamWaitEvent(sa, ix->evXB[q], 0));   // the walker of slice i - 2 has left these buffers
        hipLaunchKernelGGL(expand_slice_begin_kernel, dim3(1), dim3(1), 0, sa, prog, A.xctl, usableX, usableY, firstChunks, 0x7FFFFFFFu);
        hipLaunchKernelGGL(expand_kernel, dim3((unsigned)blocksA), dim3(256), 0, sa, A);
        hipLaunchKernelGGL(expand_slice_commit_kernel, dim3(1), dim3(1), 0, sa, prog, A.xctl, A.errorFlag);
        GM_HIP(hipGetLastError());
        GM_HIP(hipMemcpyAsync(&hprog[i & 3u], prog, sizeof(ExpandProgress), hipMemcpyDeviceToHost, sa));
        GM_HIP(hipEventRecord(ix->evX[i & 3u], sa));
        if (overlap) { GM_HIP(hipEventRecord(ix->evXA[q], sa)); GM_HIP(hipStreamWaitEvent(sb, ix->evXA[q], 0)); }
        int rc = launch_search(ix, LEAF_COUNT_NODES, A, S.blocks, sb); if (rc) return rc;
        if (overlap) GM_HIP(hipEventRecord(ix->evXB[q], sb));
        if (i >= 2u) {
            GM_HIP(hipEventSynchronize(ix->evX[(i - 2u) & 3u]));
            const unsigned long long c = hprog[(i - 2u) & 3u].committed;
            if (c >= totalChunks) break;
            if (seen && c == lastSeen) { set_error("the split search made no progress (packet buffers of %llu + %llu packets)", (unsigned long long)capX, (unsigned long long)capY); return GM_ERR_INTERNAL; }
            lastSeen = c; seen = true;
        }
    }
    if (overlap) {   // the call's stream goes on when both walkers and phase A have ended
        GM_HIP(hipStreamWaitEvent(st, ix->evXB[0], 0)); GM_HIP(hipStreamWaitEvent(st, ix->evXB[1], 0));
        GM_HIP(hipEventRecord(ix->evXA[0], sa)); GM_HIP(hipStreamWaitEvent(st, ix->evXA[0], 0));
    }
    slicesTotal += i + 1u;
    return GM_OK;
    };
    // One pass over every pattern, or two: the patterns without a substitution first (a work item per root), then the rest for the blocks
    // that are not at MAX yet (gm_expand.h: expand_strip_exact).  The second order costs an easy text a few per cent (more launches) and spares a
    // repeat-rich one most of phase A (profiles/r06).
    const bool twoPass = ix->tune.expandTwoPass != 0 && A.maxVal != 0xFFFFFFFFu;
    if (twoPass) {
        int rc = run_pass(1u, S.rootsPerBlock, ix->d_wmap + S.itemsPerBlock, std::max<uint32_t>(1u, 8192u / std::max<uint32_t>(S.rootsPerBlock, 1u))); if (rc) return rc;
        rc = run_pass(2u, S.itemsPerBlock, ix->d_wmap, S.expandBlocks); if (rc) return rc;
    } else { int rc = run_pass(0u, S.itemsPerBlock, ix->d_wmap, S.expandBlocks); if (rc) return rc; }
    ix->lastSlices = slicesTotal;
    return GM_OK;
}

// wrote (optional): the slice positions [wrote[0], wrote[1]) this call has written into d_out (everything else is untouched,
// except for the zeros of resetLimits at the sequence ends)
// whole (optional): the k-mer range [whole[0], whole[1]) of the CALL this launch is one piece of (gm_map_shard delivers a share in up to four
// launches so that the copy of one piece overlaps the search of the next).  The first piece (ix->pieceIndex == 0) then clears the
// accumulators of the whole range and starts the call's ONE correction pass; the later pieces do neither.
static int map_impl(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
                    const uint64_t* intervals, uint64_t n_intervals, const uint32_t* seq_file_id, void* d_out, hipStream_t st, uint64_t* wrote = nullptr,
                    const uint64_t* whole = nullptr, int firstOfWhole = -1)
{
    if (!d_out) { set_error("null output"); return GM_ERR_BAD_ARG; }
    if (ix && !whole) ix->piece.active = false;   // (any call that is not a piece of a share ends the share a caller was delivering: gm_map_device)
    SearchSetup S; SearchArgs A;
    int rc = prepare_search(ix, text_begin, text_len, first_seq, n_seq, p, intervals, n_intervals, st, &S, &A, /*wantJump=*/p->exclude_pseudo == 0, /*leaf queue*/p->exclude_pseudo ? 128u : 0u);
    if (rc) return rc;
    const bool ep = p->exclude_pseudo != 0;
    if (S.jump) { rc = ensure_correction_blocks(ix, p->K, p->E, S.plan.infix, st); if (rc) return rc; }
    uint32_t wordsPerKmer = 0;
    if (ep) {
        if (!ix->d_sa && !ix->d_saMark) { set_error("--exclude-pseudo needs an index with suffix array samples (sampling >= 1)"); return GM_ERR_NEED_LOCATE; }
        if (!seq_file_id) { set_error("--exclude-pseudo needs seq_file_id"); return GM_ERR_BAD_ARG; }
        uint32_t nFiles = 0;
        for (uint32_t s = 0; s < ix->nSeq; ++s) nFiles = std::max(nFiles, seq_file_id[s] + 1);
        wordsPerKmer = (nFiles + 31) / 32;
        rc = grow(&ix->d_seqFile, &ix->seqFileCap, (uint64_t)ix->nSeq); if (rc) return rc;
        GM_HIP(hipMemcpyAsync(ix->d_seqFile, seq_file_id, (size_t)ix->nSeq * 4, hipMemcpyHostToDevice, st));
        rc = grow(&ix->d_bits, &ix->bitsCap, (text_len + 1) * wordsPerKmer); if (rc) return rc;
        GM_HIP(hipStreamSynchronize(st));
        // fasta id per suffix-array row (full array, 32-bit rows, at most 256 files): built once per file assignment, one byte per row
        if (ix->d_sa && !ix->wide && nFiles <= 256u) {
            uint64_t h = 1469598103934665603ull;
            for (uint32_t s = 0; s < ix->nSeq; ++s) h = (h ^ seq_file_id[s]) * 1099511628211ull;
            if (!ix->rowFileValid || ix->rowFileSig != h) {
                ix->rowFileValid = false;
                if (!ix->d_rowFile && hipMalloc(&ix->d_rowFile, ix->nRows) != hipSuccess) { (void)hipGetLastError(); ix->d_rowFile = nullptr; }
                if (ix->d_rowFile) {
                    hipLaunchKernelGGL(row_file_kernel, dim3(grid_for(ix->nRows)), dim3(256), 0, st, (const uint32_t*)ix->d_sa, ix->nRows, ix->d_cum, ix->nSeq, ix->d_seqFile, ix->d_rowFile);
                    GM_HIP(hipGetLastError());
                    ix->rowFileSig = h; ix->rowFileValid = true;
                }
            }
        } else ix->rowFileValid = false;
    } else {
        rc = grow(&ix->d_acc, &ix->accCap, 2 * (text_len + 4) + 16); if (rc) return rc;
    }
    // E = 0 with single-row verification: plain stores into one plane per strand instead of atomics
    const bool longK = p->K > MAX_K;   // gm_longk.h: counts go through the accumulators at any E
    const bool store = !ep && p->E == 0 && A.verifyT <= 1 && !ix->tune.noStore && !longK;
    const uint64_t plane = (text_len + 4 + 15) & ~15ull;   // both planes aligned alike: finalize2 reads 16 bytes per lane
    // counting kernels on the regular partition: verified runs of k-mers go into a difference plane behind acc (gm_kernels.h: CountEnv::leaf_range)
    // (finalize_diff_kernel restarts its running sum at every multiple of stepSize from the range's first position and CountEnv::leaf_range
    //  never lets a run leave its block: both hold because ranges and chunks are whole blocks of the regular partition -- checked, not assumed)
    const bool useDiff = !ep && !store && !longK && !S.plan.useList && ix->tune.rangeAdd != 0 && S.posBase % S.plan.stepSize == 0 && S.sel.len % S.plan.stepSize == 0;
    const uint64_t diffOff = (text_len + 4 + 3) & ~3ull;
    // kernels over positions: blockIdx.y walks the shard's own chunk ranges (one range without chunks), blockIdx.x one range
    auto range_grid = [](const ChunkSel& c, uint64_t n, uint32_t perThread) {
        uint64_t nr = 1, span = n;
        if (c.len) { const uint64_t chunks = (n + c.len - 1) / c.len; nr = chunks > c.index ? (chunks - c.index + c.stride - 1) / c.stride : 0; span = c.len; }
        const uint64_t gx = std::max<uint64_t>(1, std::min<uint64_t>((span / perThread + 255) / 256, 1u << 16));
        return dim3((unsigned)gx, (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nr, 65535)), 1);
    };

    // a shard (kmer_begin/kmer_end) touches only its own positions [r0, r1) of the accumulators and of out
    const bool sharded = (p->flags & (GM_MAP_FLAG_RANGE | GM_MAP_FLAG_PIECE)) || p->kmer_begin != 0 || p->kmer_end != 0 || S.sel.len != 0;
    const uint64_t r0 = sharded ? std::min<uint64_t>(S.posBase, text_len) : 0;
    const uint64_t r1 = sharded ? std::min<uint64_t>(std::max<uint64_t>(S.posEnd, r0), text_len) : text_len;
    const uint64_t rn = r1 - r0;
    if (wrote) { wrote[0] = r0; wrote[1] = r1; }
    if (ix->pieceIndex == 0) GM_HIP(hipEventRecord(ix->ev[0], st));
    const ChunkSel sel = S.sel;   // positions are relative to r0 == posBase (a multiple of the block length)
    // the positions of the whole CALL, [c0, c1): what the first piece clears and what the call's one correction pass owns
    const bool firstPiece = firstOfWhole >= 0 ? firstOfWhole != 0 : ix->pieceIndex == 0;
    uint64_t c0 = r0, c1 = r1;
    if (whole) {   // (regular partition: a selection is never delivered in pieces; pieces begin at multiples of the chunk row from whole[0])
        const uint64_t step = S.plan.stepSize;
        const uint64_t bb = std::min<uint64_t>((whole[0] + step - 1) / step, S.plan.numBlocks), be = std::min<uint64_t>((whole[1] + step - 1) / step, S.plan.numBlocks);
        c0 = std::min<uint64_t>(bb * step, text_len);
        c1 = std::min<uint64_t>(std::max<uint64_t>(std::min<uint64_t>(be * step, S.plan.numKmers), c0), text_len);
        if (S.plan.useList || (rn > 0 && (r0 < c0 || r1 > c1))) { set_error("internal: a piece outside its call"); return GM_ERR_INTERNAL; }
    }
    const uint64_t cn = c1 - c0;
    if (cn > 0 && (!whole || firstPiece)) {
        if (ep) GM_HIP(hipMemsetAsync(ix->d_bits + c0 * wordsPerKmer, 0, cn * wordsPerKmer * sizeof(uint32_t), st));
        else if (store) {
            const size_t pb = p->value_bits == 8 ? 1 : 2;   // plane element: as wide as the result
            if (sel.len) {
                hipLaunchKernelGGL(clear_chunks_kernel, range_grid(sel, cn, 16), dim3(256), 0, st, (uint8_t*)ix->d_acc + c0 * pb, (uint32_t)pb, cn, sel);
                hipLaunchKernelGGL(clear_chunks_kernel, range_grid(sel, cn, 16), dim3(256), 0, st, (uint8_t*)ix->d_acc + (plane + c0) * pb, (uint32_t)pb, cn, sel);
            } else {
                GM_HIP(hipMemsetAsync((uint8_t*)ix->d_acc + c0 * pb, 0, cn * pb, st));
                GM_HIP(hipMemsetAsync((uint8_t*)ix->d_acc + (plane + c0) * pb, 0, cn * pb, st));
            }
        } else if (sel.len) {
            hipLaunchKernelGGL(clear_chunks_kernel, range_grid(sel, cn, 4), dim3(256), 0, st, (uint8_t*)(ix->d_acc + c0), 4u, cn, sel);
            if (useDiff) hipLaunchKernelGGL(clear_chunks_kernel, range_grid(sel, cn, 4), dim3(256), 0, st, (uint8_t*)(ix->d_acc + diffOff + c0), 4u, cn, sel);
        } else {
            GM_HIP(hipMemsetAsync(ix->d_acc + c0, 0, cn * sizeof(uint32_t), st));
            if (useDiff) GM_HIP(hipMemsetAsync(ix->d_acc + diffOff + c0, 0, cn * sizeof(uint32_t), st));
        }
    }
    // [0, 8) work counter of the main search, [8, 16) of the correction pass (which may still run beside a later piece), [16, 512) statistics
    GM_HIP(hipMemsetAsync(ix->d_small, 0, firstPiece ? SMALL_ZEROED : 8, st));   // later pieces of one call keep adding to the statistics
    A.acc = ix->d_acc; A.accPlane = plane; A.fileBits = ix->d_bits;
    A.diff = useDiff ? ix->d_acc + diffOff : nullptr;
    // self hits of the counting kernels pay only with the difference plane (one atomic per block instead of one per k-mer)
    if (!store && !useDiff) A.selfHit = 0u;
    A.maxVal = ix->tune.noSaturate ? 0xFFFFFFFFu : (p->value_bits == 8 ? 255u : 65535u);
    A.addCap = std::min<uint32_t>(A.maxVal, 0xFFFFu);
    A.noWrap = (ix->tune.noWrap != 0 && acc_cannot_wrap(p->K, p->E, S.plan.nStrands, A.addCap, S.jump ? ix->nCBlocks * (uint64_t)S.plan.stepSize : 0)) ? 1u : 0u;
    A.wordsPerKmer = wordsPerKmer; A.seqFile = ix->d_seqFile; A.rowFile = (ep && ix->rowFileValid) ? ix->d_rowFile : nullptr;

    if (S.jump && ix->nCBlocks > 0 && cn > 0 && !(S.plan.useList && S.plan.blocks.empty()) && (!whole || firstPiece)) {
        // Correction pass: the text windows that hold N, from the whole index, searched with the full rules; every occurrence inside
        // this CALL's positions adds one at its own position (ScatterEnv).  ONE launch per call, on a stream of its own and in front of
        // the main search: it is a few thousand roots whose cost is the latency of their dependent steps (3.09 Gbp: 2.4 ms at K=100
        // e=1, 12-19 ms at e=2), so it runs in a corner of the device while the main search fills the rest.  Round 4 ran it behind
        // every launch of a share -- four times per gm_map_shard call: a fixed cost per rank that did not shrink with the number of ranks.
        // Order does not matter for the result: both kernels only ever add to acc (gm_kernels.h: the invariant next to ScatterEnv::leaf).
        if (!ix->stCorr) {
            GM_HIP(hipStreamCreateWithFlags(&ix->stCorr, hipStreamNonBlocking));
            GM_HIP(hipEventCreateWithFlags(&ix->evCorrGo, hipEventDisableTiming));
            GM_HIP(hipEventCreateWithFlags(&ix->evCorrDone, hipEventDisableTiming));
            GM_HIP(hipEventCreateWithFlags(&ix->evCorrStart, hipEventDisableTiming));
        }
        SearchArgs C = A;
        C.text = ix->d_text; C.textBegin = 0; C.numKmers = ix->textLen >= p->K ? ix->textLen - p->K + 1 : 0;
        C.blockList = ix->d_cblocks; C.blockBegin = 0; C.numRoots = ix->nCBlocks * C.rootsPerBlock; C.chunkBlocks = 0;
        C.ownBegin = c0; C.ownEnd = c1; C.ownChunkLen = sel.len;
        C.selBlocks = S.plan.useList ? ix->d_blocks : nullptr; C.nSelBlocks = (uint32_t)S.plan.blocks.size();
        C.steal = C.numRoots < 64ull * 4ull * 1024ull ? 1u : C.steal;
        C.lqCap = 128u;   // leaves are located by the whole wavefront (gm_kernels.h: LeafQueueEnv)
        C.entrySlots = 0u;
        C.win2 = 0u; C.winChunks = S.rootWinChunks;   // (it draws roots and stages their windows at any alignment: a walker's packets start at nibble 0 and need a chunk less)
        C.workCounter = reinterpret_cast<unsigned long long*>(ix->d_small) + 1;   // its own counter: the two kernels run side by side
        C.stack = ix->d_stack + ix->stackCap / 2;                                  // ... and its own half of the spill area (prepare_search)
        const uint64_t useful = (C.numRoots + 255) / 256;
        const unsigned cb = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ix->numCU * std::max(1, std::min(ix->tune.blocksPerCU, 4)), useful));
        GM_HIP(hipEventRecord(ix->evCorrGo, st));           // accumulators cleared, counters zeroed
        GM_HIP(hipStreamWaitEvent(ix->stCorr, ix->evCorrGo, 0));
        // The main search must not be dispatched first: it is persistent and takes every CU, the correction pass would then start when
        // the main search ends (measured: 19 ms at the END of a K=30 e=2 call).  Its stream therefore waits for an event that the correction
        // stream records right in front of its kernel -- the small kernel is in the device's queue when the large one becomes eligible.
        GM_HIP(hipEventRecord(ix->evCorrStart, ix->stCorr));
        GM_HIP(hipStreamWaitEvent(st, ix->evCorrStart, 0));
        GM_HIP(hipEventRecord(ix->ev[1], ix->stCorr));
        rc = launch_search(ix, LEAF_SCATTER, C, std::min(cb, std::max(1u, S.blocks)), ix->stCorr); if (rc) return rc;
        GM_HIP(hipEventRecord(ix->ev[2], ix->stCorr));
        GM_HIP(hipEventRecord(ix->evCorrDone, ix->stCorr));
        ix->corrTimed = true; ix->corrPending = true;
    } else if (!whole || firstPiece) ix->corrTimed = false;
    const uint32_t slot = (uint32_t)(ix->evCount % gm_index::EV_RING);
    GM_HIP(hipEventRecord(ix->evRing[slot][0], st));
    if (S.numRoots > 0 && S.expand && !ep && !store) { rc = run_expand(ix, S, A, p, st); if (rc) return rc; }
    else if (S.numRoots > 0) { rc = launch_search(ix, ep ? LEAF_FILESET : store ? (p->value_bits == 8 ? LEAF_STORE8 : LEAF_STORE) : S.jump ? LEAF_COUNT_JUMP : LEAF_COUNT, A, S.blocks, st); if (rc) return rc; }
    if (ix->corrPending) { GM_HIP(hipStreamWaitEvent(st, ix->evCorrDone, 0)); ix->corrPending = false; }   // finalize reads what the correction pass added
    GM_HIP(hipEventRecord(ix->evRing[slot][1], st));
    ix->evCount++;
    if (text_len > 0) {
        const dim3 g4 = range_grid(sel, rn, 4), g1 = range_grid(sel, rn, 1), g16 = range_grid(sel, rn, 16), g8 = range_grid(sel, rn, 8);
        const uint16_t* pf = (const uint16_t*)ix->d_acc + r0;
        const uint8_t* pf8 = (const uint8_t*)ix->d_acc + r0;
        if (p->value_bits == 8) {
            uint8_t* o = (uint8_t*)d_out + r0;
            if (rn == 0) {}
            else if (ep) hipLaunchKernelGGL(finalize_fileset_kernel<uint8_t>, g1, dim3(256), 0, st, ix->d_bits + r0 * wordsPerKmer, wordsPerKmer, o, rn, sel);
            else if (store) hipLaunchKernelGGL((finalize2_kernel<uint8_t, uint8_t>), g16, dim3(256), 0, st, pf8, pf8 + plane, o, rn, 255u, sel);
            else if (useDiff) hipLaunchKernelGGL(finalize_diff_kernel<uint8_t>, g8, dim3(256), 0, st, ix->d_acc + r0, ix->d_acc + diffOff + r0, o, rn, 255u, sel, S.plan.stepSize);
            else hipLaunchKernelGGL(finalize_kernel<uint8_t>, g4, dim3(256), 0, st, ix->d_acc + r0, o, rn, 255u, sel);
            rc = launch_reset_limits(ix, (uint8_t*)d_out, n_seq, p->K, st);
        } else {
            uint16_t* o = (uint16_t*)d_out + r0;
            if (rn == 0) {}
            else if (ep) hipLaunchKernelGGL(finalize_fileset_kernel<uint16_t>, g1, dim3(256), 0, st, ix->d_bits + r0 * wordsPerKmer, wordsPerKmer, o, rn, sel);
            else if (store) hipLaunchKernelGGL((finalize2_kernel<uint16_t, uint16_t>), g16, dim3(256), 0, st, pf, pf + plane, o, rn, 65535u, sel);
            else if (useDiff) hipLaunchKernelGGL(finalize_diff_kernel<uint16_t>, g8, dim3(256), 0, st, ix->d_acc + r0, ix->d_acc + diffOff + r0, o, rn, 65535u, sel, S.plan.stepSize);
            else hipLaunchKernelGGL(finalize_kernel<uint16_t>, g4, dim3(256), 0, st, ix->d_acc + r0, o, rn, 65535u, sel);
            rc = launch_reset_limits(ix, (uint16_t*)d_out, n_seq, p->K, st);
        }
        if (rc) return rc;
    }
    GM_HIP(hipEventRecord(ix->ev[3], st));
    GM_HIP(hipEventRecord(ix->evDone, st));
    ix->doneValid = true;
    ix->evValid = true;
    if (ix->pieceIndex == 0) { ix->stats = gm_map_stats{}; ix->statPieces = 0; }
    if (!S.expand) ix->lastSlices = 0;
    ix->stats.kmers += S.kmers; ix->stats.roots += S.numRoots; ix->statPieces += 1;
    return GM_OK;
}

static int staged_copy_to_host(gm_index* ix, uint8_t* h_dst, const uint8_t* d_src, uint64_t bytes, hipStream_t st);
__global__ __launch_bounds__(256) void narrow_offsets_kernel(const uint64_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)in[i];
}

// ---- gm_locate: per-position occurrence lists for csv (algo.hpp:311-348) -------------------------------------
static int locate_impl(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
                       const uint64_t* intervals, uint64_t n_intervals, gm_locations* L)
{
    hipStream_t st = nullptr;
    SearchSetup S; SearchArgs A;
    if (ix) ix->piece.active = false;
    int rc = prepare_search(ix, text_begin, text_len, first_seq, n_seq, p, intervals, n_intervals, st, &S, &A, false, 128u);
    if (rc) return rc;
    if (!ix->d_sa && !ix->d_saMark) { set_error("csv output needs an index with suffix array samples (sampling >= 1)"); return GM_ERR_NEED_LOCATE; }
    if (S.sel.len) { set_error("gm_locate does not take interleaved chunks"); return GM_ERR_BAD_ARG; }
    const uint64_t W = S.posEnd - S.posBase;
    L->pos_begin = S.posBase; L->n_positions = W;
    L->plus_off = (uint64_t*)calloc(W + 1, 8); L->minus_off = (uint64_t*)calloc(W + 1, 8);
    L->plus = nullptr; L->minus = nullptr;
    if (!L->plus_off || !L->minus_off) return GM_ERR_OOM;
    if (W == 0 || S.numRoots == 0) return GM_OK;

    // the work buffers of the windows of one index are kept between calls (a csv pass over five FASTA files allocated and released
    // 4.5 GB five times; gm_index_free releases them)
    uint32_t*& d_cnt = ix->d_locCnt; uint64_t*& d_offs = ix->d_locOffs; uint64_t*& d_emit = ix->d_locEmit; uint64_t*& d_sorted = ix->d_locSorted; uint8_t*& d_tmp = ix->d_locTmp;
    uint32_t*& d_segB = ix->d_locSeg; uint32_t* d_segE = nullptr;
    size_t tmpBytes = 0;
    const uint64_t slots = 2 * W;
#define LC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); rc = (e_ == hipErrorOutOfMemory) ? GM_ERR_OOM : GM_ERR_HIP; goto done; } } while (0)
    {
        rc = grow(&d_cnt, &ix->locCntCap, slots + 1); if (rc) goto done;
        rc = grow(&d_offs, &ix->locOffsCap, slots + 1); if (rc) goto done;
        LC(hipMemset(d_cnt, 0, (slots + 1) * 4)); LC(hipMemset(ix->d_small, 0, SMALL_ZEROED));
        A.cnt2 = d_cnt;
        rc = launch_search(ix, LEAF_OCC_COUNT, A, S.blocks, st); if (rc) goto done;
        LC(rocprim::exclusive_scan(nullptr, tmpBytes, d_cnt, d_offs, (uint64_t)0, slots + 1, rocprim::plus<uint64_t>()));
        rc = grow(&d_tmp, &ix->locTmpCap, (uint64_t)(tmpBytes ? tmpBytes : 16)); if (rc) goto done;
        { size_t tb = tmpBytes; LC(rocprim::exclusive_scan(d_tmp, tb, d_cnt, d_offs, (uint64_t)0, slots + 1, rocprim::plus<uint64_t>())); }
        // (every device -> host transfer of this call goes through the page-locked ring: the plain hipMemcpy into pageable memory was
        //  most of the second that config C5's csv window took, profiles/r03/final/bench_c5_bacteria5.json)
        // the offsets land in the caller's two arrays directly (slot W, the end of the plus strand, is the start of the minus strand): no
        // 16 x W bytes of temporary, no serial pass over it -- the first touch of the fresh arrays is taken by the copy's threads
        rc = staged_copy_to_host(ix, (uint8_t*)L->plus_off, (const uint8_t*)d_offs, (W + 1) * 8, st); if (rc) goto done;
        rc = staged_copy_to_host(ix, (uint8_t*)L->minus_off, (const uint8_t*)(d_offs + W), (W + 1) * 8, st); if (rc) goto done;
        const uint64_t nPlus = L->plus_off[W], total = L->minus_off[W];
        if (total >= (1ull << 31)) { set_error("%llu occurrences in one gm_locate window; use a smaller k-mer range", (unsigned long long)total); rc = GM_ERR_TOO_LONG; goto done; }
        {
            uint64_t* mo = L->minus_off;
            const unsigned T = W >= (1ull << 20) ? 8u : 1u;
            std::thread th[7];
            auto part = [=](unsigned t) { for (uint64_t j = (W + 1) * t / T; j < (W + 1) * (t + 1) / T; ++j) mo[j] -= nPlus; };
            for (unsigned t = 1; t < T; ++t) th[t - 1] = std::thread(part, t);
            part(0);
            for (unsigned t = 1; t < T; ++t) th[t - 1].join();
        }
        L->plus = (uint64_t*)malloc((nPlus + 1) * 8); L->minus = (uint64_t*)malloc((total - nPlus + 1) * 8);
        if (!L->plus || !L->minus) { rc = GM_ERR_OOM; goto done; }
        if (total > 0) {
            rc = grow(&d_emit, &ix->locEmitCap, total); if (rc) goto done;
            rc = grow(&d_sorted, &ix->locSortedCap, total); if (rc) goto done;
            LC(hipMemset(d_cnt, 0, (slots + 1) * 4)); LC(hipMemset(ix->d_small, 0, SMALL_ZEROED));
            A.offs = d_offs; A.emit = d_emit;
            rc = launch_search(ix, LEAF_OCC_EMIT, A, S.blocks, st); if (rc) goto done;
            // std::sort of every list (algo.hpp:336,348): segmented radix sort, segments = (position, strand) slots
            rc = grow(&d_segB, &ix->locSegCap, slots + 1); if (rc) goto done;
            hipLaunchKernelGGL(narrow_offsets_kernel, dim3(grid_for(slots + 1)), dim3(256), 0, st, d_offs, slots + 1, d_segB);   // (total < 2^31)
            LC(hipGetLastError());
            d_segE = d_segB + 1;
            size_t sb = 0;
            LC(rocprim::segmented_radix_sort_keys(nullptr, sb, d_emit, d_sorted, (unsigned int)total, (unsigned int)slots, d_segB, d_segE, 0, 64));
            if (sb > tmpBytes) { rc = grow(&d_tmp, &ix->locTmpCap, (uint64_t)sb); if (rc) goto done; tmpBytes = sb; }
            { size_t tb = tmpBytes; LC(rocprim::segmented_radix_sort_keys(d_tmp, tb, d_emit, d_sorted, (unsigned int)total, (unsigned int)slots, d_segB, d_segE, 0, 64)); }
            if (nPlus) { rc = staged_copy_to_host(ix, (uint8_t*)L->plus, (const uint8_t*)d_sorted, nPlus * 8, st); if (rc) goto done; }
            if (total > nPlus) { rc = staged_copy_to_host(ix, (uint8_t*)L->minus, (const uint8_t*)(d_sorted + nPlus), (total - nPlus) * 8, st); if (rc) goto done; }
        }
        LC(hipDeviceSynchronize());
    }
done:
#undef LC
    if (hipEventRecord(ix->evDone, st) == hipSuccess) ix->doneValid = true;
    if (!rc) rc = check_device_error(ix);
    return rc;
}

static int check_device_error(gm_index* ix)
{
    uint32_t flag = 0;
    GM_HIP(hipMemcpy(&flag, reinterpret_cast<char*>(ix->d_small) + SMALL_ERR_OFF, 4, hipMemcpyDeviceToHost));   // synchronises with the device
    if (flag) {
        GM_HIP(hipMemset(reinterpret_cast<char*>(ix->d_small) + SMALL_ERR_OFF, 0, 4));
        set_error("device-side invariant violated in this or an earlier call on the index:%s%s", (flag & 1u) ? " lane stack overflow" : "",
                  (flag & 2u) ? " a wavefront of the search kernel ran past its iteration bound (iter_cap / stall_cap) and gave up" : (flag & 4u) ? " the split search could not fit one chunk of work into its packet buffers" : "");
        return GM_ERR_INTERNAL;
    }
    return GM_OK;
}

}  // namespace gm

extern "C" {

int gm_map_device(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
                  const uint64_t* intervals, uint64_t n_intervals, const uint32_t* seq_file_id, void* out_device, void* stream)
{
    if (p && (p->flags & GM_MAP_FLAG_PIECE)) {   // one launch of a share the caller delivers in several calls (include/genmap_amd.h)
        if (n_intervals > 0 || p->kmer_begin < p->whole_begin || p->kmer_end > p->whole_end) { set_error("GM_MAP_FLAG_PIECE: a piece lies inside its share, and a selection is not delivered in pieces"); return GM_ERR_BAD_ARG; }
        const uint64_t whole[2] = {p->whole_begin, p->whole_end};
        // The pieces of a share arrive in order and nothing else runs on the index between them: the first clears the accumulators and starts the
        // share's one correction pass, the later ones rely on both.  The index remembers the share it is in the middle of.
        const bool first = p->kmer_begin == p->whole_begin;
        gm_index::PieceState& ps = ix->piece;
        if (!first && !(ps.active && ps.wholeBegin == p->whole_begin && ps.wholeEnd == p->whole_end && ps.K == p->K && ps.E == p->E && ps.textBegin == text_begin && ps.textLen == text_len &&
                        ps.chunkIndex == p->chunk_index && ps.chunkStride == p->chunk_stride && ps.next == p->kmer_begin)) {
            set_error("GM_MAP_FLAG_PIECE: this piece does not continue the share the index is in the middle of (pieces come in order, first piece first, and no other call in between)");
            return GM_ERR_BAD_ARG;
        }
        ps.active = false;
        const int rc = map_impl(ix, text_begin, text_len, first_seq, n_seq, p, intervals, n_intervals, seq_file_id, out_device, (hipStream_t)stream, nullptr, whole, first ? 1 : 0);
        if (rc == GM_OK && p->kmer_end < p->whole_end)
            ps = gm_index::PieceState{true, p->whole_begin, p->whole_end, p->kmer_end, text_begin, text_len, p->K, p->E, p->chunk_index, p->chunk_stride};
        return rc;
    }
    if (ix) ix->piece.active = false;
    return map_impl(ix, text_begin, text_len, first_seq, n_seq, p, intervals, n_intervals, seq_file_id, out_device, (hipStream_t)stream);
}

int gm_map(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
           const uint64_t* intervals, uint64_t n_intervals, const uint32_t* seq_file_id, void* out_host)
{
    if (!ix || !p || !out_host) { set_error("null argument"); return GM_ERR_BAD_ARG; }
    if (p->value_bits != 8 && p->value_bits != 16) return GM_ERR_BAD_VALUE_BITS;
    GM_HIP(hipSetDevice(ix->device));
    if (!((p->flags & GM_MAP_FLAG_RANGE) || p->kmer_begin != 0 || p->kmer_end != 0 || (p->chunk_blocks > 0 && p->chunk_stride > 1)))
        return gm_map_shard(ix, text_begin, text_len, first_seq, n_seq, p, intervals, n_intervals, seq_file_id, out_host);   // every position is delivered
    void* d_out = nullptr;
    const size_t bytes = (size_t)text_len * (p->value_bits / 8);
    GM_HIP(hipMalloc(&d_out, bytes + 16));
    int rc = GM_OK;
    if ((p->flags & GM_MAP_FLAG_RANGE) || p->kmer_begin != 0 || p->kmer_end != 0 || (p->chunk_blocks > 0 && p->chunk_stride > 1)) {   // a shard leaves the other positions zero
        hipError_t e = hipMemsetAsync(d_out, 0, bytes, nullptr);
        if (e != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); hipFree(d_out); return GM_ERR_HIP; }
    }
    rc = map_impl(ix, text_begin, text_len, first_seq, n_seq, p, intervals, n_intervals, seq_file_id, d_out, nullptr);
    if (!rc) { hipError_t e = hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost); if (e != hipSuccess) { set_error("copy back failed: %s", hipGetErrorString(e)); rc = GM_ERR_HIP; } }
    if (!rc) rc = check_device_error(ix);
    hipFree(d_out);
    return rc;
}

int gm_map_files(gm_index* ix, uint32_t n_files, const uint32_t* file_first_seq, const uint32_t* file_n_seq, const gm_map_params* p,
                 const uint32_t* seq_file_id, void* const* out_host)
{
    if (!ix || !p || !file_first_seq || !file_n_seq || !out_host || n_files == 0) { set_error("null argument"); return GM_ERR_BAD_ARG; }
    if (p->value_bits != 8 && p->value_bits != 16) return GM_ERR_BAD_VALUE_BITS;
    if ((p->flags & (GM_MAP_FLAG_RANGE | GM_MAP_FLAG_PIECE)) || p->kmer_begin != 0 || p->kmer_end != 0 || (p->chunk_blocks > 0 && p->chunk_stride > 1)) {
        set_error("gm_map_files takes whole files (no k-mer range, no interleaved chunks)"); return GM_ERR_BAD_ARG;
    }
    for (uint32_t f = 0; f < n_files; ++f) {
        if (file_n_seq[f] == 0 || (uint64_t)file_first_seq[f] + file_n_seq[f] > ix->nSeq || (f > 0 && file_first_seq[f] != file_first_seq[f - 1] + file_n_seq[f - 1])) {
            set_error("gm_map_files: the files are consecutive runs of the index's sequences, in ascending order"); return GM_ERR_BAD_ARG;
        }
        if (!out_host[f]) { set_error("null output"); return GM_ERR_BAD_ARG; }
    }
    GM_HIP(hipSetDevice(ix->device));
    const uint32_t s0 = file_first_seq[0], s1 = file_first_seq[n_files - 1] + file_n_seq[n_files - 1];
    const uint64_t tb = ix->cum[s0], tl = ix->cum[s1] - tb;
    const size_t eb = p->value_bits / 8;
    void* d_out = nullptr;
    GM_HIP(hipMalloc(&d_out, tl * eb + 16));
    int rc = map_impl(ix, tb, tl, s0, s1 - s0, p, nullptr, 0, seq_file_id, d_out, nullptr);   // ONE launch over the k-mers of every file
    for (uint32_t f = 0; f < n_files && !rc; ++f) {
        const uint64_t b = ix->cum[file_first_seq[f]] - tb, n = ix->cum[file_first_seq[f] + file_n_seq[f]] - ix->cum[file_first_seq[f]];
        hipError_t e = hipMemcpy(out_host[f], (const uint8_t*)d_out + b * eb, n * eb, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error("copy back failed: %s", hipGetErrorString(e)); rc = GM_ERR_HIP; }
    }
    if (!rc) rc = check_device_error(ix);
    hipFree(d_out);
    return rc;
}

int gm_host_pin(void* host, uint64_t bytes)
{
    if (!host) return GM_ERR_BAD_ARG;
    GM_HIP(hipHostRegister(host, bytes, hipHostRegisterDefault));
    return GM_OK;
}
int gm_host_unpin(void* host)
{
    if (!host) return GM_ERR_BAD_ARG;
    GM_HIP(hipHostUnregister(host));
    return GM_OK;
}

int gm_device_alloc(int device, uint64_t bytes, void** dptr)
{
    if (!dptr) return GM_ERR_BAD_ARG;
    int rc = select_device(device); if (rc) return rc;
    GM_HIP(hipMalloc(dptr, bytes));
    GM_HIP(hipMemset(*dptr, 0, bytes));
    GM_HIP(hipDeviceSynchronize());   // the clear is finished before the handle can reach another process
    return GM_OK;
}
int gm_device_free(int device, void* dptr) { GM_HIP(hipSetDevice(device)); GM_HIP(hipFree(dptr)); return GM_OK; }
int gm_ipc_export(int device, void* dptr, uint8_t handle[64])
{
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!dptr || !handle) return GM_ERR_BAD_ARG;
    GM_HIP(hipSetDevice(device));
    hipIpcMemHandle_t h;
    GM_HIP(hipIpcGetMemHandle(&h, dptr));
    memcpy(handle, &h, 64);
    return GM_OK;
}
int gm_ipc_open(int device, const uint8_t handle[64], void** dptr)
{
    if (!dptr || !handle) return GM_ERR_BAD_ARG;
    GM_HIP(hipSetDevice(device));
    hipIpcMemHandle_t h; memcpy(&h, handle, 64);
    GM_HIP(hipIpcOpenMemHandle(dptr, h, hipIpcMemLazyEnablePeerAccess));
    return GM_OK;
}
int gm_ipc_close(int device, void* dptr) { GM_HIP(hipSetDevice(device)); GM_HIP(hipIpcCloseMemHandle(dptr)); return GM_OK; }
int gm_push_pieces(int device, void* dst, const void* src, uint64_t first_byte, uint64_t pitch_bytes, uint64_t piece_bytes, uint64_t n_rows,
                   uint64_t last_piece_bytes, void* stream)
{
    if (!dst || !src) return GM_ERR_BAD_ARG;
    GM_HIP(hipSetDevice(device));
    for (uint64_t i = 0; i < n_rows; ++i) {
        const uint64_t off = first_byte + i * pitch_bytes, len = (i + 1 == n_rows && last_piece_bytes) ? last_piece_bytes : piece_bytes;
        GM_HIP(hipMemcpyAsync((uint8_t*)dst + off, (const uint8_t*)src + off, len, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return GM_OK;
}

}  // extern "C"

namespace gm {
// Device -> pageable host memory at more than the runtime's staging rate: DMA into a page-locked ring of four 64-MiB slots
// on `st` (the source must be ready on that stream), and while the next slots fill, host threads copy a finished slot to
// its destination.  Blocks until everything has arrived.
static int staged_copy_to_host(gm_index* ix, uint8_t* h_dst, const uint8_t* d_src, uint64_t bytes, hipStream_t st)
{
    constexpr uint64_t SLOT = 64ull << 20;
    constexpr int NS = 4;
    if (!ix->h_stage) {
        GM_HIP(hipHostMalloc(&ix->h_stage, SLOT * NS, hipHostMallocDefault));
        for (auto& e : ix->evStage) GM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int64_t nsub = (int64_t)((bytes + SLOT - 1) / SLOT);
    auto len_of = [&](int64_t k) { return std::min<uint64_t>(SLOT, bytes - (uint64_t)k * SLOT); };
    for (int64_t k = 0; k < nsub + NS - 1; ++k) {
        if (k < nsub) {
            GM_HIP(hipMemcpyAsync(ix->h_stage + (k % NS) * SLOT, d_src + (uint64_t)k * SLOT, len_of(k), hipMemcpyDeviceToHost, st));
            GM_HIP(hipEventRecord(ix->evStage[k % NS], st));
        }
        const int64_t j = k - (NS - 1);   // the slot that must be free before issue k + 1 reuses it
        if (j >= 0 && j < nsub) {
            GM_HIP(hipEventSynchronize(ix->evStage[j % NS]));
            const uint8_t* src = ix->h_stage + (j % NS) * SLOT;
            uint8_t* dst = h_dst + (uint64_t)j * SLOT;
            const uint64_t len = len_of(j);
            // (the destination is usually fresh pageable memory: the copy is bound by its first-touch page faults, which scale with the
            //  threads that take them -- 2 GB of csv locations per pass of config C5: 16 threads instead of 4)
            constexpr int TMAX = 16;
            const int T = len >= (8ull << 20) ? (int)std::min<unsigned>(TMAX, std::max(4u, std::thread::hardware_concurrency() / 4u)) : 4;
            std::thread th[TMAX - 1];
            for (int t = 1; t < T; ++t) th[t - 1] = std::thread([=] { memcpy(dst + len * t / T, src + len * t / T, len * (t + 1) / T - len * t / T); });
            memcpy(dst, src, len / T);
            for (int t = 1; t < T; ++t) th[t - 1].join();
        }
    }
    return GM_OK;
}
static bool host_memory_is_pinned(const void* p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}
}  // namespace gm

extern "C" {

int gm_map_shard(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
                 const uint64_t* intervals, uint64_t n_intervals, const uint32_t* seq_file_id, void* out_host)
{
    if (!ix || !p || !out_host) { set_error("null argument"); return GM_ERR_BAD_ARG; }
    if (p->value_bits != 8 && p->value_bits != 16) return GM_ERR_BAD_VALUE_BITS;
    if (p->K < 1 || p->K > MAX_K_LONG) return GM_ERR_BAD_K;
    GM_HIP(hipSetDevice(ix->device));
    ix->piece.active = false;   // (its own pieces go through map_impl directly)
    const size_t eb = p->value_bits / 8;
    if (!ix->stCompute) {
        GM_HIP(hipStreamCreateWithFlags(&ix->stCompute, hipStreamNonBlocking));
        GM_HIP(hipStreamCreateWithFlags(&ix->stCopy, hipStreamNonBlocking));
        for (auto& e : ix->evShard) GM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (ix->shardOutCap < text_len * eb + 16) {
        hipFree(ix->d_shardOut); ix->d_shardOut = nullptr; ix->shardOutCap = 0;
        GM_HIP(hipMalloc(&ix->d_shardOut, text_len * eb + 16));
        ix->shardOutCap = text_len * eb + 16;
    }
    uint8_t* d_out = (uint8_t*)ix->d_shardOut;
    uint8_t* h_out = (uint8_t*)out_host;
    const uint64_t numKmers = text_len >= p->K ? text_len - p->K + 1 : 0;
    const bool ranged = (p->flags & GM_MAP_FLAG_RANGE) || p->kmer_begin != 0 || p->kmer_end != 0;
    const uint64_t kb = ranged ? std::min<uint64_t>(p->kmer_begin, numKmers) : 0, ke = ranged ? std::min<uint64_t>(p->kmer_end, numKmers) : numKmers;
    const bool chunked = p->chunk_blocks > 0 && p->chunk_stride > 1;
    if (n_intervals > 0 || !chunked) {
        // a selection, or a plain contiguous share (gm_map is the share "everything")
        const uint32_t infix = long_k_infix(p->K, p->infix > 0 ? (uint32_t)p->infix : (p->overlap >= 0 ? default_infix_length(p->K, p->E, p->overlap) : tuned_infix_length(p->K, p->E, p->exclude_pseudo != 0 || p->K > MAX_K)));
        if (infix == 0 || infix > p->K) return GM_ERR_BAD_OVERLAP;
        const uint64_t step = p->K - infix + 1;
        // whole blocks: the share ends where the next one begins; the tail past the last k-mer is all zeros (resetLimits)
        const uint64_t b = (kb + step - 1) / step * step, e = ke >= numKmers ? text_len : (ke + step - 1) / step * step;
        // large shares without a selection go in a few launches so that the copy of one piece overlaps the search of the next
        const uint32_t S = (n_intervals == 0 && e > b && e - b >= (1ull << 26)) ? 4u : 1u;
        const bool pinned = S > 1 && host_memory_is_pinned(h_out);
        uint64_t pbs[4] = {0, 0, 0, 0}, pes[4] = {0, 0, 0, 0};
        ix->pieceIndex = 0;
        for (uint32_t s2 = 0; s2 < S; ++s2) {   // every launch is queued before the first piece is collected
            gm_map_params q = *p;
            uint64_t pb = b, pe = e;
            if (S > 1) {
                pb = b + (e - b) * s2 / S / step * step; pe = s2 + 1 == S ? e : b + (e - b) * (s2 + 1) / S / step * step;
                q.flags |= GM_MAP_FLAG_RANGE; q.kmer_begin = pb; q.kmer_end = std::min<uint64_t>(pe, ke);
                if (pe <= pb) continue;
            }
            uint64_t wrote[2] = {0, 0};
            const uint64_t whole[2] = {b, std::min<uint64_t>(e, ke)};   // (S > 1: the pieces of ONE call -- one clear, one correction pass)
            int rc = map_impl(ix, text_begin, text_len, first_seq, n_seq, &q, intervals, n_intervals, seq_file_id, d_out, ix->stCompute, wrote, S > 1 ? whole : nullptr);
            ix->pieceIndex += 1;
            if (rc) { ix->pieceIndex = 0; return rc; }
            // A selection's blocks begin at interval starts, not at multiples of the block length: the share delivers exactly the
            // positions its blocks span (what map_impl wrote; d_out is not cleared elsewhere), never the step-rounded window --
            // a block belongs to the share that holds its first k-mer, so the shares' spans are disjoint.  Unselected positions
            // outside every span are zero by definition and stay the caller's.
            if (n_intervals > 0) { pb = wrote[0]; pe = wrote[1]; }
            pbs[s2] = pb; pes[s2] = pe;
            GM_HIP(hipEventRecord(ix->evShard[s2], ix->stCompute));
        }
        ix->pieceIndex = 0;
        for (uint32_t s2 = 0; s2 < S; ++s2) {
            if (pes[s2] <= pbs[s2]) continue;
            GM_HIP(hipStreamWaitEvent(ix->stCopy, ix->evShard[s2], 0));
            if (pinned || S == 1) GM_HIP(hipMemcpyAsync(h_out + pbs[s2] * eb, d_out + pbs[s2] * eb, (pes[s2] - pbs[s2]) * eb, hipMemcpyDeviceToHost, ix->stCopy));
            else { int rc = staged_copy_to_host(ix, h_out + pbs[s2] * eb, d_out + pbs[s2] * eb, (pes[s2] - pbs[s2]) * eb, ix->stCopy); if (rc) return rc; }
        }
        GM_HIP(hipStreamSynchronize(ix->stCopy));
        GM_HIP(hipStreamSynchronize(ix->stCompute));
        return check_device_error(ix);
    }
    const uint32_t infix = long_k_infix(p->K, p->infix > 0 ? (uint32_t)p->infix : (p->overlap >= 0 ? default_infix_length(p->K, p->E, p->overlap) : tuned_infix_length(p->K, p->E, p->exclude_pseudo != 0 || p->K > MAX_K)));
    if (infix == 0 || infix > p->K) return GM_ERR_BAD_OVERLAP;
    const uint64_t step = p->K - infix + 1, chunkLen = (uint64_t)p->chunk_blocks * step, rowLen = chunkLen * p->chunk_stride;
    const uint64_t base = (kb + step - 1) / step * step;                       // first block of the range
    const uint64_t span = ke > base ? ke - base : 0;
    const uint64_t rows = (span + rowLen - 1) / rowLen;                        // one chunk of every shard per row
    const uint32_t S = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(rows, 1), 4);   // launches: copy of launch s overlaps compute of s + 1
    ix->pieceIndex = 0;
    for (uint32_t s = 0; s < S; ++s) {
        const uint64_t r0 = rows * s / S, r1 = rows * (s + 1) / S;
        if (r1 <= r0) continue;
        gm_map_params q = *p;
        q.flags |= GM_MAP_FLAG_RANGE;
        q.kmer_begin = base + r0 * rowLen; q.kmer_end = std::min<uint64_t>(base + r1 * rowLen, ke);   // rows start at multiples of the row length: chunk numbers keep their residue
        const uint64_t whole[2] = {base, ke};
        int rc = map_impl(ix, text_begin, text_len, first_seq, n_seq, &q, nullptr, 0, seq_file_id, d_out, ix->stCompute, nullptr, S > 1 ? whole : nullptr);
        ix->pieceIndex += 1;
        if (rc) { ix->pieceIndex = 0; return rc; }
        GM_HIP(hipEventRecord(ix->evShard[s], ix->stCompute));
        GM_HIP(hipStreamWaitEvent(ix->stCopy, ix->evShard[s], 0));
        // own chunks of rows [r0, r1): a strided copy; the last row may hold a short (or no) chunk of this shard
        const uint64_t first = base + r0 * rowLen + (uint64_t)p->chunk_index * chunkLen;
        uint64_t full = r1 - r0;
        const uint64_t lastBegin = first + (full - 1) * rowLen;
        const uint64_t limit = ke >= numKmers ? text_len : ke;                 // the final shard also delivers the zeroed tail
        uint64_t lastLen = lastBegin >= limit ? 0 : std::min<uint64_t>(chunkLen, limit - lastBegin);
        if (lastLen < chunkLen) full -= 1; else lastLen = 0;
        if (full > 0) GM_HIP(hipMemcpy2DAsync(h_out + first * eb, rowLen * eb, d_out + first * eb, rowLen * eb, chunkLen * eb, full, hipMemcpyDeviceToHost, ix->stCopy));
        if (lastLen > 0) GM_HIP(hipMemcpyAsync(h_out + lastBegin * eb, d_out + lastBegin * eb, lastLen * eb, hipMemcpyDeviceToHost, ix->stCopy));
    }
    ix->pieceIndex = 0;
    // the tail past the last k-mer (K - 1 zeros) belongs to the chunk that holds position numKmers - 1 when that chunk is short;
    // otherwise it lies in later chunk slots nobody owns: the shard owning the LAST chunk delivers it
    if (ke >= numKmers && numKmers > 0 && base < numKmers) {   // (a range that starts in the last partial block holds no whole block: nothing to deliver)
        const uint64_t lastChunk = (numKmers - 1 - base) / chunkLen;
        if (lastChunk % p->chunk_stride == p->chunk_index) {
            const uint64_t from = base + (lastChunk + 1) * chunkLen;
            if (from < text_len) GM_HIP(hipMemcpyAsync(h_out + from * eb, d_out + from * eb, (text_len - from) * eb, hipMemcpyDeviceToHost, ix->stCopy));
        }
    }
    GM_HIP(hipStreamSynchronize(ix->stCopy));
    GM_HIP(hipStreamSynchronize(ix->stCompute));
    return check_device_error(ix);
}

int gm_locate(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
              const uint64_t* intervals, uint64_t n_intervals, gm_locations** out)
{
    if (!out) { set_error("null argument"); return GM_ERR_BAD_ARG; }
    gm_locations* L = (gm_locations*)calloc(1, sizeof(gm_locations));
    if (!L) return GM_ERR_OOM;
    int rc = locate_impl(ix, text_begin, text_len, first_seq, n_seq, p, intervals, n_intervals, L);
    if (rc) { gm_locations_free(L); return rc; }
    *out = L;
    return GM_OK;
}

void gm_locations_free(gm_locations* L)
{
    if (!L) return;
    free(L->plus_off); free(L->minus_off); free(L->plus); free(L->minus); free(L);
}

int gm_map_runs(gm_index* ix, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq, const gm_map_params* p,
                const uint64_t* intervals, uint64_t n_intervals, const uint32_t* seq_file_id, gm_runs** out)
{
    if (!ix || !p || !out) { set_error("null argument"); return GM_ERR_BAD_ARG; }
    if (p->value_bits != 8 && p->value_bits != 16) return GM_ERR_BAD_VALUE_BITS;
    GM_HIP(hipSetDevice(ix->device));
    if (text_len >= 0xFFFFFFFFull) { set_error("gm_map_runs addresses 32-bit slice positions: split the file or use gm_map"); return GM_ERR_TOO_LONG; }
    gm_runs* R = (gm_runs*)calloc(1, sizeof(gm_runs));
    if (!R) return GM_ERR_OOM;
    void* d_c = nullptr; uint8_t* d_head = nullptr; uint32_t *d_starts = nullptr, *d_count = nullptr; uint16_t* d_val = nullptr; void* d_tmp = nullptr;
    size_t tmpBytes = 0;
    int rc = GM_OK;
#define RC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); rc = (e_ == hipErrorOutOfMemory) ? GM_ERR_OOM : GM_ERR_HIP; goto done; } } while (0)
    {
        const uint64_t n = text_len;
        RC(hipMalloc(&d_c, n * (p->value_bits / 8) + 16));
        rc = map_impl(ix, text_begin, text_len, first_seq, n_seq, p, intervals, n_intervals, seq_file_id, d_c, nullptr);
        if (rc) goto done;
        if (n > 0) {
            RC(hipMalloc(&d_head, n)); RC(hipMalloc(&d_starts, (n + 1) * 4)); RC(hipMalloc(&d_count, 4));
            if (p->value_bits == 8) hipLaunchKernelGGL(run_heads_kernel<uint8_t>, dim3(grid_for(n)), dim3(256), 0, 0, (const uint8_t*)d_c, n, d_head);
            else hipLaunchKernelGGL(run_heads_kernel<uint16_t>, dim3(grid_for(n)), dim3(256), 0, 0, (const uint16_t*)d_c, n, d_head);
            hipLaunchKernelGGL(seq_heads_kernel, dim3(grid_for(n_seq)), dim3(256), 0, 0, ix->d_cumLocal, n_seq, n, d_head);
            rocprim::counting_iterator<uint32_t> iota(0);
            RC(rocprim::select(nullptr, tmpBytes, iota, d_head, d_starts, d_count, n));
            RC(hipMalloc(&d_tmp, tmpBytes ? tmpBytes : 16));
            RC(rocprim::select(d_tmp, tmpBytes, iota, d_head, d_starts, d_count, n));
            uint32_t nHeads = 0;
            RC(hipMemcpy(&nHeads, d_count, 4, hipMemcpyDeviceToHost));
            RC(hipMalloc(&d_val, ((size_t)nHeads + 1) * 2));
            if (p->value_bits == 8) hipLaunchKernelGGL(run_values_kernel<uint8_t>, dim3(grid_for(nHeads)), dim3(256), 0, 0, (const uint8_t*)d_c, d_starts, nHeads, d_val);
            else hipLaunchKernelGGL(run_values_kernel<uint16_t>, dim3(grid_for(nHeads)), dim3(256), 0, 0, (const uint16_t*)d_c, d_starts, nHeads, d_val);
            std::vector<uint32_t> starts(nHeads); std::vector<uint16_t> vals(nHeads);
            RC(hipMemcpy(starts.data(), d_starts, (size_t)nHeads * 4, hipMemcpyDeviceToHost));
            RC(hipMemcpy(vals.data(), d_val, (size_t)nHeads * 2, hipMemcpyDeviceToHost));
            uint64_t keep = 0;
            for (uint32_t r = 0; r < nHeads; ++r) keep += vals[r] != 0;
            R->start = (uint64_t*)malloc((keep + 1) * 8); R->length = (uint64_t*)malloc((keep + 1) * 8); R->value = (uint16_t*)malloc((keep + 1) * 2);
            if (!R->start || !R->length || !R->value) { rc = GM_ERR_OOM; goto done; }
            uint64_t k = 0;
            for (uint32_t r = 0; r < nHeads; ++r) {
                if (vals[r] == 0) continue;   // runs of 0 are never written (src/output.hpp:98,152)
                R->start[k] = starts[r]; R->length[k] = (r + 1 < nHeads ? starts[r + 1] : n) - starts[r]; R->value[k] = vals[r]; ++k;
            }
            R->n_runs = k;
        }
        rc = check_device_error(ix);
    }
done:
#undef RC
    hipFree(d_c); hipFree(d_head); hipFree(d_starts); hipFree(d_count); hipFree(d_val); hipFree(d_tmp);
    if (rc) { gm_runs_free(R); return rc; }
    *out = R;
    return GM_OK;
}

void gm_runs_free(gm_runs* R)
{
    if (!R) return;
    free(R->start); free(R->length); free(R->value); free(R);
}

int gm_index_set_tuning(gm_index* ix, const char* name, int64_t value)
{
    if (!ix || !name) return GM_ERR_BAD_ARG;
    const Tuning dflt;   // -1 restores these (for part_bias, which may be negative, -1 is a value: 0 is its default)
    struct { const char* n; int* f; int d; int64_t lo, hi; } tab[] = {
        {"verify_t", &ix->tune.verifyT, dflt.verifyT, 0, (int64_t)VERIFY_TMAX}, {"lds_stack", &ix->tune.ldsStack, dflt.ldsStack, 0, 64},
        {"blocks_per_cu", &ix->tune.blocksPerCU, dflt.blocksPerCU, 1, 8}, {"qtable", &ix->tune.qtable, dflt.qtable, 0, 16},
        {"sat_min_w", &ix->tune.satMinW, dflt.satMinW, 1, 0x7FFFFFFF}, {"fetch_batch", &ix->tune.fetchBatch, dflt.fetchBatch, 1, 64},
        {"probation", &ix->tune.probation, dflt.probation, 0, 255}, {"verify_cost", &ix->tune.verifyCost, dflt.verifyCost, 0, 1 << 20},
        {"no_store", &ix->tune.noStore, dflt.noStore, 0, 1}, {"no_saturate", &ix->tune.noSaturate, dflt.noSaturate, 0, 1},
        {"skip_dup", &ix->tune.skipDup, dflt.skipDup, 0, 1}, {"coop", &ix->tune.coop, dflt.coop, 0, 1}, {"use_ctx", &ix->tune.useCtx, dflt.useCtx, 0, 1},
        {"steal", &ix->tune.steal, dflt.steal, 0, 64}, {"part_bias", &ix->tune.partBias, dflt.partBias, -255, 255},
        {"child_tables", &ix->tune.childTables, dflt.childTables, 0, 1}, {"oss_weights", &ix->tune.ossWeights, dflt.ossWeights, 0, 0xFFFFFF},   // (-1: 5,4,7,8 at e = 2, the even split elsewhere)
        {"jump", &ix->tune.jump, dflt.jump, 0, 16}, {"self_hit", &ix->tune.selfHit, dflt.selfHit, 0, 1}, {"jump_filter", &ix->tune.jumpFilter, dflt.jumpFilter, 0, 2},
        {"range_add", &ix->tune.rangeAdd, dflt.rangeAdd, 0, 1}, {"verify_t_ext", &ix->tune.verifyTExt, dflt.verifyTExt, 0, (int64_t)VERIFY_TMAX},
        {"fast_verify", &ix->tune.fastVerify, dflt.fastVerify, 0, 1}, {"lds_pad", &ix->tune.ldsPad, dflt.ldsPad, 0, 65536},
        {"jump_layouts", &ix->tune.jumpLayouts, dflt.jumpLayouts, 0, 1},   // 0: groups of jump patterns in the LOW / MID layouts only (round 4), -1 / 1: at any three adjacent characters
        {"no_wrap", &ix->tune.noWrap, dflt.noWrap, 0, 1},   // 0: every add into an accumulator returns the old value and checks for a wrap-around (-1 / 1: only where one is possible)
        {"iter_cap", &ix->tune.iterCap, dflt.iterCap, 1, 0x7FFFFFFF}, {"stall_cap", &ix->tune.stallCap, dflt.stallCap, 1, 0x7FFFFFFF},   // bounds of a hung search loop (tests force them)
        {"pat_batch", &ix->tune.patBatch, dflt.patBatch, 1, 64},
        {"expand", &ix->tune.expand, dflt.expand, 0, 1}, {"expand_mb", &ix->tune.expandMB, dflt.expandMB, 1, 1 << 20},   // the split search (gm_expand.h)
        {"expand_chunk", &ix->tune.expandChunk, dflt.expandChunk, 1, 1 << 16}, {"expand_occ", &ix->tune.expandOcc, dflt.expandOcc, 1, 64}, {"expand_overlap", &ix->tune.expandOverlap, dflt.expandOverlap, 0, 1}, {"expand_two_pass", &ix->tune.expandTwoPass, dflt.expandTwoPass, 0, 1}, {"expand_share", &ix->tune.expandShare, dflt.expandShare, 0, 1}, {"sat_draw_w", &ix->tune.satDrawW, dflt.satDrawW, 0, 0x7FFFFFFF},
        {"win2", &ix->tune.win2, dflt.win2, 0, 1},   // needle windows at 2 bits per symbol
        {"jump_groups", &ix->tune.jumpGroups, dflt.jumpGroups, 0, 1},   // groups of jump patterns behind the existence bitmap: 0 never, 1 wherever possible, -1 where they save table reads
    };
    for (auto& t : tab) if (!strcmp(t.n, name)) {
        const bool isBias = t.f == &ix->tune.partBias;
        if (value == -1 && !isBias) { *t.f = t.d; return GM_OK; }
        if (value < t.lo || value > t.hi) { set_error("tuning knob '%s': value %lld outside [%lld, %lld]", name, (long long)value, (long long)t.lo, (long long)t.hi); return GM_ERR_BAD_ARG; }
        *t.f = (int)value;
        return GM_OK;
    }
    set_error("unknown tuning knob '%s'", name);
    return GM_ERR_BAD_ARG;
}

int gm_index_sync(gm_index* ix)
{
    if (!ix) return GM_ERR_BAD_ARG;
    GM_HIP(hipSetDevice(ix->device));
    if (ix->doneValid) GM_HIP(hipEventSynchronize(ix->evDone));
    return check_device_error(ix);
}

int gm_map_kernel_times(const gm_index* ix, double* ms, uint32_t n, uint32_t* n_out)
{
    if (!ix || !ms || !n_out) return GM_ERR_BAD_ARG;
    GM_HIP(hipSetDevice(ix->device));
    const uint32_t have = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(ix->evCount, gm_index::EV_RING), n);
    for (uint32_t i = 0; i < have; ++i) {
        const uint32_t slot = (uint32_t)((ix->evCount - have + i) % gm_index::EV_RING);
        GM_HIP(hipEventSynchronize(ix->evRing[slot][1]));
        float t = 0;
        GM_HIP(hipEventElapsedTime(&t, ix->evRing[slot][0], ix->evRing[slot][1]));
        ms[i] = t;
    }
    *n_out = have;
    return GM_OK;
}

int gm_last_map_stats(const gm_index* cix, gm_map_stats* out)
{
    gm_index* ix = const_cast<gm_index*>(cix);
    if (!ix || !out) return GM_ERR_BAD_ARG;
    if (!ix->evValid) { *out = gm_map_stats{}; return GM_OK; }
    GM_HIP(hipSetDevice(ix->device));
    GM_HIP(hipEventSynchronize(ix->ev[3]));
    float a = 0, b = 0;
    for (uint32_t k = 0; k < std::min<uint32_t>(std::max<uint32_t>(ix->statPieces, 1u), gm_index::EV_RING); ++k) {   // a call delivered in pieces: the sum of its launches
        float t = 0;
        const uint32_t slot = (uint32_t)((ix->evCount - 1 - k) % gm_index::EV_RING);
        GM_HIP(hipEventElapsedTime(&t, ix->evRing[slot][0], ix->evRing[slot][1]));
        a += t;
    }
    GM_HIP(hipEventElapsedTime(&b, ix->ev[0], ix->ev[3]));
    ix->stats.search_ms = a; ix->stats.total_ms = b;
    unsigned long long cnt[50] = {0};
    GM_HIP(hipMemcpy(cnt, reinterpret_cast<char*>(ix->d_small) + 16, sizeof(cnt), hipMemcpyDeviceToHost));
    ix->stats.node_steps = cnt[0]; ix->stats.rank_lines = cnt[1];
    for (int i = 0; i < 48; ++i) ix->stats.detail[i] = cnt[2 + i];
    ix->stats.detail[38] = ix->lastQ;   // longest q-mer table of the call | jump length << 8
    ix->stats.detail[47] = (ix->stats.detail[47] & 0xFFFFFFFFFFFFull) | (uint64_t)ix->lastSlices << 48;   // node packets of the split search (counters) | its slices (every build)
    if (ix->corrTimed) { float c = 0; GM_HIP(hipEventElapsedTime(&c, ix->ev[1], ix->ev[2])); ix->stats.detail[37] = (uint64_t)(c * 1000.0f); }   // correction pass, microseconds
    int rc = check_device_error(ix);
    *out = ix->stats;
    return rc;
}

}  // extern "C"
