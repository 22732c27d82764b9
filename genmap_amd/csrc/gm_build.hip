// gm_build.hip -- GPU construction of the bidirectional FM index (gfx950).
//
// Replaces the CPU index construction of the reference (genmap index):
//   /root/reference/src/indexing.hpp:73-148        fwd index, reverse(text), rev index
//   /root/reference/src/seqan_libdivsufsort.h:36-240  SA (libdivsufsort) -> BWT, sentinels, SA samples
// with an MI355X-first design: the whole suffix array lives in HBM (288 GB: 32 B per text symbol is
// affordable even for 3.1 Gbp) and is sorted by PREFIX DOUBLING, every round being one rocPRIM
// device-wide radix sort of (rank[i], rank[i+h]) pairs plus streaming kernels.  Conventions kept from the
// reference because locate results depend on them: one sentinel after every sequence, sentinels smaller
// than every letter and ordered by position (seqan_libdivsufsort.h:80-91,121-123); the reverse index is
// built on each sequence reversed, same sequence order (indexing.hpp:130).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <vector>
#include <cstdio>
#include "gm_internal.h"

namespace gm {

// ---- sentinel text -------------------------------------------------------------------------------------
// sym[p] (codes 0..5) and key[p] (sentinel s -> s, letter c -> nSeq + c) for the forward or reversed text.
__global__ __launch_bounds__(256) void make_symbols_kernel(const uint8_t* __restrict__ codes, const uint64_t* __restrict__ cum,
                                                           uint32_t nSeq, uint64_t textLen, int rev,
                                                           uint8_t* __restrict__ sym, uint32_t* __restrict__ key, uint32_t* __restrict__ sa)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // sentinel-free position, or textLen + s for sentinels
    if (i < textLen) {
        uint32_t lo = 0, hi = nSeq;   // sequence containing i: last s with cum[s] <= i
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cum[mid] <= i) lo = mid; else hi = mid; }
        const uint64_t b = cum[lo], e = cum[lo + 1];
        const uint64_t p = (rev ? (b + (e - 1 - i)) : i) + lo;
        const uint32_t c = codes[i];
        sym[p] = (uint8_t)c; key[p] = nSeq + c; sa[p] = (uint32_t)p;
    } else if (i < textLen + nSeq) {
        const uint32_t s = (uint32_t)(i - textLen);
        const uint64_t p = cum[s + 1] + s;
        sym[p] = (uint8_t)SYM_SENT; key[p] = s; sa[p] = (uint32_t)p;
    }
}

__global__ __launch_bounds__(256) void head_flags32_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ flag, uint64_t n)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) flag[j] = (j > 0 && keys[j] != keys[j - 1]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void head_flags64_kernel(const uint64_t* __restrict__ keys, uint32_t* __restrict__ flag, uint64_t n)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) flag[j] = (j > 0 && keys[j] != keys[j - 1]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void scatter_rank_kernel(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ dense, uint32_t* __restrict__ rank, uint64_t n)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) rank[sa[j]] = dense[j];
}
__global__ __launch_bounds__(256) void make_keys_kernel(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ rank, uint64_t* __restrict__ keys,
                                                        uint64_t n, uint64_t h)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        const uint64_t p = sa[j];
        const uint64_t r1 = rank[p];
        // a suffix shorter than h ends in a (unique) sentinel and already has a unique rank: second key irrelevant
        const uint64_t r2 = (p + h < n) ? rank[p + h] : 0ull;
        keys[j] = r1 << 32 | r2;
    }
}
__global__ __launch_bounds__(256) void bwt_kernel(const uint32_t* __restrict__ sa, const uint8_t* __restrict__ sym, uint8_t* __restrict__ bwt, uint64_t n)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) { const uint32_t p = sa[j]; bwt[j] = p ? sym[p - 1] : sym[n - 1]; }
}

static inline unsigned grid_for(uint64_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }
// for grid-stride kernels: at most 2^30 work-items per launch (a dispatch holds fewer than 2^32 per dimension)
static inline unsigned grid_cap(uint64_t n, unsigned bs = 256) { return (unsigned)std::min<uint64_t>((n + bs - 1) / bs, 1u << 22); }

// Suffix array of the sentinel text (forward or reversed) in d_sa_out (n x u32), its BWT in d_bwt (n x u8).
// Both are caller-provided device buffers; everything else is allocated and freed here.
int build_sa_bwt(const uint8_t* d_codes, const uint64_t* d_cum, uint32_t nSeq, uint64_t textLen, int rev,
                 uint32_t* d_sa_out, uint8_t* d_bwt, int* roundsOut)
{
    const uint64_t n = textLen + nSeq;
    uint8_t* d_sym = nullptr;
    uint32_t *d_key32 = nullptr, *d_key32b = nullptr, *d_own = nullptr, *d_rank = nullptr, *d_flag = nullptr, *d_dense = nullptr;
    uint64_t *d_k64 = nullptr, *d_k64b = nullptr;
    void* d_tmp = nullptr;
    size_t tmpBytes = 0;
    int rc = GM_OK;
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); rc = (e_ == hipErrorOutOfMemory ? GM_ERR_OOM : GM_ERR_HIP); goto done; } } while (0)
    {
        uint32_t* cur = d_sa_out;   // holds the current order
        uint32_t* alt = nullptr;    // sort output
        HC(hipMalloc(&d_sym, n));
        HC(hipMalloc(&d_own, n * 4)); HC(hipMalloc(&d_rank, n * 4));
        HC(hipMalloc(&d_flag, n * 4)); HC(hipMalloc(&d_dense, n * 4));
        HC(hipMalloc(&d_k64, n * 8)); HC(hipMalloc(&d_k64b, n * 8));
        d_key32 = reinterpret_cast<uint32_t*>(d_k64);    // round 0 keys alias the 64-bit key buffers
        d_key32b = reinterpret_cast<uint32_t*>(d_k64b);
        alt = d_own;
        hipLaunchKernelGGL(make_symbols_kernel, dim3(grid_for(n)), dim3(256), 0, 0, d_codes, d_cum, nSeq, textLen, rev, d_sym, d_key32, cur);
        HC(hipGetLastError());

        size_t t1 = 0, t2 = 0, t3 = 0;
        unsigned keyBits = 1; while ((1ull << keyBits) < (uint64_t)nSeq + NLET) ++keyBits;
        unsigned rbits = 1; while ((1ull << rbits) < n) ++rbits;
        HC(rocprim::radix_sort_pairs(nullptr, t1, d_key32, d_key32b, cur, alt, n, 0, keyBits));
        HC(rocprim::radix_sort_pairs(nullptr, t2, d_k64, d_k64b, cur, alt, n, 0, 32 + rbits));
        HC(rocprim::inclusive_scan(nullptr, t3, d_flag, d_dense, n, rocprim::plus<uint32_t>()));
        tmpBytes = std::max(t1, std::max(t2, t3));
        HC(hipMalloc(&d_tmp, tmpBytes ? tmpBytes : 16));

        // round 0: order by first symbol (sentinels are unique symbols ordered by sequence number)
        size_t tb = tmpBytes;
        HC(rocprim::radix_sort_pairs(d_tmp, tb, d_key32, d_key32b, cur, alt, n, 0, keyBits));
        std::swap(cur, alt);
        hipLaunchKernelGGL(head_flags32_kernel, dim3(grid_for(n)), dim3(256), 0, 0, d_key32b, d_flag, n);
        tb = tmpBytes;
        HC(rocprim::inclusive_scan(d_tmp, tb, d_flag, d_dense, n, rocprim::plus<uint32_t>()));
        hipLaunchKernelGGL(scatter_rank_kernel, dim3(grid_for(n)), dim3(256), 0, 0, cur, d_dense, d_rank, n);
        uint32_t maxRank = 0;
        HC(hipMemcpy(&maxRank, d_dense + (n - 1), 4, hipMemcpyDeviceToHost));
        int rounds = 0;
        for (uint64_t h = 1; maxRank != (uint32_t)(n - 1); h <<= 1) {
            if (h >= n) { set_error("prefix doubling did not converge"); rc = GM_ERR_INTERNAL; goto done; }
            hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(n)), dim3(256), 0, 0, cur, d_rank, d_k64, n, h);
            tb = tmpBytes;
            HC(rocprim::radix_sort_pairs(d_tmp, tb, d_k64, d_k64b, cur, alt, n, 0, 32 + rbits));
            std::swap(cur, alt);
            hipLaunchKernelGGL(head_flags64_kernel, dim3(grid_for(n)), dim3(256), 0, 0, d_k64b, d_flag, n);
            tb = tmpBytes;
            HC(rocprim::inclusive_scan(d_tmp, tb, d_flag, d_dense, n, rocprim::plus<uint32_t>()));
            hipLaunchKernelGGL(scatter_rank_kernel, dim3(grid_for(n)), dim3(256), 0, 0, cur, d_dense, d_rank, n);
            HC(hipMemcpy(&maxRank, d_dense + (n - 1), 4, hipMemcpyDeviceToHost));
            ++rounds;
        }
        if (roundsOut) *roundsOut = rounds;
        if (cur != d_sa_out) HC(hipMemcpy(d_sa_out, cur, n * 4, hipMemcpyDeviceToDevice));
        hipLaunchKernelGGL(bwt_kernel, dim3(grid_for(n)), dim3(256), 0, 0, d_sa_out, d_sym, d_bwt, n);
        HC(hipGetLastError());
        HC(hipDeviceSynchronize());
    }
done:
    hipFree(d_sym); hipFree(d_own); hipFree(d_rank); hipFree(d_flag); hipFree(d_dense);
    hipFree(d_k64); hipFree(d_k64b); hipFree(d_tmp);
#undef HC
    return rc;
}

// ---- 64-bit rows ---------------------------------------------------------------------------------------------
// Same prefix doubling; ranks need more than 32 bits, so the (rank[i], rank[i+h]) key of a round no longer fits one 64-bit
// radix key: every round sorts twice, stably, LSD fashion -- by rank[i+h], then by rank[i].
__global__ __launch_bounds__(256) void make_symbols_wide_kernel(const uint8_t* __restrict__ codes, const uint64_t* __restrict__ cum,
                                                                uint32_t nSeq, uint64_t textLen, int rev,
                                                                uint8_t* __restrict__ sym, uint64_t* __restrict__ key, uint64_t* __restrict__ sa)
{
    // grid-stride: a launch holds fewer than 2^32 work-items per dimension, these texts have more symbols than that
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < textLen + nSeq; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i < textLen) {
            uint32_t lo = 0, hi = nSeq;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cum[mid] <= i) lo = mid; else hi = mid; }
            const uint64_t b = cum[lo], e = cum[lo + 1];
            const uint64_t p = (rev ? (b + (e - 1 - i)) : i) + lo;
            const uint32_t c = codes[i];
            sym[p] = (uint8_t)c; key[p] = (uint64_t)nSeq + c; sa[p] = p;
        } else {
            const uint32_t s = (uint32_t)(i - textLen);
            const uint64_t p = cum[s + 1] + s;
            sym[p] = (uint8_t)SYM_SENT; key[p] = s; sa[p] = p;
        }
    }
}
// flag[j] = 1 when suffix sa[j] starts a new group: its (first, second) rank pair differs from its predecessor's
__global__ __launch_bounds__(256) void head_flags_pair_kernel(const uint64_t* __restrict__ sa, const uint64_t* __restrict__ rank, uint64_t* __restrict__ flag, uint64_t n, uint64_t h)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        if (j == 0) { flag[0] = 0; continue; }
        const uint64_t p = sa[j], q = sa[j - 1];
        const uint64_t p2 = p + h < n ? rank[p + h] + 1 : 0, q2 = q + h < n ? rank[q + h] + 1 : 0;
        flag[j] = (rank[p] != rank[q] || p2 != q2) ? 1ull : 0ull;
    }
}
__global__ __launch_bounds__(256) void head_flags_key_kernel(const uint64_t* __restrict__ keys, uint64_t* __restrict__ flag, uint64_t n)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        flag[j] = (j > 0 && keys[j] != keys[j - 1]) ? 1ull : 0ull;
}
__global__ __launch_bounds__(256) void scatter_rank_wide_kernel(const uint64_t* __restrict__ sa, const uint64_t* __restrict__ dense, uint64_t* __restrict__ rank, uint64_t n)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) rank[sa[j]] = dense[j];
}
// key of a suffix for one of the two passes: which = 1 -> rank[i + h] (0 for suffixes shorter than h, they sort first), which = 0 -> rank[i]
__global__ __launch_bounds__(256) void make_key_wide_kernel(const uint64_t* __restrict__ sa, const uint64_t* __restrict__ rank, uint64_t* __restrict__ keys, uint64_t n, uint64_t h, int which)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = sa[j];
        keys[j] = which ? (p + h < n ? rank[p + h] + 1 : 0ull) : rank[p];
    }
}
__global__ __launch_bounds__(256) void bwt_wide_kernel(const uint64_t* __restrict__ sa, const uint8_t* __restrict__ sym, uint8_t* __restrict__ bwt, uint64_t n)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) { const uint64_t p = sa[j]; bwt[j] = p ? sym[p - 1] : sym[n - 1]; }
}

int build_sa_bwt_wide(const uint8_t* d_codes, const uint64_t* d_cum, uint32_t nSeq, uint64_t textLen, int rev,
                      uint64_t* d_sa_out, uint8_t* d_bwt, int* roundsOut)
{
    const uint64_t n = textLen + nSeq;
    uint8_t* d_sym = nullptr;
    uint64_t *d_own = nullptr, *d_rank = nullptr, *d_flag = nullptr, *d_keyA = nullptr, *d_keyB = nullptr;
    void* d_tmp = nullptr;
    size_t tmpBytes = 0;
    int rc = GM_OK;
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); rc = (e_ == hipErrorOutOfMemory ? GM_ERR_OOM : GM_ERR_HIP); goto done; } } while (0)
    {
        uint64_t* cur = d_sa_out;
        uint64_t* alt = nullptr;
        HC(hipMalloc(&d_sym, n));
        HC(hipMalloc(&d_own, n * 8)); HC(hipMalloc(&d_rank, n * 8)); HC(hipMalloc(&d_flag, n * 8));
        HC(hipMalloc(&d_keyA, n * 8)); HC(hipMalloc(&d_keyB, n * 8));
        alt = d_own;
        hipLaunchKernelGGL(make_symbols_wide_kernel, dim3(grid_cap(n)), dim3(256), 0, 0, d_codes, d_cum, nSeq, textLen, rev, d_sym, d_keyA, cur);
        HC(hipGetLastError());
        unsigned keyBits = 1; while ((1ull << keyBits) < (uint64_t)nSeq + NLET) ++keyBits;
        unsigned rbits = 1; while ((1ull << rbits) < n + 1) ++rbits;
        size_t t1 = 0, t2 = 0;
        HC(rocprim::radix_sort_pairs(nullptr, t1, d_keyA, d_keyB, cur, alt, n, 0, std::max(keyBits, rbits)));
        HC(rocprim::inclusive_scan(nullptr, t2, d_flag, d_keyA, n, rocprim::plus<uint64_t>()));
        tmpBytes = std::max(t1, t2);
        HC(hipMalloc(&d_tmp, tmpBytes ? tmpBytes : 16));
        // round 0: order by first symbol
        size_t tb = tmpBytes;
        HC(rocprim::radix_sort_pairs(d_tmp, tb, d_keyA, d_keyB, cur, alt, n, 0, keyBits));
        std::swap(cur, alt);
        hipLaunchKernelGGL(head_flags_key_kernel, dim3(grid_cap(n)), dim3(256), 0, 0, d_keyB, d_flag, n);
        tb = tmpBytes;
        HC(rocprim::inclusive_scan(d_tmp, tb, d_flag, d_keyA, n, rocprim::plus<uint64_t>()));   // dense ranks in d_keyA
        hipLaunchKernelGGL(scatter_rank_wide_kernel, dim3(grid_cap(n)), dim3(256), 0, 0, cur, d_keyA, d_rank, n);
        uint64_t maxRank = 0;
        HC(hipMemcpy(&maxRank, d_keyA + (n - 1), 8, hipMemcpyDeviceToHost));
        int rounds = 0;
        for (uint64_t h = 1; maxRank != n - 1; h <<= 1) {
            if (h >= n) { set_error("prefix doubling did not converge"); rc = GM_ERR_INTERNAL; goto done; }
            for (int which = 1; which >= 0; --which) {   // stable LSD: second key first
                hipLaunchKernelGGL(make_key_wide_kernel, dim3(grid_cap(n)), dim3(256), 0, 0, cur, d_rank, d_keyA, n, h, which);
                tb = tmpBytes;
                HC(rocprim::radix_sort_pairs(d_tmp, tb, d_keyA, d_keyB, cur, alt, n, 0, rbits));
                std::swap(cur, alt);
            }
            hipLaunchKernelGGL(head_flags_pair_kernel, dim3(grid_cap(n)), dim3(256), 0, 0, cur, d_rank, d_flag, n, h);
            tb = tmpBytes;
            HC(rocprim::inclusive_scan(d_tmp, tb, d_flag, d_keyA, n, rocprim::plus<uint64_t>()));
            hipLaunchKernelGGL(scatter_rank_wide_kernel, dim3(grid_cap(n)), dim3(256), 0, 0, cur, d_keyA, d_keyB, n);   // new ranks in d_keyB ...
            HC(hipMemcpyAsync(d_rank, d_keyB, n * 8, hipMemcpyDeviceToDevice, 0));                                          // ... (the flags read the old ones)
            HC(hipMemcpy(&maxRank, d_keyA + (n - 1), 8, hipMemcpyDeviceToHost));
            HC(hipGetLastError());
            ++rounds;
        }
        if (roundsOut) *roundsOut = rounds;
        if (cur != d_sa_out) HC(hipMemcpy(d_sa_out, cur, n * 8, hipMemcpyDeviceToDevice));
        hipLaunchKernelGGL(bwt_wide_kernel, dim3(grid_cap(n)), dim3(256), 0, 0, d_sa_out, d_sym, d_bwt, n);
        HC(hipGetLastError());
        HC(hipDeviceSynchronize());
    }
done:
    hipFree(d_sym); hipFree(d_own); hipFree(d_rank); hipFree(d_flag); hipFree(d_keyA); hipFree(d_keyB); hipFree(d_tmp);
#undef HC
    return rc;
}

}  // namespace gm
