// gather_bench2.hip -- what does a random rank-block read cost on MI355X beyond the TLB reach?  (measurement tool, not product)
//
// gather_bench.hip showed that random aligned reads fall from ~55 G blocks/s to a BYTE-proportional ~1.25 TB/s once the
// footprint passes ~4 GiB (39 G/s for 32 B, 20 G/s for 64 B, 10 G/s for 128 B, but 49 G/s for 16 B).  A lane reads a B-byte
// block with B/16 global_load_dwordx4 instructions, so "bytes" and "lane-load instructions" cannot be told apart there.
// This tool separates them: the same random blocks are read
//   solo   one lane reads the whole block (B/16 loads per lane per block)              -- what the search kernel did in round 1
//   coop   G = B/16 adjacent lanes read one block with ONE load each (lanes of a group take turns being the owner of the
//          block; the pieces are handed to the owner with DPP/ds_bpermute)             -- same bytes, 1/G of the lane-loads
// Every lane runs CH dependent chains (like range lo / range hi of a rank query).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

__global__ void fill_kernel(uint4* p, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { uint64_t h = mix(i); p[i] = make_uint4((uint32_t)h, (uint32_t)(h >> 32), (uint32_t)i, 7u); }
}

struct Tabs { const uint4* t[4]; uint64_t nblocks[4]; int n; };   // the footprint may be split over several allocations

template <int NV, int CH>
__global__ __launch_bounds__(256) void solo_kernel(Tabs T, int iters, uint32_t* out)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) s[c] = mix(gid * CH + c);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v[CH][NV];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int a = T.n == 1 ? 0 : (int)((s[c] >> 60) % (uint64_t)T.n);
            const uint4* p = T.t[a] + (s[c] % T.nblocks[a]) * NV;
#pragma unroll
            for (int j = 0; j < NV; ++j) v[c][j] = p[j];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) x += v[c][j].x ^ v[c][j].y ^ v[c][j].z ^ v[c][j].w;
            acc += x;
            s[c] = mix(s[c] + x);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// groups of G = NV adjacent lanes: in round r the group reads the block of its lane r (chain c), lane g taking piece g
template <int NV, int CH>
__global__ __launch_bounds__(256) void coop_kernel(Tabs T, int iters, uint32_t* out)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, g = lane & (NV - 1), base = lane & ~(NV - 1);
    uint64_t s[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) s[c] = mix(gid * CH + c);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t x[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int a = T.n == 1 ? 0 : (int)((s[c] >> 60) % (uint64_t)T.n);
            const uint64_t blk = s[c] % T.nblocks[a];
            uint32_t mine = 0;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                // owner of this round: lane base + r
                const uint32_t blo = __shfl((uint32_t)blk, base + r), bhi = __shfl((uint32_t)(blk >> 32), base + r);
                const int aa = T.n == 1 ? 0 : __shfl(a, base + r);
                const uint4 v = T.t[aa][(((uint64_t)bhi << 32) | blo) * NV + g];
                uint32_t part = v.x ^ v.y ^ v.z ^ v.w;
                // hand the pieces to the owner (sum over the group)
#pragma unroll
                for (int o = 1; o < NV; o <<= 1) part += __shfl_xor(part, o);
                if (g == r) mine = part;
            }
            x[c] = mine;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) { acc += x[c]; s[c] = mix(s[c] + x[c]); }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int NV, int CH, bool COOP>
static void run(const Tabs& T, uint64_t bytes, int blocks, int iters, uint32_t* d_out, const char* label)
{
    Tabs t = T;
    for (int a = 0; a < t.n; ++a) t.nblocks[a] = bytes / t.n / (16ull * NV);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) CK(hipEventRecord(a));
        if (COOP) hipLaunchKernelGGL((coop_kernel<NV, CH>), dim3(blocks), dim3(256), 0, 0, t, rep ? iters : iters / 4, d_out);
        else hipLaunchKernelGGL((solo_kernel<NV, CH>), dim3(blocks), dim3(256), 0, 0, t, rep ? iters : iters / 4, d_out);
        CK(hipDeviceSynchronize());
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    double reads = (double)blocks * 256.0 * iters * CH;
    printf("%-6s %-4s B=%3d footprint=%6.0f MiB in %d alloc  blocks=%5d  %8.2f ms  %7.2f Gblock/s  %8.1f GB/s  %7.2f G lane-loads/s\n", label, COOP ? "coop" : "solo",
           16 * NV, bytes / 1048576.0, t.n, blocks, ms, reads / ms / 1e6, reads * 16.0 * NV / ms / 1e6, reads * (COOP ? 1 : NV) / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    const uint64_t maxGiB = argc > 1 ? strtoull(argv[1], 0, 10) : 32ull;
    const int nAlloc = argc > 2 ? atoi(argv[2]) : 1;
    Tabs T; T.n = nAlloc;
    uint32_t* d_out; CK(hipMalloc(&d_out, 64));
    for (int a = 0; a < nAlloc; ++a) {
        uint4* p; CK(hipMalloc(&p, (maxGiB << 30) / nAlloc));
        hipLaunchKernelGGL(fill_kernel, dim3(cus * 8), dim3(256), 0, 0, p, (maxGiB << 30) / nAlloc / 16);
        T.t[a] = p; T.nblocks[a] = 0;
    }
    CK(hipDeviceSynchronize());
    const int iters = 300;
    const double sizesGiB[] = {0.5, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64};
    for (double sg : sizesGiB) {
        if (sg > (double)maxGiB) break;
        const uint64_t sz = (uint64_t)(sg * 1024.0) << 20;
        run<1, 2, false>(T, sz, cus * 8, iters, d_out, "g");
        run<2, 2, false>(T, sz, cus * 8, iters, d_out, "g");
        run<2, 2, true>(T, sz, cus * 8, iters, d_out, "g");
        run<4, 2, false>(T, sz, cus * 8, iters, d_out, "g");
        run<4, 2, true>(T, sz, cus * 8, iters, d_out, "g");
        run<8, 2, true>(T, sz, cus * 8, iters, d_out, "g");
        if (argc > 3) {   // more requests in flight: is the ~48 G/s of the rows above a limit of the memory system or of the concurrency?
            run<1, 4, false>(T, sz, cus * 8, iters, d_out, "ch4");
            run<2, 4, true>(T, sz, cus * 8, iters, d_out, "ch4");
            run<1, 8, false>(T, sz, cus * 8, iters, d_out, "ch8");
            run<2, 8, true>(T, sz, cus * 8, iters, d_out, "ch8");
            run<1, 2, false>(T, sz, cus * 16, iters, d_out, "b16");
            run<2, 2, true>(T, sz, cus * 16, iters, d_out, "b16");
        }
    }
    return 0;
}
