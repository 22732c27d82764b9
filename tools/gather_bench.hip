// gather_bench.hip -- random-line-read microbenchmark for MI355X (measurement tool, not product).
// Each lane runs a dependent chain of steps; every step reads TWO independent random aligned blocks of
// B bytes (like a rank query at range lo / range hi) and derives the next addresses from the data read.
// Reports block reads per second and GB/s for B in {32,64,128}, several table sizes and occupancies:
// the achievable "random HBM line" rate is the roofline the search kernel is priced against.
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

template <int NV, int CHAINS>   // NV = uint4 per block (B = 16*NV)
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ tab, uint64_t nblocks, int iters, uint32_t* out)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s[c] = mix(gid * CHAINS + c);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v[CHAINS][NV];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            const uint4* p = tab + (s[c] % nblocks) * NV;
#pragma unroll
            for (int j = 0; j < NV; ++j) v[c][j] = p[j];
        }
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) x += v[c][j].x ^ v[c][j].y ^ v[c][j].z ^ v[c][j].w;
            acc += x;
            s[c] = mix(s[c] + x);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int NV, int CHAINS>
static void run(const uint4* tab, uint64_t bytes, int blocks, int iters, uint32_t* d_out, const char* label)
{
    uint64_t nblocks = bytes / (16ull * NV);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<NV, CHAINS>), dim3(blocks), dim3(256), 0, 0, tab, nblocks, iters / 4, d_out);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<NV, CHAINS>), dim3(blocks), dim3(256), 0, 0, tab, nblocks, iters, d_out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    double reads = (double)blocks * 256.0 * iters * CHAINS;
    printf("%-10s B=%3d table=%6.0f MiB blocks=%5d chains=%d  %8.2f ms  %7.2f Gread/s  %8.1f GB/s\n", label, 16 * NV, bytes / 1048576.0, blocks,
           CHAINS, ms, reads / ms / 1e6, reads * 16.0 * NV / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    uint64_t maxBytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 16ull) << 30;
    uint4* tab; CK(hipMalloc(&tab, maxBytes));
    uint32_t* d_out; CK(hipMalloc(&d_out, 64));
    hipLaunchKernelGGL(fill_kernel, dim3(cus * 8), dim3(256), 0, 0, tab, maxBytes / 16);
    CK(hipDeviceSynchronize());
    const int iters = 400;
    if (argc > 2) {   // sweep: where does the random-read rate fall off with the footprint (TLB reach, MALL)?
        for (uint64_t sz = 128ull << 20; sz <= maxBytes; sz <<= 1) {
            run<2, 2>(tab, sz, cus * 8, iters, d_out, "sweep");
            run<4, 2>(tab, sz, cus * 8, iters, d_out, "sweep");
            run<8, 2>(tab, sz, cus * 4, iters, d_out, "sweep");
        }
        return 0;
    }
    uint64_t sizes[] = {64ull << 20, 2ull << 30, maxBytes};
    for (uint64_t sz : sizes) {
        if (sz > maxBytes) continue;
        for (int perCU : {2, 4, 8}) {
            int blocks = cus * perCU;
            run<2, 2>(tab, sz, blocks, iters, d_out, "gather");
            run<4, 2>(tab, sz, blocks, iters, d_out, "gather");
            run<8, 2>(tab, sz, blocks, iters, d_out, "gather");
        }
        run<4, 1>(tab, sz, cus * 8, iters, d_out, "1chain");
        run<4, 4>(tab, sz, cus * 4, iters, d_out, "4chain");
        run<1, 2>(tab, sz, cus * 8, iters, d_out, "16B");
    }
    return 0;
}
