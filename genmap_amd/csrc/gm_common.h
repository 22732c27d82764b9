// gm_common.h -- shared constants of the MI355X-native (k,e)-mappability engine.
// Compiles as plain C++17 (host tools, the CPU logic harness under tests/emu) and as HIP (gfx950).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GM_HD __host__ __device__ __forceinline__
#else
#define GM_HD inline
#endif

namespace gm {

// symbol codes of the text and of the BWTs (N is a real 5th index symbol, src/algo.hpp:111-112,148-149;
// the sentinel is explicit in the rank planes, one after EVERY sequence, src/seqan_libdivsufsort.h:80-91)
enum : uint32_t { SYM_A = 0, SYM_C = 1, SYM_G = 2, SYM_T = 3, SYM_N = 4, SYM_SENT = 5, NLET = 5 };

constexpr uint32_t MAX_ERRORS = 4;   // "E > 4 not yet supported." src/mappability.hpp:187
constexpr uint32_t MAX_K = 255;      // 16-byte node encoding: needle-window coordinates <= 2K-1 <= 509 fit 9 bits; OSS block lengths fit 8
constexpr uint32_t MAX_K_LONG = 32768;   // longer k-mers run the plain tree walk of gm_longk.h (16-bit coordinates, blocks of at most 255 k-mers)

GM_HD uint32_t complement(uint32_t c) { return c < 4u ? 3u - c : c; }   // N stays N (src/algo.hpp:5-8)

}  // namespace gm
