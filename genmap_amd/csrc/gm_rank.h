// gm_rank.h -- rank dictionary of the bidirectional FM index, laid out for MI355X.
//
// Stands where SeqAn's EPR dictionary (Levels<.., LevelsPrefixRDConfig<.., 2 levels, 1 word/block>>,
// /root/reference/src/common.hpp:38-49) stands in the reference.  Redesigned, not ported:
//   * ONE self-contained, naturally aligned block per rank query -- absolute 32-bit counts for all five
//     letters (A,C,G,T,N) followed by three bit planes of the block's symbols.  A query is a single
//     aligned 32/64/128-byte HBM read (no superblock level, no second dependent access), and one
//     read yields the ranks of EVERY letter: all sigma children of a search node (the mismatch
//     branching of src/find2_index_approx.hpp:242-299, src/algo.hpp:106-116) cost the same two
//     lines (range lo / range hi) as a single-character extension.
//   * bit-PLANAR symbols (plane j holds bit j of 32*WPP consecutive codes): per-letter in-block rank
//     is 2-3 logic ops + v_bcnt per 32 symbols, no per-letter table.
//   * the sentinel is an ordinary code (5) in the planes, so Dna5 needs no exception path and the
//     "smaller symbols" term of the bidirectional range update is width - sum(letter counts).
//
// Block = WPB 32-bit words:  [0..4] cumulative counts of A,C,G,T,N before the block,
//                            [5 + j*WPP + w] = word w of plane j (j = 0,1,2; bit t = symbol 32*w+t).
//   WPP=1 ->  32 B blocks,  32 symbols (8.0 bit/symbol)
//   WPP=3 ->  64 B blocks,  96 symbols (5.3 bit/symbol)   [2 spare words]
//   WPP=9 -> 128 B blocks, 288 symbols (3.6 bit/symbol)
// Which one is fastest is a measured property of the memory system (profiles/), not a guess.
//
// Indexes of 2^32 - 1 rows or more (the reference's 64-bit BWT variants, /root/reference/src/indexing.hpp:158-169,
// src/mappability.hpp:373-385) use the WIDE geometry, selected by WPP = 2 throughout the templates:
//   WPP=2 ->  64 B blocks,  64 symbols: [0..9] five 64-bit cumulative counts, [10 + 2j + w] word w of plane j.
// Rows, ranges and text positions are then uint64_t (BlockGeom<2>::row_t); everything else is the same code.
#pragma once
#include "gm_common.h"

namespace gm {

template <int WPP> struct BlockGeom {
    static_assert(WPP == 1 || WPP == 3 || WPP == 9, "supported block shapes");
    typedef uint32_t row_t;                                          // SA rows, range widths, text positions
    static constexpr uint32_t SPB = 32u * WPP;                       // symbols per block
    static constexpr uint32_t WPB = (WPP == 1) ? 8u : (WPP == 3) ? 16u : 32u;  // words per block
    static constexpr uint32_t HDRW = 5u;                             // words of cumulative counts
    static constexpr uint32_t BYTES = WPB * 4u;
};
template <> struct BlockGeom<2> {                                    // the wide geometry: 64-bit rows
    typedef uint64_t row_t;
    static constexpr uint32_t SPB = 64u, WPB = 16u, HDRW = 10u, BYTES = 64u;
};
constexpr int WPP_WIDE = 2;

GM_HD uint32_t popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}

// number of blocks needed for n rows: rank(n) must be answerable, so one block past n/SPB.
template <int WPP> GM_HD uint64_t num_blocks(uint64_t n) { return n / BlockGeom<WPP>::SPB + 1; }

// In-block letter counts of the first `off` symbols of block words blk[0..WPB), added to the header.
// out[c] = rank_c(position) for c = A,C,G,T,N.
template <int WPP> GM_HD void block_rank(const uint32_t* blk, uint32_t off, typename BlockGeom<WPP>::row_t out[NLET])
{
    constexpr uint32_t H = BlockGeom<WPP>::HDRW;
    uint32_t cA = 0, cC = 0, cG = 0, cT = 0, cN = 0;
#pragma unroll
    for (int w = 0; w < WPP; ++w) {
        int rem = (int)off - 32 * w;
        uint32_t m = rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
        uint32_t p0 = blk[H + w], p1 = blk[H + WPP + w], p2 = blk[H + 2 * WPP + w];
        uint32_t let = ~p2 & m;
        cA += popc32(let & ~p1 & ~p0);
        cC += popc32(let & ~p1 & p0);
        cG += popc32(let & p1 & ~p0);
        cT += popc32(let & p1 & p0);
        cN += popc32(p2 & ~p0 & m);
    }
    if (H == 5u) { out[0] = blk[0] + cA; out[1] = blk[1] + cC; out[2] = blk[2] + cG; out[3] = blk[3] + cT; out[4] = blk[4] + cN; }
    else {
        typedef typename BlockGeom<WPP>::row_t R;
        out[0] = (R)((uint64_t)blk[1] << 32 | blk[0]) + cA; out[1] = (R)((uint64_t)blk[3] << 32 | blk[2]) + cC; out[2] = (R)((uint64_t)blk[5] << 32 | blk[4]) + cG;
        out[3] = (R)((uint64_t)blk[7] << 32 | blk[6]) + cT; out[4] = (R)((uint64_t)blk[9] << 32 | blk[8]) + cN;
    }
}

// Pack the planes of one block from symbol codes (positions past n are padded with the sentinel code,
// which no letter count ever includes).  counts[] are filled by a separate prefix pass.
template <int WPP> GM_HD void pack_planes(const uint8_t* bwt, uint64_t n, uint64_t block, uint32_t* blk)
{
    constexpr uint32_t SPB = BlockGeom<WPP>::SPB;
    for (int w = 0; w < WPP; ++w) {
        uint32_t p0 = 0, p1 = 0, p2 = 0;
        for (uint32_t t = 0; t < 32; ++t) {
            uint64_t i = block * SPB + 32u * w + t;
            uint32_t c = i < n ? (uint32_t)bwt[i] : (uint32_t)SYM_SENT;
            p0 |= (c & 1u) << t;
            p1 |= ((c >> 1) & 1u) << t;
            p2 |= ((c >> 2) & 1u) << t;
        }
        blk[BlockGeom<WPP>::HDRW + w] = p0; blk[BlockGeom<WPP>::HDRW + WPP + w] = p1; blk[BlockGeom<WPP>::HDRW + 2 * WPP + w] = p2;
    }
}

}  // namespace gm
