// gm_host.h -- host-side planning of one computeMappability call (no device code).
//
// Mirrors the scheduling half of /root/reference/src/algo.hpp:405-476 and the option arithmetic of
// /root/reference/src/mappability.hpp:519-543: which k-mer blocks exist, how many k-mers each holds,
// and the OSS block-length record of every block shape.  Used by the C-ABI implementation and by the
// CPU logic harness in tests/emu (which runs the same lane code without a GPU).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>
#include <algorithm>
#include "gm_oss.h"

namespace gm {

// SearchParams.overlap after mappability.hpp:519-543, i.e. the length of the common infix.
// xo < 0: option not given.  Returns 0 on error (explicit xo too large -> PARSE_ERROR in the reference).
inline uint32_t default_infix_length(uint32_t K, uint32_t E, int32_t xo)
{
    unsigned overlap;
    if (xo >= 0) overlap = (unsigned)xo;
    else if (E == 0) overlap = (unsigned)(K * 0.7);
    else {
        unsigned mm = std::min(std::max(K, 30u), 100u);
        overlap = (unsigned)((K * mm) * std::pow((double)0.7f, (double)E) / 100.0);
    }
    uint64_t maxPossibleOverlap = std::min(K - 1u, K - E - 2u);   // unsigned wrap as in the reference
    if (overlap > maxPossibleOverlap) {
        if (xo >= 0) return 0;
        overlap = (unsigned)maxPossibleOverlap;
    }
    return K - overlap;
}

// Common-infix length this build uses when the caller gives neither -xo nor an explicit infix.  The reference's rule
// (default_infix_length) was tuned for a CPU; the result of computeMappability does not depend on it (the reference's
// own tests rerun every case with other -xo values, tests/tests.sh:47-60), so the GPU picks the block shape that
// measured fastest on MI355X (profiles/r01e_infix_sweeps.txt): with a q-mer table and verification of narrow nodes,
// e = 0 wants a 16-character infix (one lookup + ~4 steps per half) and blocks of at most 31 k-mers; e >= 1 wants
// the infix long enough that the OSS search ends on narrow ranges.
// locating: the call locates its occurrences (--exclude-pseudo, csv).  Those kernels walk the tree from its root (no jump patterns), nothing
// saturates (every occurrence is wanted), and on a multi-genome index -- what these options are for -- nearly every k-mer has a handful
// of occurrences: sharing the infix search between the k-mers of a block does not pay for the extension phase it adds.  C5's index
// (five bacteria, 21 Mbp), -ep pass in ms for n = 1 / 3-4 / round-2 shape: K=24 e=1 43 / 74 / 109 (n=5; 165 with the n=8 of the jumping
// kernels), K=30 e=1 42 / 69 / 97, K=30 e=2 131 / 145 / 227, K=50 e=1 52 / 72 / 186, K=100 e=1 91 / 68 / 190 (profiles/r03/c5_block_shape.txt)
inline uint32_t tuned_infix_length(uint32_t K, uint32_t E, bool locating = false)
{
    auto clampu = [](uint32_t v, uint32_t lo, uint32_t hi) { return v < lo ? lo : (v > hi ? hi : v); };
    uint32_t n;
    if (K > MAX_K) {
        // gm_longk.h walks the plain tree: the infix (K - n + 1 characters) is searched once per block, the extension costs ~n log2 n steps
        // per surviving path -- long blocks pay at any E and for every leaf policy
        n = clampu(K / 4, 48, 255);
        return std::max(K - n + 1, std::min(K, std::max<uint32_t>(E + 2, oss_scheme(E > MAX_ERRORS ? 0 : E).s[0].nb)));
    }
    if (locating) {   // (e = 0 alike: K=24 28 ms against 116 with blocks of 8, K=30 25 / 120, K=100 13.5 with n = 4 against 14.1 with 31 and 29 with 1)
        n = K < 64 ? 1 : 4;
        const uint32_t minInfixL = std::min(K, std::max<uint32_t>(E + 2, oss_scheme(E > MAX_ERRORS ? 0 : E).s[0].nb));
        return std::max(K - std::min(n, K) + 1, minInfixL);
    }
    switch (E) {
        // the exact infix is one table read plus one step: 17 characters for the 4^16 table of indexes beyond 2^30 rows
        // (profiles/r03/sweep_qtable16.txt: K=30 n=14 54.1 ms, n=13 54.7; with the 4^15 table n=14 61.5, n=15 62.6)
        case 0: n = K > 16 ? clampu(K - 16, 1, 31) : 1; break;
        case 1:
            // re-measured on the 3.09 Gbp index with cooperative reads and verification records (profiles/r02/sweep_grch38_steps.txt):
            // the longest block whose first (exact) part still has >= 17 characters (infix >= 35), and whose window K + n - 1
            // stays within 127 symbols (five 32-symbol LDS chunks) when that is in reach.  K=50 +9 %, 64 +13 %, 100 +11 %,
            // 150 +27 %, 250 +48 % over n = clamp(K / 6, 5, 16); K <= 43 keeps that rule (K=30: n = 5 is the optimum).
            if (K >= 128) n = 48;
            else if (K >= 60) { n = clampu(K / 4, 16, 24); if (K + n - 1 > 127 && K <= 112) n = 128 - K; }
            else if (K >= 44) n = std::min<uint32_t>(16, K - 34);
            // K <= 43, re-measured in round 3 with jump patterns (profiles/r03/sweep_shapes.txt): the jump makes the top of the
            // tree cheap, so longer blocks pay: K=30 n=8 +10 % over n=5 (flat up to 12), K=24 n=8 +13 %
            else n = clampu(K / 3, 5, 8);
            break;
        case 2: n = clampu(K / 6, 6, 16); break;   // K=30: n=6 +4 % over 7 with jump patterns and blocks of 5,4,7,8 (r03)
        case 3: n = clampu(K / 4, 9, 16); break;
        default: n = clampu(K / 4, 11, 16); break;
    }
    if (n > K) n = K;
    uint32_t infix = K - n + 1;
    const uint32_t minInfix = std::min(K, std::max<uint32_t>(E + 2, oss_scheme(E > MAX_ERRORS ? 0 : E).s[0].nb));
    return std::max(infix, minInfix);
}

// K > MAX_K: the common infix is kept long enough that a block holds at most 255 k-mers (the result does not depend on it)
inline uint32_t long_k_infix(uint32_t K, uint32_t infix) { return (K > MAX_K && infix >= 1 && infix <= K && K - infix + 1 > 255u) ? K - 254u : infix; }

struct MapPlan {
    uint32_t K = 0, E = 0, infix = 0, stepSize = 0, nSearches = 0, nStrands = 0;
    uint64_t textLen = 0, numKmers = 0;
    bool useList = false;
    // when useList: (low 32 bits of the first k-mer position, k-mers | high bits of the position << 8) -- block_pos / block_n
    std::vector<std::pair<uint32_t, uint32_t>> blocks;
    static uint64_t block_pos(const std::pair<uint32_t, uint32_t>& b) { return (uint64_t)(b.second >> 8) << 32 | b.first; }
    static uint32_t block_n(const std::pair<uint32_t, uint32_t>& b) { return b.second & 0xFFu; }
    uint64_t numBlocks = 0;
    std::vector<OssRecord> table;                        // [(n-1)*8 + s], n = 1..stepSize
    std::vector<OssRecordL> tableL;                      // the same for K > MAX_K (gm_longk.h); `table` is then all zeros
    uint64_t numRoots() const { return numBlocks * nSearches * nStrands; }
};

enum PlanError { PLAN_OK = 0, PLAN_BAD_E = -2, PLAN_BAD_K = -6, PLAN_BAD_OVERLAP = -5, PLAN_TOO_LONG = -7 };

// intervals: half-open (begin,end) pairs in slice coordinates (mappability.hpp:334-357); overlapping
// intervals are merged (the reference skips already-filled positions, algo.hpp:236-242; the value of a
// position does not depend on which block computes it).
inline int make_map_plan(uint32_t K, uint32_t E, uint32_t infix, int revcompl, uint64_t textLen,
                         const uint64_t* intervals, uint64_t nIntervals, MapPlan* out, int partBias = 0, uint32_t ossWeights = 0)
{
    MapPlan& p = *out;
    if (E > MAX_ERRORS) return PLAN_BAD_E;
    if (K < 1 || K > MAX_K_LONG) return PLAN_BAD_K;
    if (infix < 1 || infix > K) return PLAN_BAD_OVERLAP;
    const bool longK = K > MAX_K;
    if (longK && K - infix + 1 > 255u) return PLAN_BAD_OVERLAP;   // (callers clamp: long_k_infix; block lists hold the k-mers of a block in 8 bits)
    if (textLen >= (1ull << 40)) return PLAN_TOO_LONG;
    p.K = K; p.E = E; p.infix = infix; p.stepSize = K - infix + 1;   // algo.hpp:416
    p.nSearches = oss_scheme(E).ns; p.nStrands = revcompl ? 2 : 1;
    p.textLen = textLen;
    p.numKmers = textLen >= K ? textLen - K + 1 : 0;                 // algo.hpp:414 underflows for textLen < K
    p.table.assign((size_t)p.stepSize * 8, OssRecord{0, 0, 0, 0});
    p.tableL.clear();
    if (longK) p.tableL.assign((size_t)p.stepSize * 8, OssRecordL{{0, 0, 0, 0, 0, 0}, 0, 0, 0, 0});
    for (uint32_t n = 1; n <= p.stepSize; ++n)
        for (uint32_t s = 0; s < p.nSearches; ++s)
        {
            // partBias (e = 1, two blocks): characters moved from the second block to the first; the scheme is exact for any
            // positive lengths (gm_oss.h), the reference splits evenly
            uint32_t lens[OSS_MAXB]; const uint32_t* lp = nullptr;
            const uint32_t L = K - n + 1;
            if (E == 1 && partBias != 0 && L >= 2) {
                const int a = std::max(1, std::min((int)L - 1, (int)(L / 2 + (L & 1u)) + partBias));
                lens[0] = (uint32_t)a; lens[1] = L - (uint32_t)a; lp = lens;
            }
            // ossWeights (any E >= 1): block i (left to right) gets a share of the infix proportional to nibble i of the
            // value -- e.g. 0x7755 at e = 2 makes the two right blocks, with which searches 2 and 3 start, longer
            const uint32_t nb = oss_scheme(E).s[0].nb;
            if (E >= 1 && ossWeights != 0 && L >= nb) {
                uint32_t W = 0, sum = 0;
                for (uint32_t i = 0; i < nb; ++i) W += std::max(1u, (ossWeights >> (4u * i)) & 15u);
                for (uint32_t i = 0; i < nb; ++i) { lens[i] = std::max(1u, L * std::max(1u, (ossWeights >> (4u * i)) & 15u) / W); sum += lens[i]; }
                for (uint32_t i = 0; sum < L; i = (i + 1) % nb) { lens[i]++; sum++; }           // remainder: left to right
                for (uint32_t i = nb; sum > L; ) { i = i ? i - 1 : nb - 1; if (lens[i] > 1) { lens[i]--; sum--; } }   // (minimum lengths pushed it over)
                lp = lens;
            }
            if (longK) { if (!oss_make_record_long(E, s, L, &p.tableL[(size_t)(n - 1) * 8 + s], lp)) return PLAN_BAD_OVERLAP; }
            else if (!oss_make_record(E, s, L, &p.table[(size_t)(n - 1) * 8 + s], lp)) return PLAN_BAD_OVERLAP;
        }
    p.blocks.clear();
    if (nIntervals == 0) {
        p.useList = false;
        p.numBlocks = (p.numKmers + p.stepSize - 1) / p.stepSize;   // algo.hpp:435
    } else {
        p.useList = true;
        std::vector<std::pair<uint64_t, uint64_t>> iv;
        for (uint64_t k = 0; k < nIntervals; ++k) {
            uint64_t b = intervals[2 * k], e = std::min(intervals[2 * k + 1], p.numKmers);   // algo.hpp:231-233
            if (b < e) iv.emplace_back(b, e);
        }
        std::sort(iv.begin(), iv.end());
        std::vector<std::pair<uint64_t, uint64_t>> mg;
        for (auto& x : iv) {
            if (!mg.empty() && x.first <= mg.back().second) mg.back().second = std::max(mg.back().second, x.second);
            else mg.push_back(x);
        }
        for (auto& x : mg)
            for (uint64_t i = x.first; i < x.second; i += p.stepSize)                          // algo.hpp:448-451
                p.blocks.emplace_back((uint32_t)i, (uint32_t)std::min<uint64_t>(p.stepSize, x.second - i) | (uint32_t)(i >> 32) << 8);
        p.numBlocks = p.blocks.size();
    }
    return PLAN_OK;
}

// ---- correction pass of N-less searches (gm_engine.h: Env::NLESS) ----------------------------------------------------------------
// Starts t of the text windows [t, t + K) that overlap a run of N by 1..E letters (or hold a whole run of at most E letters) and
// lie inside one sequence: exactly the windows that can be an occurrence with N in the text (a window holding more than E letters
// of one run cannot; windows also touching other runs are kept -- they cost a search, not correctness).  No window without N is
// ever listed.  runs: [begin, end) of maximal N runs of the concatenated text, sorted; cum: sequence limits.  out: [begin, end) pairs.
inline void n_window_intervals(const std::vector<std::pair<uint64_t, uint64_t>>& runs, const std::vector<uint64_t>& cum, uint32_t K, uint32_t E,
                               std::vector<uint64_t>& out)
{
    out.clear();
    if (E == 0 || K == 0) return;
    std::vector<std::pair<int64_t, int64_t>> iv;   // inclusive
    for (auto& r : runs) {
        const int64_t s = (int64_t)r.first, e = (int64_t)r.second, k = K, x = E;
        iv.emplace_back(s - k + 1, std::min(s - k + x, e - 1));   // the window's tail overlaps the head of the run (and it does touch it)
        iv.emplace_back(std::max(e - x, s - k + 1), e - 1);       // the window's head overlaps the tail of the run
        if (std::min(e - s, k) <= x) iv.emplace_back(s - k + 1, e - 1);   // no window can hold more than E letters of this run (short run, or K <= E)
    }
    std::sort(iv.begin(), iv.end());
    size_t sq = 0;
    int64_t curB = 0, curE = -1;
    auto emit = [&](int64_t b, int64_t e2) {   // clip [b, e2] to whole windows of single sequences
        while (b <= e2) {
            while (sq + 1 < cum.size() && (int64_t)cum[sq + 1] <= b) ++sq;
            if (sq + 1 >= cum.size()) return;
            const int64_t lastStart = (int64_t)cum[sq + 1] - (int64_t)K;   // last window start inside sequence sq
            const int64_t hi = std::min(e2, lastStart);
            if (hi >= b) { out.push_back((uint64_t)b); out.push_back((uint64_t)hi + 1); }
            b = std::max(b, (int64_t)cum[sq + 1]);
            if (hi == e2) return;
        }
    };
    for (auto& v : iv) {
        int64_t b = std::max<int64_t>(v.first, 0), e2 = v.second;
        if (e2 < b) continue;
        if (curE >= curB && b <= curE + 1) { curE = std::max(curE, e2); continue; }
        if (curE >= curB) emit(curB, curE);
        curB = b; curE = e2;
    }
    if (curE >= curB) emit(curB, curE);
}

}  // namespace gm
