// gm_engine.h -- the search engine of one lane: node encoding, step planning, child generation.
//
// What it replaces: the recursion of the reference,
//   _optimalSearchSchemeGM / ...ChildrenGM / ...ExactGM   /root/reference/src/find2_index_approx.hpp:223-457
//   extend / approxSearch / extendExact                   /root/reference/src/algo.hpp:26-218
// re-organised for a 64-wide wavefront: there is no recursion and no per-function control flow.  A search
// NODE is 16 bytes (two SA range starts, the width, 32 bits of position/phase), and EVERY node advances
// by the same step -- "extend the matched interval [a,bx) by one character in the node's direction":
// two aligned rank-block reads (range lo / range hi), from which all sigma children follow.  Whether the
// node is inside an OSS block or in the extension phase only changes a few integer fields, so lanes in
// different phases of different k-mer blocks execute one instruction stream without divergence.
//
// Phases (mode):  OSS    inside block `t` of search s of the common infix        (find2_index_approx.hpp)
//                 EXT_R  extending right until bx == t                           (algo.hpp:90-126, Rev)
//                 EXT_L  extending left  until a  == t                           (algo.hpp:127-163, Fwd)
//                 SPLIT  matched interval complete for this level; becomes EXT_R + (pushed) EXT_L with
//                        the halving targets of algo.hpp:53-56,68-71 / :196-211 -- or a LEAF when
//                        bx - a == K (algo.hpp:38-49, :180-193): width is added to the k-mer's counter.
// Coordinates are those of the reference's `needles` window (algo.hpp:256): a = first matched index,
// bx = one past the last matched index (the reference's b + 1), 0 <= a <= bx <= K + n - 1.
#pragma once
#include "gm_common.h"
#include "gm_rank.h"
#include "gm_oss.h"

namespace gm {

enum : uint32_t { M_OSS = 0, M_EXT_R = 1, M_EXT_L = 2, M_SPLIT = 3 };

// meta: a (9 bits) | bx<<9 (9) | t<<18 (9) | errs<<27 (3) | mode<<30 (2): window coordinates reach 2K-1 <= 509 (K <= MAX_K)
// R = row type: uint32_t, or uint64_t for indexes of 2^32 - 1 rows or more (gm_rank.h: BlockGeom<2>)
template <typename R> struct NodeT { R flo, rlo, w; uint32_t meta; };
typedef NodeT<uint32_t> Node;
constexpr uint32_t META_ERRS_SHIFT = 27;

GM_HD uint32_t meta_pack(uint32_t a, uint32_t bx, uint32_t t, uint32_t errs, uint32_t mode)
{
    return a | bx << 9 | t << 18 | errs << META_ERRS_SHIFT | mode << 30;
}
GM_HD uint32_t meta_a(uint32_t m) { return m & 0x1FFu; }
GM_HD uint32_t meta_bx(uint32_t m) { return (m >> 9) & 0x1FFu; }
GM_HD uint32_t meta_t(uint32_t m) { return (m >> 18) & 0x1FFu; }
GM_HD uint32_t meta_errs(uint32_t m) { return (m >> META_ERRS_SHIFT) & 7u; }
GM_HD uint32_t meta_mode(uint32_t m) { return m >> 30; }

// the root a lane is currently working on: one (k-mer block, strand, search) triple
template <typename R> struct RootT {
    R win;             // slice-relative text offset of the block's window = position of its first k-mer
    uint32_t n;        // k-mers in the block; window length W = K + n - 1; common infix = [n-1, K)
    uint32_t strand;   // 1: the window is read reverse-complemented (algo.hpp:286-287)
    uint32_t search;   // index of the OSS search inside the scheme (selects rec)
    OssRecord rec;
};
typedef RootT<uint32_t> Root;

template <typename R> GM_HD NodeT<R> root_node(const RootT<R>& rt, R nRows)
{
    // _optimalSearchSchemeGM(..., s.startPos, s.startPos + 1, 0, s, 0, Rev()) find2_index_approx.hpp:441
    uint32_t a = (rt.n - 1u) + oss_start(rt.rec);
    NodeT<R> nd; nd.flo = 0; nd.rlo = 0; nd.w = nRows; nd.meta = meta_pack(a, a, 0, 0, M_OSS);
    return nd;
}

// SPLIT -> (EXT_R kept, EXT_L returned through `left`).  algo.hpp:53-56 and :68-71 (same in :196-211).
template <typename R> GM_HD void split_node(NodeT<R>& nd, NodeT<R>& left, uint32_t K)
{
    uint32_t m = nd.meta, a = meta_a(m), bx = meta_bx(m), errs = meta_errs(m);
    uint32_t alm = bx - K;                                 // b + 1 - length  (>= 0)
    uint32_t bx_new = bx + ((a + K - bx + 1u) >> 1);       // b + ceil((a + K - 1 - b) / 2), exclusive
    uint32_t a_new = alm + ((a - alm - 1u) >> 1);          // a > alm whenever the interval is not full
    left = nd;
    left.meta = meta_pack(a, bx, a_new, errs, M_EXT_L);
    nd.meta = meta_pack(a, bx, bx_new, errs, M_EXT_R);
}

struct Plan {
    uint32_t right;      // 1: append text[bx] (reverse-index BWT, SeqAn's Rev); 0: prepend text[a-1] (Fwd)
    uint32_t exact;      // only the needle character may be taken (no error budget in this segment)
    uint32_t minErr;     // OSS lower bound still to be met inside the current block
    uint32_t charsLeft;  // OSS: characters left in the current block, including this one
    uint32_t pos;        // window coordinate of the character to compare with
};

GM_HD Plan make_plan(uint32_t meta, const OssRecord& rec, uint32_t E)
{
    Plan p;
    uint32_t a = meta_a(meta), bx = meta_bx(meta), t = meta_t(meta), errs = meta_errs(meta), mode = meta_mode(meta);
    if (mode == M_OSS) {
        uint32_t u = oss_u(rec, t), l = oss_l(rec, t);
        p.right = oss_right(rec, t);
        p.exact = (u == errs);                        // maxErrorsLeftInBlock == 0  (find2:388,397)
        p.minErr = l > errs ? l - errs : 0u;          // find2:389
        p.charsLeft = oss_bl(rec, t) - (bx - a);      // find2:247
    } else {
        p.right = (mode == M_EXT_R);
        p.exact = (errs == E);                        // errorsLeft == 0  (algo.hpp:106,117,143,154,175)
        p.minErr = 0; p.charsLeft = 0;
    }
    p.pos = p.right ? bx : a - 1u;
    return p;
}

// structural outcome of taking one character (identical for every child of the node)
struct Post { uint32_t meta0; uint32_t leaf; uint32_t kmer; };   // meta0 has errs = 0; children OR their errs in

GM_HD Post make_post(uint32_t meta, const Plan& pl, const OssRecord& rec, uint32_t K)
{
    uint32_t a = meta_a(meta), bx = meta_bx(meta), t = meta_t(meta), mode = meta_mode(meta);
    if (pl.right) bx += 1u; else a -= 1u;
    bool done;
    if (mode == M_OSS) {
        done = false;
        if (bx - a == oss_bl(rec, t)) {               // block complete (find2:263, :335-344, :358-367)
            t += 1u;
            done = (t == oss_nb(rec));                // "Done": delegate -> extend  (find2:392-395, algo.hpp:262-298)
        }
    } else {
        done = pl.right ? (bx == t) : (a == t);       // algo.hpp:101-105,138-142
    }
    Post ps; ps.leaf = 0; ps.kmer = a;
    if (done) {
        if (bx - a == K) { ps.leaf = 1; mode = M_SPLIT; }   // algo.hpp:38,180
        else mode = M_SPLIT;
        t = 0;
    }
    ps.meta0 = meta_pack(a, bx, t, 0, mode);
    return ps;
}

// One step of one lane.  Env supplies memory:
//   void rank2(uint32_t right, uint32_t lo, uint32_t hi, uint32_t rl[5], uint32_t rh[5])
//   uint32_t text_char(const Root&, uint32_t pos)     (already complemented for strand 1)
//   void push(const Node&)                            (lane-private LIFO)
//   void leaf(const Root&, uint32_t kmer, uint32_t flo, uint32_t w)   one matching string of k-mer `kmer`:
//                                                     rows [flo, flo+w) of the forward SA (countOccurrences /
//                                                     itAll.push_back, algo.hpp:42-48,185-191)
//   void leaf_flush(const Root&, uint32_t kmer)       after the last leaf() of a step
//   uint32_t C(uint32_t c)                            (first row of letter c)
//   void note_step(uint32_t mode, uint32_t width)     (statistics hook)
//   bool any(bool)                                    (true if the predicate holds in any lane of the wavefront)
// On return `have` tells whether nd holds a node to continue with.
template <class Env> GM_HD void lane_children(NodeT<typename Env::row_t>& nd, bool& have, const RootT<typename Env::row_t>& rt, uint32_t K, uint32_t E, Env& env, const Plan& pl,
                                             const typename Env::row_t rl[NLET], const typename Env::row_t rh[NLET]);

template <class Env>
GM_HD void lane_step(NodeT<typename Env::row_t>& nd, bool& have, const RootT<typename Env::row_t>& rt, uint32_t K, uint32_t E, Env& env)
{
    typedef typename Env::row_t R;
    const Plan pl = make_plan(nd.meta, rt.rec, E);
    env.note_step(meta_mode(nd.meta), nd.w);   // instrumentation hook (empty unless GM_COUNTERS)
    const R plo = pl.right ? nd.rlo : nd.flo;
    R rl[NLET], rh[NLET];
    env.rank2(pl.right, plo, plo + nd.w, rl, rh);
    lane_children(nd, have, rt, K, E, env, pl, rl, rh);
}

// The part of a step after the two rank vectors are known.  The kernel calls make_plan / rank / lane_children separately
// when the rank blocks are read cooperatively (gm_kernels.h: rank2_coop runs with every lane of the wavefront enabled).
template <class Env>
GM_HD void lane_children(NodeT<typename Env::row_t>& nd, bool& have, const RootT<typename Env::row_t>& rt, uint32_t K, uint32_t E, Env& env, const Plan& pl,
                         const typename Env::row_t rl[NLET], const typename Env::row_t rh[NLET])
{
    typedef typename Env::row_t R;
    typedef NodeT<R> Node;
    const uint32_t tc = env.text_char(rt, pl.pos);
    const Post ps = make_post(nd.meta, pl, rt.rec, K);
    const uint32_t errs = meta_errs(nd.meta);
    const R olo = pl.right ? nd.flo : nd.rlo;

    // per letter: occurrences, rows below it inside the range, first row of the child; validity as a bit mask
    R cnt[NLET], sm[NLET], pn[NLET], tot = 0;
#pragma unroll
    for (int x = 0; x < (int)NLET; ++x) { cnt[x] = rh[x] - rl[x]; tot += cnt[x]; pn[x] = env.C((uint32_t)x) + rl[x]; }
    R run = nd.w - tot;   // sentinels in BWT[lo,hi) sort before every letter
    uint32_t nonEmpty = 0;
    // Env::NLESS: the text letter N is never followed (neither here nor by the verification below): this pass then finds exactly
    // the occurrences whose text window holds no N, and the occurrences with N in the text are added by a second pass that
    // searches the N-holding text windows themselves (gm_api.hip: correction pass; Hamming distance is symmetric).  N is the
    // largest letter, so the range arithmetic of the other children does not change.
    constexpr int NX = Env::NLESS ? (int)SYM_N : (int)NLET;
#pragma unroll
    for (int x = 0; x < (int)NLET; ++x) { sm[x] = run; run += cnt[x]; if (x < NX) nonEmpty |= (cnt[x] != 0u ? 1u : 0u) << x; }
    // Which letters may be taken, as a mask (the same three conditions per letter would cost ~10 instructions each):
    //   delta(x) = x != needle letter || needle letter is N                       find2:250, algo.hpp:111-112,148-149
    //   exact segment: only delta == 0                                            find2:330, algo.hpp:117-125
    //   lower bound of the block still reachable: charsLeft + delta >= minErr + 1   find2:254-258
    const uint32_t matchBit = tc < SYM_N ? 1u << tc : 0u;
    const bool okMatch = !(pl.minErr > 0u && pl.charsLeft < pl.minErr + 1u);
    const bool okMiss = !pl.exact && !(pl.minErr > 0u && pl.charsLeft < pl.minErr);
    const uint32_t valid = nonEmpty & ((okMatch ? matchBit : 0u) | (okMiss ? ((1u << NX) - 1u) & ~matchBit : 0u));

    Node keep; keep.flo = keep.rlo = keep.w = keep.meta = 0;
    bool haveKeep = false;
    // Order of the children: the matching child first (it ends up deepest in the LIFO), then the mismatching children in
    // alphabet order; the lane continues with the last one.  Continuing with a child that has spent an error bounds the
    // stack by 4*E + log2(n) + c and lets exact-mode lanes finish before the mismatch rounds.  Only the
    // matching child needs a lane-dependent register pick; the mismatch rounds index cnt/pn/sm statically.
    const bool hasMatch = tc < SYM_N && ((valid >> tc) & 1u) != 0u;
    if (hasMatch) {
        const R cx = tc == 0 ? cnt[0] : tc == 1 ? cnt[1] : tc == 2 ? cnt[2] : cnt[3];
        const R pnew = tc == 0 ? pn[0] : tc == 1 ? pn[1] : tc == 2 ? pn[2] : pn[3];
        const R onew = olo + (tc == 0 ? sm[0] : tc == 1 ? sm[1] : tc == 2 ? sm[2] : sm[3]);
        if (ps.leaf) { env.note_wave(9); env.leaf(rt, ps.kmer, pl.right ? onew : pnew, cx); }
        else {
            keep.flo = pl.right ? onew : pnew;
            keep.rlo = pl.right ? pnew : onew;
            keep.w = cx;
            keep.meta = ps.meta0 | (errs << META_ERRS_SHIFT);
            haveKeep = true;
        }
    }
    // Env::EXACT_ONLY: the kernel was compiled for E = 0 (every segment is exact, pl.exact holds everywhere): no mismatching
    // child exists, so none of the code below -- nor the registers it keeps alive -- is generated
    const uint32_t miss = Env::EXACT_ONLY ? 0u : (hasMatch ? valid & ~(1u << tc) : valid);   // pattern N: every child is a mismatch (find2:250)
    if (!Env::EXACT_ONLY && env.any(miss != 0u)) {
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            const bool on = ((miss >> x) & 1u) != 0u;
            if (!env.any(on)) continue;
            if (on) {
                env.note_wave(8);
                const R pnew = pn[x], onew = olo + sm[x];
                if (ps.leaf) env.leaf(rt, ps.kmer, pl.right ? onew : pnew, cnt[x]);
                else {
                    if (haveKeep) env.push(keep);
                    keep.flo = pl.right ? onew : pnew;
                    keep.rlo = pl.right ? pnew : onew;
                    keep.w = cnt[x];
                    keep.meta = ps.meta0 | ((errs + 1u) << META_ERRS_SHIFT);
                    haveKeep = true;
                }
            }
        }
    }
    if (ps.leaf) { env.note_wave(10); env.leaf_flush(rt, ps.kmer); }
    nd = keep; have = haveKeep;
}

// ---- verification of narrow nodes ------------------------------------------------------------------------------
// A node whose SA range holds a single row has exactly one candidate text location for EVERY string below it in
// the search tree.  Instead of walking that subtree with rank queries (one or two random HBM lines per character),
// look the location up in the resident suffix array (one read) and compare the needle window with the text there:
//   * OSS phase: the remaining blocks of the current search are replayed with their cumulative error bounds
//     l[b] <= errors <= u[b] at every block end -- exactly the paths _optimalSearchSchemeChildrenGM / ...ExactGM
//     (find2_index_approx.hpp:223-369) would follow at this location (the lower-bound pruning :254-258 is an early
//     form of "errors >= l[b] at the block end"), so no error configuration is counted by two searches;
//   * extension phase: every k-mer the node still covers (starts in [smin, smax]) is a hit iff the location
//     extends to it without leaving the sequence and with at most E mismatches in total (algo.hpp:26-218 explores the
//     same single path per k-mer; pattern N and text N count as mismatches, a sentinel ends the occurrence).
// Env additionally supplies:
//   Env::Item item(uint32_t row)                                the candidate location of a forward SA row: item.p0 = its text
//                                                               position (sentinel text); the kernel's item also carries
//                                                               the 56 text symbols around p0 (one 32-byte record per row)
//   uint64_t needle8(const Root&, uint32_t q, bool down)        8 needle symbols, byte j = needle(q + j) or needle(q - j)
//   uint64_t text8(const Item&, int32_t off, bool down)         8 sentinel-text symbols, byte j = textS[p0 + off +- j];
//                                                               positions outside the text read as sentinels (5)
//   void leaf_at(const Root&, uint32_t kmer, uint32_t textPos)  one occurrence of k-mer `kmer` at textPos
// Symbols are compared eight at a time (one 64-bit word per side).

GM_HD uint64_t bytes_nonzero(uint64_t x)   // 0x80 in every byte of x that is not zero
{
    return (((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) & 0x8080808080808080ull;
}
GM_HD uint32_t ctz64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__ffsll((long long)x) - 1u;
#else
    return (uint32_t)__builtin_ctzll(x);
#endif
}

// Scan up to `need` characters from needle coordinate q0 (upwards or downwards) at the location anchored by (p0, a0).
// Returns how many characters can be taken: the scan stops in front of a sentinel and in front of mismatch number
// budget + 1.  cnt = mismatches among the taken characters, pos[j] = 1-based offset of mismatch j (j < 4).
template <class Env>
GM_HD uint32_t scan_side(Env& env, const RootT<typename Env::row_t>& rt, const typename Env::Item& it, uint32_t a0, uint32_t q0, bool down, uint32_t need, uint32_t budget,
                         uint32_t& cnt, uint32_t pos[4])
{
    cnt = 0;
    for (uint32_t i = 0; i < need; i += 8u) {
        const uint32_t q = down ? q0 - i : q0 + i;
        const uint64_t n8 = env.needle8(rt, q, down);
        const uint64_t t8 = env.text8(it, (int32_t)q - (int32_t)a0, down);
        env.note_chunk(); env.note_wave(12);
        uint64_t ev = bytes_nonzero(n8 ^ t8) | (0x8080808080808080ull & ~bytes_nonzero(n8 ^ 0x0404040404040404ull))   // mismatch, pattern N
                      | (0x8080808080808080ull & ~bytes_nonzero(t8 ^ 0x0505050505050505ull));                           // sentinel
        const uint32_t left = need - i;
        if (left < 8u) ev &= (1ull << (8u * left)) - 1ull;
        uint64_t sent = 0x8080808080808080ull & ~bytes_nonzero(t8 ^ 0x0505050505050505ull);
        if (Env::NLESS) {   // a text N ends the occurrence like a sentinel (those occurrences belong to the correction pass)
            sent |= 0x8080808080808080ull & ~bytes_nonzero(t8 ^ 0x0404040404040404ull);   // (always an event of `ev` too)
        }
        while (ev) {
            env.note_wave(13);
            const uint32_t bit = ctz64(ev);
            ev &= ev - 1ull;
            const uint32_t j = bit >> 3;
            if ((sent >> bit) & 1ull) return i + j;          // the occurrence ends here
            if (cnt == budget) return i + j;                 // one mismatch too many
#pragma unroll
            for (int k = 0; k < 4; ++k) if ((uint32_t)k == cnt) pos[k] = i + j + 1u;
            ++cnt;
        }
    }
    return need;
}

// ---- fast verification: short windows compared in one go ---------------------------------------------------------------------
// The scans above fetch needle and text eight symbols at a time, every fetch a dependent memory request: a verification round of 64
// items lasts as long as its longest chain of them (record, then up to five needle reads at K = 30), with the whole wavefront waiting.
// For windows of at most FV_MAXW symbols whose every needed text symbol lies inside the row's 56-symbol record (the host checks:
// K <= 32 and every window coordinate a node can be verified at is <= CTX_LEFT), all reads of an item are issued at once -- the record
// and two or three 16-byte chunks of the 4-bit text that hold the needle window -- and the comparison of the WHOLE window with the text
// around the location becomes two 64-bit masks over window coordinates: `mm` (this position is an event: mismatch, needle N, or stop)
// and `st` (the occurrence ends here: sentinel; a text N too in N-less passes).  The replay of the OSS blocks and the run logic of
// verify_with stay what they are: scan_side has an overload that reads the masks instead of memory.
constexpr uint32_t FV_MAXW = 48;
constexpr int32_t CTX_LEFT = 24, CTX_SYMS = 56;   // the record of a suffix-array row holds textS[p0 - CTX_LEFT .. p0 - CTX_LEFT + 55], 4 bits per symbol (gm_kernels.h)
template <typename R> struct MaskItemT { R p0; uint64_t mm, st; };

GM_HD uint64_t bitrev64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
#endif
}
GM_HD uint64_t nib_reverse16(uint64_t x)   // the 16 nibbles of x in reverse order
{
    x = __builtin_bswap64(x);
    return ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
}
GM_HD uint64_t nib_complement(uint64_t x)  // A<->T, C<->G, N stays N, per nibble (codes 0..4)
{
    const uint64_t n4 = x & 0x4444444444444444ull;
    return (x ^ 0x3333333333333333ull) ^ ((n4 >> 1) | (n4 >> 2));
}
GM_HD uint64_t nib_flags_to_bits(uint64_t f)   // flags at bit 0 of each of the 16 nibbles -> bits 0..15
{
    uint64_t t = (f | (f >> 3)) & 0x0303030303030303ull;
    t = (t | (t >> 6)) & 0x000F000F000F000Full;
    t = (t | (t >> 12)) & 0x000000FF000000FFull;
    return (t | (t >> 24)) & 0xFFFFull;
}
// 16 symbols starting `sh` nibbles (0..15) into the 32-nibble string hi:lo
GM_HD uint64_t nib_funnel(uint64_t lo, uint64_t hi, uint32_t sh) { return sh ? (lo >> (4u * sh)) | (hi << (64u - 4u * sh)) : lo; }

// c[0..11]: three 16-byte chunks of the 4-bit text, the first one holding the window's first symbol at nibble `woff` (0..31); W <= FV_MAXW
// symbols; strand 1: the needle is the window read backwards and complemented.  r[0..6]: the row's record; a0 <= CTX_LEFT: the window
// coordinate that the record's anchor p0 is aligned with.  NLESS: a text N ends the occurrence like a sentinel.
template <bool NLESS>
GM_HD void fv_masks(const uint32_t c[12], uint32_t woff, uint32_t W, uint32_t strand, const uint32_t r[7], uint32_t a0, uint64_t& mm, uint64_t& st)
{
    uint64_t s0 = (uint64_t)c[1] << 32 | c[0], s1 = (uint64_t)c[3] << 32 | c[2], s2 = (uint64_t)c[5] << 32 | c[4], s3 = (uint64_t)c[7] << 32 | c[6],
             s4 = (uint64_t)c[9] << 32 | c[8], s5 = (uint64_t)c[11] << 32 | c[10];
    if (woff & 16u) { s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; }
    const uint32_t wr = woff & 15u;
    uint64_t q0 = nib_funnel(s0, s1, wr), q1 = nib_funnel(s1, s2, wr), q2 = nib_funnel(s2, s3, wr);   // window symbols 0..15, 16..31, 32..47
    if (strand) {
        // needle(pos) = complement(window(W - 1 - pos)): the 48-symbol string reversed, moved down by 48 - W symbols, complemented
        uint64_t v0 = nib_reverse16(q2), v1 = nib_reverse16(q1), v2 = nib_reverse16(q0), v3 = 0ull;
        const uint32_t d = FV_MAXW - W;
        if (d & 32u) { v0 = v2; v1 = 0ull; v2 = 0ull; } else if (d & 16u) { v0 = v1; v1 = v2; v2 = 0ull; }
        const uint32_t dr = d & 15u;
        q0 = nib_complement(nib_funnel(v0, v1, dr)); q1 = nib_complement(nib_funnel(v1, v2, dr)); q2 = nib_complement(nib_funnel(v2, v3, dr));
    }
    // the text at window coordinates: window(i) <-> record symbol i + CTX_LEFT - a0; behind the record: sentinels (never looked at)
    uint64_t t0 = (uint64_t)r[1] << 32 | r[0], t1 = (uint64_t)r[3] << 32 | r[2], t2 = (uint64_t)r[5] << 32 | r[4], t3 = 0x5555555500000000ull | r[6], t4 = 0x5555555555555555ull;
    const uint32_t sh = (uint32_t)CTX_LEFT - a0;
    if (sh & 16u) { t0 = t1; t1 = t2; t2 = t3; t3 = t4; }
    const uint32_t tr = sh & 15u;
    const uint64_t x0 = nib_funnel(t0, t1, tr), x1 = nib_funnel(t1, t2, tr), x2 = nib_funnel(t2, t3, tr);
    constexpr uint64_t M1 = 0x1111111111111111ull;
    auto flags = [&](uint64_t q, uint64_t x, uint64_t& fm, uint64_t& fs) {
        const uint64_t d = q ^ x;
        fs = NLESS ? (x >> 2) & M1 : (x >> 2) & x & M1;                      // sentinel (5); N-less: a text N (4) too
        fm = ((d | (d >> 1) | (d >> 2)) & M1) | ((q >> 2) & M1) | fs;       // differs, or the needle holds N (N never matches: find2:250)
    };
    uint64_t m0, m1, m2, e0, e1, e2;
    flags(q0, x0, m0, e0); flags(q1, x1, m1, e1); flags(q2, x2, m2, e2);
    mm = nib_flags_to_bits(m0) | nib_flags_to_bits(m1) << 16 | nib_flags_to_bits(m2) << 32;
    st = (e0 | e1 | e2) ? nib_flags_to_bits(e0) | nib_flags_to_bits(e1) << 16 | nib_flags_to_bits(e2) << 32 : 0ull;   // (stops are rare)
}

// scan_side over the masks of a MaskItem: same contract, no memory.  (q0 <= 47; going down, position q0 - i is bit i of the reversed mask)
template <class Env, typename R>
GM_HD uint32_t scan_side(Env& env, const RootT<R>&, const MaskItemT<R>& it, uint32_t, uint32_t q0, bool down, uint32_t need, uint32_t budget,
                         uint32_t& cnt, uint32_t pos[4])
{
    cnt = 0;
    if (need == 0u) return 0u;
    env.note_chunk(); env.note_wave(12);
    uint64_t e = down ? bitrev64(it.mm) >> (63u - q0) : it.mm >> q0;
    uint64_t s = down ? bitrev64(it.st) >> (63u - q0) : it.st >> q0;
    if (need < 64u) e &= (1ull << need) - 1ull;
#pragma unroll
    for (uint32_t k = 0; k <= MAX_ERRORS; ++k) {
        if (!e) break;
        const uint32_t bit = ctz64(e);
        e &= e - 1ull;
        if ((s >> bit) & 1ull) return bit;          // the occurrence ends here
        if (cnt == budget) return bit;              // one mismatch too many
#pragma unroll
        for (int j = 0; j < 4; ++j) if ((uint32_t)j == cnt) pos[j] = bit + 1u;
        ++cnt;
    }
    return need;
}

// k-mers s0..s1 of the root's block all hit at the verified location (p0 is aligned with needle coordinate a0).  Leaf policies that
// only count (Env::RANGE_ADD) take the run whole: void leaf_range(const Root&, uint32_t s0, uint32_t s1); the others get one
// leaf_at per k-mer with its text position.
template <class Env>
GM_HD void emit_kmer_run(Env& env, const RootT<typename Env::row_t>& rt, uint32_t s0, uint32_t s1, typename Env::row_t p0, uint32_t a0)
{
    env.note_run();
    if constexpr (Env::RANGE_ADD) env.leaf_range(rt, s0, s1);
    else for (uint32_t s = s0; s <= s1; ++s) env.leaf_at(rt, s, p0 - (a0 - s));
}

// Self hit (gm_kernels.h: search_body).  A forward-strand node with no error spent and ONE row left is the window's own location (a
// string always matches itself) and nothing else can be found below it: the text there IS the needle, no mismatching child exists.
// So every k-mer the node still covers gains exactly one occurrence -- without a suffix-array read, a record or another rank block.
// In the OSS phase the self hit belongs to the search whose remaining lower bounds are all zero (find2:389-392; the bounds are
// cumulative, the last is the largest): the other searches drop the node.  The caller has checked that the window holds no N (such
// windows take the ordinary path, which knows which k-mers the N spoils); k-mers that cross a sequence end are zeroed by resetLimits
// whatever is added.  Returns whether [smin, smax] gain their occurrence (false: the node is dropped without a hit).
template <typename R>
GM_HD bool self_hit_kmers(uint32_t meta, const RootT<R>& rt, uint32_t K, uint32_t& smin, uint32_t& smax)
{
    const uint32_t a = meta_a(meta), bx = meta_bx(meta), t = meta_t(meta), md = meta_mode(meta);
    if (md == M_OSS) { smin = 0u; smax = rt.n - 1u; return oss_l(rt.rec, oss_nb(rt.rec) - 1u) == 0u; }
    if (md == M_EXT_R) { smin = t - K; smax = a; }
    else if (md == M_EXT_L) { smin = bx - K; smax = t; }
    else { smin = bx - K; smax = a; }
    return true;
}

// `it`: the candidate location -- Env::Item (symbols are fetched by the scans, eight at a time), or a MaskItem (below: the whole
// comparison is at hand as two 64-bit masks; scan_side has an overload for either)
// (the node's fields unpacked, the search's record as OssRecord or OssRecordL: gm_longk.h verifies with 16-bit coordinates)
template <class Env, class ItemT, class RecT>
GM_HD void verify_fields(const ItemT& it, uint32_t a, uint32_t bx, uint32_t t, uint32_t errs, uint32_t mode, const RecT& rec, const RootT<typename Env::row_t>& rt, uint32_t K, uint32_t E, Env& env)
{
    typedef typename Env::row_t R;
    const R p0 = it.p0;   // aligned with needle coordinate a0 (a changes below, keep the anchor)
    env.note_item(mode);
    const uint32_t a0 = a;
    uint32_t scratch[4];
    if (mode == M_OSS) {
        const uint32_t nb = oss_nb(rec);
        for (uint32_t bi = t; bi < nb; ++bi) {
            env.note_wave(11);
            const uint32_t right = oss_right(rec, bi), blen = oss_bl(rec, bi), u = oss_u(rec, bi), l = oss_l(rec, bi);
            const uint32_t need = blen - (bx - a);
            uint32_t c = 0;
            const uint32_t got = scan_side(env, rt, it, a0, right ? bx : a - 1u, !right, need, u - errs, c, scratch);
            if (got < need) return;            // sentinel, or more than u[b] errors (find2:388,397-401)
            errs += c;
            if (errs < l) return;              // lower bound of the block not met (find2:254-258, :389-392)
            if (right) bx += need; else a -= need;
        }
        mode = M_SPLIT;   // infix complete: [a,bx) == [n-1, K)
    }
    uint32_t smin, smax;
    if (mode == M_EXT_R) { smin = t - K; smax = a; }
    else if (mode == M_EXT_L) { smin = bx - K; smax = t; }
    else { smin = bx - K; smax = a; }
    const uint32_t budget = E - errs;                 // mismatches still allowed
    uint32_t rp[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu}, lp[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
    uint32_t rc = 0, lc = 0;
    const uint32_t rlim = scan_side(env, rt, it, a0, bx, false, smax + K - bx, budget, rc, rp);
    const uint32_t llim = scan_side(env, rt, it, a0, a - 1u, true, a - smin, budget, lc, lp);
    // only the k-mers the two scans reach: a - s <= llim and s + K - bx <= rlim (a chance hit reaches none: nothing below runs)
    if (bx + rlim < K) return;
    if (a > llim && a - llim > smin) smin = a - llim;
    if (bx + rlim - K < smax) smax = bx + rlim - K;
    if (smin > smax) return;
    // k-mer s holds L(s) = #{j : lp[j] <= a - s} mismatches on the left and R(s) = #{j : rp[j] <= s + K - bx} on the right; it is a hit
    // iff L(s) + R(s) <= budget.  L falls and R rises with s, so the hits are the union over i = 0..budget of the RUNS
    //   { s : L(s) <= i } and { s : R(s) <= budget - i }  =  [a + 1 - lp[i], rp[budget - i] + bx - K - 1]
    // (a mismatch that does not exist -- entry 0xFFFF, or index budget: the scans stop in front of mismatch budget + 1 -- bounds nothing).
    // Both ends grow as i falls: the runs are merged in one ordered pass and handed to the leaf policy whole -- a frequency call adds a
    // run with two memory operations instead of one per k-mer (K = 100: 24 k-mers per block).
    int32_t curLo = 1, curHi = 0;
#pragma unroll
    for (uint32_t step = 0; step <= MAX_ERRORS; ++step) {
        if (step > budget) break;
        env.note_wave(14);
        const uint32_t i = budget - step, r = step;   // i left mismatches allowed, r = budget - i right ones
        const uint32_t lpi = i == 0 ? lp[0] : i == 1 ? lp[1] : i == 2 ? lp[2] : i == 3 ? lp[3] : 0xFFFFu;
        const uint32_t rpr = r == 0 ? rp[0] : r == 1 ? rp[1] : r == 2 ? rp[2] : r == 3 ? rp[3] : 0xFFFFu;
        int32_t lo = (int32_t)a + 1 - (int32_t)(i < budget ? lpi : 0xFFFFu), hi = (int32_t)(r < budget ? rpr : 0xFFFFu) + (int32_t)bx - (int32_t)K - 1;
        if (lo < (int32_t)smin) lo = (int32_t)smin;
        if (hi > (int32_t)smax) hi = (int32_t)smax;
        if (lo > hi) continue;
        if (curLo <= curHi && lo <= curHi + 1) { if (hi > curHi) curHi = hi; continue; }
        if (curLo <= curHi) emit_kmer_run(env, rt, (uint32_t)curLo, (uint32_t)curHi, p0, a0);
        curLo = lo; curHi = hi;
    }
    if (curLo <= curHi) emit_kmer_run(env, rt, (uint32_t)curLo, (uint32_t)curHi, p0, a0);
}

template <class Env, class ItemT>
GM_HD void verify_with(const ItemT& it, uint32_t meta, const RootT<typename Env::row_t>& rt, uint32_t K, uint32_t E, Env& env)
{
    verify_fields(it, meta_a(meta), meta_bx(meta), meta_t(meta), meta_errs(meta), meta_mode(meta), rt.rec, rt, K, E, env);
}

template <class Env>
GM_HD void verify_item(typename Env::row_t row, uint32_t meta, const RootT<typename Env::row_t>& rt, uint32_t K, uint32_t E, Env& env)
{
    const typename Env::Item it = env.item(row);
    verify_with(it, meta, rt, K, E, env);
}

// upper bound of simultaneously stacked nodes of one lane (DESIGN.md "stack bound")
inline uint32_t stack_bound(uint32_t E, uint32_t stepSize)
{
    uint32_t lg = 0;
    while ((1u << lg) < stepSize) ++lg;
    return 4u * E + lg + 8u;
}

}  // namespace gm
