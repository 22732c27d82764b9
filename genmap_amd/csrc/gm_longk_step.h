// gm_longk_step.h -- the per-node logic of the long k-mer kernel (gm_longk.h), as plain host/device code: the kernel runs it per lane, the
// CPU logic harness (tests/emu) runs the same functions one node at a time against the oracle.
// What it follows: _optimalSearchSchemeGM / ...ChildrenGM / ...ExactGM   /root/reference/src/find2_index_approx.hpp:223-457
//                  extend / approxSearch / extendExact                   /root/reference/src/algo.hpp:26-218
// through the restatement of gm_engine.h (make_plan, make_post, lane_children, split_node, verify_with), with unpacked 16-bit window
// coordinates and the search's record as OssRecordL (gm_oss.h).
#pragma once
#include "gm_engine.h"

namespace gm {

template <typename R> struct LNodeT { R flo, rlo, w; uint32_t ab, tem; };   // ab = a | bx << 16, tem = t | errs << 16 | mode << 24

// _optimalSearchSchemeGM(..., s.startPos, s.startPos + 1, 0, s, 0, Rev()) find2_index_approx.hpp:441
template <typename R> GM_HD LNodeT<R> long_root_node(uint32_t n, const OssRecordL& rec, R nRows)
{
    const uint32_t a0 = n - 1u + rec.start;
    LNodeT<R> nd; nd.flo = 0; nd.rlo = 0; nd.w = nRows; nd.ab = a0 | a0 << 16; nd.tem = M_OSS << 24;
    return nd;
}

// k-mer starts (block coordinates) still covered by a node (gm_kernels.h: covered_kmers)
template <typename R> GM_HD void long_covered(const LNodeT<R>& nd, uint32_t n, uint32_t K, uint32_t& smin, uint32_t& smax)
{
    const uint32_t pa = nd.ab & 0xFFFFu, pbx = nd.ab >> 16, pt = nd.tem & 0xFFFFu, pm = nd.tem >> 24;
    if (pm == M_OSS) { smin = 0u; smax = n - 1u; } else if (pm == M_EXT_R) { smin = pt - K; smax = pa; } else if (pm == M_EXT_L) { smin = pbx - K; smax = pt; } else { smin = pbx - K; smax = pa; }
}

// One node of one lane.  charAt(pos): the needle's symbol at window coordinate pos (already complemented on the reverse strand);
// push(node): the lane's LIFO.  On return `have` tells whether nd holds the node to continue with.
template <class Env, class CharAt, class Push>
GM_HD void long_node(LNodeT<typename Env::row_t>& nd, bool& have, const RootT<typename Env::row_t>& rt, const OssRecordL& recRef, uint32_t K, uint32_t E,
                     typename Env::row_t verifyT, typename Env::row_t satMinW, Env& env, CharAt charAt, Push push)
{
    typedef typename Env::row_t R;
    typedef LNodeT<R> LN;
    const OssRecordL* const rec = &recRef;
    {
        uint32_t a = nd.ab & 0xFFFFu, bx = nd.ab >> 16, t = nd.tem & 0xFFFFu, errs = (nd.tem >> 16) & 0xFFu, mode = nd.tem >> 24;
        if (nd.w <= verifyT) {
            // Narrow node: every string below it lies at the one text location of each of its rows.  On a genome a long k-mer is down to
            // a single row after its first ~20 characters; the remaining hundreds of rank steps become one suffix-array read and a scan of
            // the text there (gm_engine.h: verify_fields -- the remaining OSS blocks replayed with their bounds, then the runs of k-mers
            // that extend with at most E mismatches).
            for (R r = 0; r < nd.w; ++r) {
                const typename Env::Item it = env.item(nd.flo + r);
                verify_fields(it, a, bx, t, errs, mode, *rec, rt, K, E, env);
            }
            have = false;
            return;
        }
        if (mode == M_SPLIT) {
            // SPLIT -> EXT_R kept, EXT_L pushed: the halving targets of algo.hpp:53-56 and :68-71 (same in :196-211); gm_engine.h: split_node
            const uint32_t alm = bx - K;
            const uint32_t bxNew = bx + ((a + K - bx + 1u) >> 1);
            const uint32_t aNew = alm + ((a - alm - 1u) >> 1);
            LN left = nd;
            left.tem = aNew | errs << 16 | M_EXT_L << 24;
            bool leftDone = false, rightDone = false;
            if (nd.w >= satMinW) {   // (both halves have the parent's width)
                leftDone = env.saturated(rt, bx - K, aNew);
                rightDone = env.saturated(rt, bxNew - K, a);
            }
            if (rightDone) {
                if (leftDone) { have = false; return; }
                mode = M_EXT_L; t = aNew;
            } else {
                if (!leftDone) push(left);
                mode = M_EXT_R; t = bxNew;
            }
        }
        // ---- gm_engine.h: make_plan ----
        uint32_t right, exact, minErr = 0, charsLeft = 0;
        if (mode == M_OSS) {
            const uint32_t u = (rec->w >> (3u * t)) & 7u, l = (rec->z >> (3u * t)) & 7u;
            right = (rec->z >> (18u + t)) & 1u;
            exact = (u == errs);                                  // find2:388,397
            minErr = l > errs ? l - errs : 0u;                    // find2:389
            charsLeft = (uint32_t)rec->bl[t] - (bx - a);          // find2:247
        } else {
            right = (mode == M_EXT_R);
            exact = (errs == E);                                  // algo.hpp:106,117,143,154,175
        }
        const uint32_t pos = right ? bx : a - 1u;
        const uint32_t tc = charAt(pos);
        const R plo = right ? nd.rlo : nd.flo;
        R rl[NLET], rh[NLET];
        env.rank2(right, plo, plo + nd.w, rl, rh);
        // ---- make_post ----
        if (right) bx += 1u; else a -= 1u;
        bool done;
        if (mode == M_OSS) {
            done = false;
            if (bx - a == (uint32_t)rec->bl[t]) { t += 1u; done = (t == (uint32_t)rec->nb); }   // find2:263, :335-344, :358-367, :392-395
        } else done = right ? (bx == t) : (a == t);               // algo.hpp:101-105,138-142
        bool leaf = false;
        if (done) { leaf = (bx - a == K); mode = M_SPLIT; t = 0; }   // algo.hpp:38,180
        const uint32_t ab1 = a | bx << 16, tem0 = t | mode << 24;
        // ---- lane_children ----
        const R olo = right ? nd.flo : nd.rlo;
        R cnt[NLET], sm[NLET], pn[NLET], tot = 0;
#pragma unroll
        for (int x = 0; x < (int)NLET; ++x) { cnt[x] = rh[x] - rl[x]; tot += cnt[x]; pn[x] = env.C((uint32_t)x) + rl[x]; }
        R run = nd.w - tot;   // sentinels sort before every letter
        uint32_t nonEmpty = 0;
#pragma unroll
        for (int x = 0; x < (int)NLET; ++x) { sm[x] = run; run += cnt[x]; nonEmpty |= (cnt[x] != 0u ? 1u : 0u) << x; }
        const uint32_t matchBit = tc < SYM_N ? 1u << tc : 0u;     // a needle N mismatches everything (find2:250, algo.hpp:111-112,148-149)
        const bool okMatch = !(minErr > 0u && charsLeft < minErr + 1u);   // find2:254-258
        const bool okMiss = !exact && !(minErr > 0u && charsLeft < minErr);
        const uint32_t valid = nonEmpty & ((okMatch ? matchBit : 0u) | (okMiss ? ((1u << NLET) - 1u) & ~matchBit : 0u));
        LN keep; keep.flo = keep.rlo = keep.w = 0; keep.ab = ab1; keep.tem = 0;
        bool haveKeep = false;
        // the matching child first (deepest in the LIFO), then the mismatching ones in alphabet order; the lane continues with the last
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int x = 0; x < (int)NLET; ++x) {
                const bool isMatch = (uint32_t)x == tc && tc < SYM_N;   // (needle N against text N is a mismatch)
                if ((pass == 0) != isMatch) continue;
                if (!((valid >> x) & 1u)) continue;
                const R pnew = pn[x], onew = olo + sm[x];
                const R cf = right ? onew : pnew, cr = right ? pnew : onew;
                if (leaf) env.leaf(rt, a, cf, cnt[x]);
                else {
                    if (haveKeep) push(keep);
                    keep.flo = cf; keep.rlo = cr; keep.w = cnt[x];
                    keep.tem = tem0 | (errs + (isMatch ? 0u : 1u)) << 16;
                    haveKeep = true;
                }
            }
        }
        if (leaf) env.leaf_flush(rt, a);
        nd = keep; have = haveKeep;
    }
}

}  // namespace gm
