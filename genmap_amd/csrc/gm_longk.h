// gm_longk.h -- k-mers longer than MAX_K = 255 (the reference takes any -K: /root/reference/src/mappability.hpp:425-426).
//
// The persistent kernel of gm_kernels.h packs a search node into 16 bytes with 9-bit window coordinates, keeps every lane's needle
// window in LDS and the OSS record of its root in four registers with 8-bit block lengths: all of that is sized for the k-mer lengths
// mappability is computed for in practice (K = 24 .. 250).  Longer k-mers take this kernel instead: the same search -- the same
// children in the same order, the same pruning, the same leaf policies (gm_kernels.h: CountEnv, FileSetEnv, OccCountEnv, OccEmitEnv) --
// as a plain depth-first walk, one root per lane at a time:
//   * a node is {fwd lo, rev lo, width, a | bx << 16, t | errs << 16 | mode << 24}: 16-bit window coordinates;
//   * the needle is read from the one-byte-per-symbol text (a k-mer block's window is contiguous: cache lines shared by its steps);
//   * the search's OssRecordL (gm_oss.h: 16-bit block lengths) is read from its table when needed;
//   * pending nodes live on a lane-private stack in HBM (level-major: a level of the 64 lanes is contiguous), bounded by stack_bound();
//   * narrow nodes (at most verifyT rows; needs the resident suffix array) are settled against the text by verify_fields (gm_engine.h):
//     on a genome a long k-mer has one row left after ~20 characters, the rest of the walk is one suffix-array read and a text scan;
//   * no q-mer tables, no jump patterns, no cooperative block reads, the text letter N is followed like any other (no correction pass).
// What it follows: _optimalSearchSchemeGM / ...ChildrenGM / ...ExactGM   /root/reference/src/find2_index_approx.hpp:223-457
//                  extend / approxSearch / extendExact                   /root/reference/src/algo.hpp:26-218
// through the restatement of gm_engine.h (make_plan, make_post, lane_children, split_node), with unpacked coordinates.
#pragma once
#include "gm_kernels.h"

namespace gm {

template <typename R> struct LNodeT { R flo, rlo, w; uint32_t ab, tem; };   // ab = a | bx << 16, tem = t | errs << 16 | mode << 24

template <int WPP, class EnvT>
__global__ __launch_bounds__(256) void longk_kernel(const SearchArgs A)
{
    typedef typename BlockGeom<WPP>::row_t R;
    typedef LNodeT<R> LN;
    typedef RootT<R> Root;
    extern __shared__ uint4 smem[];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    EnvT env(A, nullptr, A.K);
    if constexpr (EnvT::LEAFQ) {   // the locating policies queue their leaves per wavefront; here the queue has no room (lqCap = 0): every lane walks its own leaves
        env.lq = smem;
        env.lqCtl = reinterpret_cast<uint32_t*>(smem) + wv * 80u;
        if (lane == 0u) env.lqCtl[0] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    LN* const stk = reinterpret_cast<LN*>(A.stack) + tid;   // level i of this lane: stk[i * nth]
    LN* const stkW = reinterpret_cast<LN*>(A.stack) + (tid - lane);   // ... of lane l of this wavefront: stkW[i * nth + l]
    uint8_t* const pairing = reinterpret_cast<uint8_t*>(smem) + 4u * 80u * 4u + wv * 64u;   // work sharing: victim of thief number i
    const OssRecordL* const tableL = reinterpret_cast<const OssRecordL*>(A.tableL);
    const OssRecordL* rec = tableL;
    const uint32_t K = A.K, E = A.E;
    uint32_t sp = 0, sbase = 0;   // the lane's stack entries live at levels [sbase, sbase + sp): sbase rises when a neighbour takes the bottom entry
    bool have = false, exhausted = false;
    LN nd; nd.flo = nd.rlo = nd.w = 0; nd.ab = nd.tem = 0;
    Root rt; rt.win = 0; rt.n = 1; rt.strand = 0; rt.search = 0; rt.rec = OssRecord{0, 0, 0, 0};
    uint32_t W = K;
    uint32_t guard = 0;
    auto push = [&](const LN& x) {
        if (sbase + sp < A.stackDepth) { stk[(size_t)(sbase + sp) * nth] = x; ++sp; }
        else atomicOr(A.errorFlag, 1u);   // never expected: stack_bound(E, stepSize) + STEAL_LEVELS
    };
    for (;;) {
        // ---- work sharing inside the wavefront (the scheme of gm_kernels.h: search_body) once the roots have run out: a lane with nothing
        // left takes the BOTTOM entry (the oldest, i.e. largest pending subtree) of a lane that holds a node and a stack.  One lane walks
        // its root's whole subtree with a dependent memory round trip per step: a root inside a repeat family (thousands of near-identical
        // copies: ~1e5 nodes at e = 1) would otherwise hold the kernel for tenths of a second after every other lane has finished.
        if (A.steal) {
            const bool idle = !have && sp == 0u && (exhausted || A.steal >= 2u);   // (steal >= 2: before a lane draws its next root, too)
            const bool rich = have && sp >= 1u && sbase < STEAL_LEVELS;
            const unsigned long long im = __ballot(idle), vm = __ballot(rich);
            if (im != 0ull && vm != 0ull) {
                const uint32_t np = min((uint32_t)__popcll(im), (uint32_t)__popcll(vm));
                const uint32_t ri = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
                const uint32_t rv = __builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0u));
                const bool robbed = rich && rv < np, thief = idle && ri < np;
                if (robbed) pairing[rv] = (uint8_t)lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint32_t src = lane;
                if (thief) src = pairing[ri];
                const int a4 = (int)(src << 2);
                const uint32_t vsb = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)sbase);
                const uint32_t vwin = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)(uint32_t)rt.win);
                const uint32_t vwinHi = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)(uint32_t)((uint64_t)rt.win >> 32));
                const uint32_t vnss = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)(rt.n | rt.strand << 9 | rt.search << 10));
                const uint32_t vrh = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)env.root_hits());
                if (thief) {
                    nd = stkW[(size_t)vsb * nth + src];
                    have = true;
                    rt.win = (R)((uint64_t)vwinHi << 32 | vwin); rt.n = vnss & 0x1FFu; rt.strand = (vnss >> 9) & 1u; rt.search = vnss >> 10;
                    rec = tableL + ((size_t)(rt.n - 1u) * 8u + rt.search);
                    W = K + rt.n - 1u;
                    env.set_root_hits(vrh);   // a root that saturates its k-mers keeps doing so in the thief's hands
                }
                if (robbed) { sp -= 1u; sbase = sp ? sbase + 1u : 0u; }
            }
        }
        if (__ballot(have || sp != 0u || !exhausted) == 0ull) break;   // the whole wavefront has nothing left (lanes leave together: the exchange above needs all of them)
        if (!have) {
            if (sp > 0u) {
                --sp; nd = stk[(size_t)(sbase + sp) * nth]; have = true;
                if (sp == 0u) sbase = 0u;
                if (nd.w >= (R)A.satMinW) {   // pending work for k-mers that already reached MAX is dropped (gm_kernels.h: search_body, covered_kmers)
                    const uint32_t pa = nd.ab & 0xFFFFu, pbx = nd.ab >> 16, pt = nd.tem & 0xFFFFu, pm = nd.tem >> 24;
                    uint32_t smin, smax;
                    if (pm == M_OSS) { smin = 0u; smax = rt.n - 1u; } else if (pm == M_EXT_R) { smin = pt - K; smax = pa; } else if (pm == M_EXT_L) { smin = pbx - K; smax = pt; } else { smin = pbx - K; smax = pa; }
                    if (env.saturated(rt, smin, smax)) { have = false; continue; }
                }
            }
            else if (!exhausted) {
                const unsigned long long r = atomicAdd(A.workCounter, 1ull);
                if (r >= A.numRoots) { exhausted = true; continue; }
                // root r -> (k-mer block, strand, search): the arithmetic of gm_stage1.inc
                unsigned long long gb = r / A.rootsPerBlock;
                const uint32_t rr = (uint32_t)(r - gb * A.rootsPerBlock);
                if (A.chunkBlocks) {
                    const unsigned long long q = gb / A.chunkBlocks;
                    gb = (q * A.chunkStride + A.chunkIndex) * A.chunkBlocks + (gb - q * A.chunkBlocks);
                }
                gb += A.blockBegin;
                if (A.blockList) { const uint2 e = A.blockList[gb]; rt.win = (R)((uint64_t)(e.y >> 8) << 32 | e.x); rt.n = e.y & 0xFFu; }
                else {
                    rt.win = (R)(gb * A.stepSize);
                    const uint64_t left = A.numKmers - (uint64_t)rt.win;
                    rt.n = left < A.stepSize ? (uint32_t)left : A.stepSize;
                }
                rt.strand = rr >= A.nSearches ? 1u : 0u;
                rt.search = rr - rt.strand * A.nSearches;
                rec = tableL + ((size_t)(rt.n - 1u) * 8u + rt.search);
                W = K + rt.n - 1u;
                env.on_root();
                // _optimalSearchSchemeGM(..., s.startPos, s.startPos + 1, 0, s, 0, Rev()) find2_index_approx.hpp:441
                const uint32_t a0 = rt.n - 1u + rec->start;
                nd.flo = 0; nd.rlo = 0; nd.w = (R)A.nRows; nd.ab = a0 | a0 << 16; nd.tem = M_OSS << 24;
                have = true;
            }
            if (!have) continue;   // (nothing left for this lane: it waits for its wavefront, or for a share of a neighbour's work)
        }
        if (++guard > A.guardCap && A.guardKeep != 0u) { atomicOr(A.errorFlag, 2u); break; }   // (iter_cap: tests force the bound)
        uint32_t a = nd.ab & 0xFFFFu, bx = nd.ab >> 16, t = nd.tem & 0xFFFFu, errs = (nd.tem >> 16) & 0xFFu, mode = nd.tem >> 24;
        if (nd.w <= (R)A.verifyT) {
            // Narrow node: every string below it lies at the one text location of each of its rows.  On a genome a long k-mer is down to
            // a single row after its first ~20 characters; the remaining hundreds of rank steps become one suffix-array read and a scan of
            // the text there (gm_engine.h: verify_fields -- the remaining OSS blocks replayed with their bounds, then the runs of k-mers
            // that extend with at most E mismatches).
            for (R r = 0; r < nd.w; ++r) {
                const typename EnvT::Item it = env.item(nd.flo + r);
                verify_fields(it, a, bx, t, errs, mode, *rec, rt, K, E, env);
            }
            have = false;
            continue;
        }
        if (mode == M_SPLIT) {
            // SPLIT -> EXT_R kept, EXT_L pushed: the halving targets of algo.hpp:53-56 and :68-71 (same in :196-211); gm_engine.h: split_node
            const uint32_t alm = bx - K;
            const uint32_t bxNew = bx + ((a + K - bx + 1u) >> 1);
            const uint32_t aNew = alm + ((a - alm - 1u) >> 1);
            LN left = nd;
            left.tem = aNew | errs << 16 | M_EXT_L << 24;
            bool leftDone = false, rightDone = false;
            if (nd.w >= (R)A.satMinW) {   // (both halves have the parent's width)
                leftDone = env.saturated(rt, bx - K, aNew);
                rightDone = env.saturated(rt, bxNew - K, a);
            }
            if (rightDone) {
                if (leftDone) { have = false; continue; }
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
        uint32_t tc = A.text[(size_t)rt.win + (rt.strand ? W - 1u - pos : pos)];
        if (rt.strand) tc = complement(tc);
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
