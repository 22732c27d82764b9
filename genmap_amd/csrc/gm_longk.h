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
//   * the q-mer table for the first characters of a root, but no jump patterns, no cooperative block reads, the text letter N is followed like any other (no correction pass).
// What it follows: _optimalSearchSchemeGM / ...ChildrenGM / ...ExactGM   /root/reference/src/find2_index_approx.hpp:223-457
//                  extend / approxSearch / extendExact                   /root/reference/src/algo.hpp:26-218
// through the restatement of gm_engine.h (make_plan, make_post, lane_children, split_node), with unpacked coordinates.
#pragma once
#include "gm_kernels.h"
#include "gm_longk_step.h"

namespace gm {


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
    uint32_t guard = 0, stall = 0;
    bool did = false;   // this lane handled a node in the last iteration
    auto push = [&](const LN& x) {
        if (sbase + sp < A.stackDepth) { stk[(size_t)(sbase + sp) * nth] = x; ++sp; }
        else atomicOr(A.errorFlag, 1u);   // never expected: stack_bound(E, stepSize) + STEAL_LEVELS
    };
    for (;;) {
        // a hung loop ends as an error here too (search_body: stall_cap): consecutive iterations in which no lane of the wavefront handled a node
        if (A.guardKeep == 0u) {
            stall = __ballot(did) != 0ull ? 0u : stall + 1u;
            if (stall > A.guardCap) { atomicOr(A.errorFlag, 2u); break; }
        }
        did = false;
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
                    uint32_t smin, smax;
                    long_covered(nd, rt.n, K, smin, smax);
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
                nd = long_root_node<R>(rt.n, *rec, (R)A.nRows);
                have = true;
                // the first q characters of the search's first (always exact) block from the table of all q-mers: one read instead of the q
                // widest steps (gm_kernels.h: stage 2 of search_body); an N among them ends an exact block before it starts
                const uint32_t q = ((rt.search < 4u ? A.qlenPacked[0] : A.qlenPacked[1]) >> (8u * (rt.search & 3u))) & 0xFFu;
                if (q != 0u) {
                    const uint32_t a0 = nd.ab & 0xFFFFu;
                    uint32_t idx = 0, bad = 0;
                    for (uint32_t i = 0; i < q; ++i) {
                        uint32_t c = A.text[(size_t)rt.win + (rt.strand ? W - 1u - (a0 + i) : a0 + i)];
                        bad |= c >> 2;
                        if (rt.strand) c = 3u - (c & 3u);
                        idx = idx << 2 | (c & 3u);
                    }
                    if (bad) have = false;
                    else {
                        R eFlo, eRlo, eW;
                        NodeIO<R>::load_qentry(((A.qselMask >> rt.search) & 1u) ? A.qtabB : A.qtabA, idx, eFlo, eRlo, eW);
                        if (eW == 0) have = false;
                        else { nd.flo = eFlo; nd.rlo = eRlo; nd.w = eW; nd.ab = a0 | (a0 + q) << 16; }
                    }
                }
            }
            if (!have) continue;   // (nothing left for this lane, or a root that ended at its table entry)
        }
        if (++guard > A.guardCap && A.guardKeep != 0u) { atomicOr(A.errorFlag, 2u); break; }   // (iter_cap: tests force the bound)
        did = true;
        // one node: settled against the text when it is narrow, else split / stepped (gm_longk_step.h: the same code the CPU harness runs)
        long_node(nd, have, rt, *rec, K, E, (R)A.verifyT, (R)A.satMinW, env,
                  [&](uint32_t pos) { const uint32_t c = A.text[(size_t)rt.win + (rt.strand ? W - 1u - pos : pos)]; return rt.strand ? complement(c) : c; }, push);
    }
}

}  // namespace gm
