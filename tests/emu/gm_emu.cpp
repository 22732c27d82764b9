// gm_emu.cpp -- CPU LOGIC HARNESS for the device engine (test infrastructure only).
//
// Runs the very same lane code the HIP kernel runs (genmap_amd/csrc/gm_engine.h, gm_rank.h, gm_oss.h,
// gm_host.h) one lane at a time on the host, so that the search logic can be checked against the oracle
// in a container without a GPU.  It is NOT a product path: libgenmap_amd.so never contains or calls it,
// and the GPU parity tests (-m gpu) go through the C ABI only.
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../genmap_amd/csrc/gm_engine.h"
#include "../../genmap_amd/csrc/gm_host.h"
#include "../../genmap_amd/csrc/gm_longk_step.h"
#include "../../genmap_amd/csrc/gm_expand.h"

using namespace gm;

static uint32_t g_ossWeights = 0;   // relative OSS block lengths for the plans below (gm_host.h: make_map_plan), 0 = even split
extern "C" void gm_emu_set_oss_weights(uint32_t w) { g_ossWeights = w; }
static int g_jumpGroups = 0;        // 1: patterns that differ in their last three characters only are read through one word of an existence bitmap (gm_oss.h)
extern "C" void gm_emu_set_jump_groups(int on) { g_jumpGroups = on; }   // 2: groups in every layout (any three adjacent characters), 1: LOW / MID only
static int g_selfHit = 1;           // self hits of the counting pass (gm_engine.h: self_hit_kmers)
extern "C" void gm_emu_set_self_hit(int on) { g_selfHit = on; }
static int g_stateMachine = 1;      // roots with jump patterns go through the device's pattern-fetch state machine (gm_oss.h: jump_decide); 0: a plain loop over the items
extern "C" void gm_emu_set_state_machine(int on) { g_stateMachine = on; }
static uint64_t g_hangs = 0;        // roots whose state machine did not finish within its iteration bound (+ 10^6 per wrong word address)
extern "C" uint64_t gm_emu_hangs(int reset) { const uint64_t v = g_hangs; if (reset) g_hangs = 0; return v; }
static int g_fastVerify = 1;        // narrow nodes settled from the masks of gm_engine.h: fv_masks where the device would (K <= 32, short windows)
extern "C" void gm_emu_set_fast_verify(int on) { g_fastVerify = on; }
static uint64_t g_fastItems = 0;    // items verified that way since the last reset (tests make sure the path is exercised)
extern "C" uint64_t gm_emu_fast_items(int reset) { const uint64_t v = g_fastItems; if (reset) g_fastItems = 0; return v; }

static int g_expand = 0;            // 1: roots with jump patterns go through phase A of the split search (gm_expand.h: one work item per (root, item), node
extern "C" void gm_emu_set_expand(int on) { g_expand = on; }   // packets in three lists) and the packets are walked list by list, from their own windows
static int g_nbFilter = 1;          // the neighbour filters of one- and two-row table entries (needs the suffix array); 2: one-row entries only, 0: off
extern "C" void gm_emu_set_nb_filter(int v) { g_nbFilter = v; }
static uint64_t g_packets[4] = {0, 0, 0, 0};   // packets per list since the last reset, [3]: nodes ended by the neighbour filters
extern "C" void gm_emu_packets(uint64_t* out, int reset) { for (int i = 0; i < 4; ++i) { out[i] = g_packets[i]; if (reset) g_packets[i] = 0; } }

template <int WPP> struct HostIndex {
    std::vector<uint32_t> blk[2];
    uint32_t C[NLET + 1];
    uint64_t n;
    void build(const uint8_t* bf, const uint8_t* br, uint64_t rows, uint32_t nseq)
    {
        n = rows;
        const uint8_t* b[2] = {bf, br};
        constexpr uint32_t SPB = BlockGeom<WPP>::SPB, WPB = BlockGeom<WPP>::WPB;
        for (int d = 0; d < 2; ++d) {
            uint64_t nb = num_blocks<WPP>(rows);
            blk[d].assign(nb * WPB, 0);
            uint32_t run[NLET] = {0, 0, 0, 0, 0};
            for (uint64_t q = 0; q < nb; ++q) {
                uint32_t* p = &blk[d][q * WPB];
                for (int c = 0; c < (int)NLET; ++c) p[c] = run[c];
                pack_planes<WPP>(b[d], rows, q, p);
                for (uint32_t t = 0; t < SPB; ++t) { uint64_t i = q * SPB + t; if (i < rows && b[d][i] < NLET) run[b[d][i]]++; }
            }
            if (d == 0) { uint32_t acc = nseq; for (int c = 0; c < (int)NLET; ++c) { C[c] = acc; acc += run[c]; } C[NLET] = acc; }
        }
    }
};

template <int WPP, bool NL = false, bool RA = false> struct EmuEnv {
    typedef uint32_t row_t;
    static constexpr bool EXACT_ONLY = false;
    static constexpr bool NLESS = NL;
    // the main pass takes verified runs of k-mers whole, like the device's CountEnv: +1 / -1 in a difference plane that the end of run()
    // sums inside each block (regular partition), or k-mer by k-mer into acc (selections)
    static constexpr bool RANGE_ADD = RA;
    std::vector<uint32_t>* diff = nullptr;
    uint64_t selfHits = 0;
    void leaf_range(const Root& rt, uint32_t s0, uint32_t s1)
    {
        const uint32_t lo = rt.win + (rt.strand ? rt.n - 1 - s1 : s0), hi = lo + (s1 - s0);
        if (diff) { (*diff)[lo] += 1u; if (hi + 1u < rt.win + rt.n) (*diff)[hi + 1u] -= 1u; }
        else for (uint32_t p = lo; p <= hi; ++p) { uint32_t& v = (*acc)[p]; if (v != 0xFFFFFFFFu) ++v; }
    }
    // correction pass (scatter): a located occurrence adds one to ITS OWN slice position (the needle is a text window with N)
    bool scatter = false;
    const uint64_t* cumAll = nullptr; uint32_t nSeqAll = 0; uint64_t sliceBegin = 0, sliceLen = 0;
    void scatter_at(uint32_t sentPos)
    {
        uint32_t lo = 0, hi = nSeqAll;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cumAll[mid] + mid <= sentPos) lo = mid; else hi = mid; }
        const uint64_t g = (uint64_t)sentPos - lo;             // sentinel-free position
        if (g < sliceBegin || g - sliceBegin >= sliceLen) return;
        const uint64_t q = g - sliceBegin;
        if (selBlocks) {   // a selection: only the positions the main pass computes (blocks sorted by position)
            size_t a = 0, b = selBlocks->size();
            while (b - a > 1) { const size_t m = (a + b) >> 1; if (MapPlan::block_pos((*selBlocks)[m]) <= q) a = m; else b = m; }
            if (selBlocks->empty() || MapPlan::block_pos((*selBlocks)[a]) > q || q >= MapPlan::block_pos((*selBlocks)[a]) + MapPlan::block_n((*selBlocks)[a])) return;
        }
        uint32_t& v = (*acc)[q]; if (v != 0xFFFFFFFFu) ++v;
    }
    const std::vector<std::pair<uint32_t, uint32_t>>* selBlocks = nullptr;
    const uint32_t* saArr = nullptr;
    const std::vector<uint8_t>* textSent = nullptr;
    uint64_t verified = 0;
    struct Item { uint32_t p0; };
    Item item(uint32_t row) const { return Item{saArr[row]}; }
    // fast verification: the device's reads restated on the host -- three 16-byte chunks of the 4-bit text around the window (here the
    // slice's own packing: any alignment occurs) and the row's 56-symbol record -- through the SAME mask builder (gm_engine.h: fv_masks)
    typedef MaskItemT<uint32_t> MaskItem;
    uint64_t textAvail = 0;   // bytes readable from `text`
    MaskItem mask_item(uint32_t row, uint32_t meta, const Root& rt) const
    {
        uint32_t c[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, r[7] = {0, 0, 0, 0, 0, 0, 0};
        const uint64_t g = rt.win, g0 = g & ~31ull;
        const uint32_t wo = (uint32_t)(g & 31u), W = K + rt.n - 1;
        const uint32_t chunks = wo + W > 64u ? 3u : 2u;           // (the device requests the third chunk only when the window reaches it)
        for (uint32_t k = 0; k < 32u * chunks; ++k) { const uint64_t p = g0 + k; const uint32_t code = p < textAvail ? text[p] : 0u; c[k >> 3] |= (code & 15u) << (4u * (k & 7u)); }
        const uint32_t p0 = saArr[row];
        for (int32_t i = 0; i < CTX_SYMS; ++i) {
            const int64_t idx = (int64_t)p0 - CTX_LEFT + i;
            const uint32_t code = (idx < 0 || idx >= (int64_t)ix->n) ? (uint32_t)SYM_SENT : (*textSent)[idx];
            r[i >> 3] |= code << (4u * (i & 7u));
        }
        MaskItem it; it.p0 = p0;
        fv_masks<NL>(c, wo, W, rt.strand, r, meta_a(meta), it.mm, it.st);
        ++g_fastItems;
        return it;
    }
    uint64_t needle8(const Root& rt, uint32_t q, bool down) const
    {
        uint64_t v = 0;
        const uint32_t W = K + rt.n - 1;
        for (uint32_t j = 0; j < 8; ++j) {
            const int64_t qq = down ? (int64_t)q - j : (int64_t)q + j;
            uint64_t c = 0;   // outside the window: any value (the scanner masks those bytes)
            if (qq >= 0 && qq < (int64_t)W) c = text_char(rt, (uint32_t)qq);
            v |= c << (8 * j);
        }
        return v;
    }
    uint64_t text8(const Item& it, int32_t off, bool down) const
    {
        const uint32_t p0 = it.p0;
        uint64_t v = 0;
        for (uint32_t j = 0; j < 8; ++j) {
            const int64_t idx = (int64_t)p0 + off + (down ? -(int64_t)j : (int64_t)j);
            const uint64_t c = (idx < 0 || idx >= (int64_t)ix->n) ? (uint64_t)SYM_SENT : (*textSent)[idx];
            v |= c << (8 * j);
        }
        return v;
    }
    void leaf_at(const Root& rt, uint32_t kmer, uint32_t textPos) { if (scatter) { scatter_at(textPos); return; } leafSum += 1; leaf_flush(rt, kmer); }
    const HostIndex<WPP>* ix;
    const uint8_t* text;
    uint32_t K;
    std::vector<Node> stack;
    std::vector<uint32_t>* acc;
    size_t maxDepth = 0;
    uint64_t steps = 0;
    void rank2(uint32_t right, uint32_t lo, uint32_t hi, uint32_t rl[NLET], uint32_t rh[NLET])
    {
        constexpr uint32_t SPB = BlockGeom<WPP>::SPB, WPB = BlockGeom<WPP>::WPB;
        const uint32_t* base = ix->blk[right].data();
        block_rank<WPP>(base + (size_t)(lo / SPB) * WPB, lo % SPB, rl);
        block_rank<WPP>(base + (size_t)(hi / SPB) * WPB, hi % SPB, rh);
        ++steps;
    }
    const uint64_t* pwin = nullptr;   // != null: the needle window of the packet being walked (gm_expand.h), 4 bits per symbol from nibble 0
    uint64_t textLenSlice = 0;
    uint32_t text_char(const Root& rt, uint32_t pos) const
    {
        uint32_t W = K + rt.n - 1;
        if (pwin) {
            const uint32_t nib = rt.strand ? W - 1 - pos : pos;
            const uint32_t c = (uint32_t)(pwin[nib >> 4] >> (4u * (nib & 15u))) & 15u;
            return rt.strand ? complement(c) : c;
        }
        return rt.strand ? complement(text[rt.win + (W - 1 - pos)]) : text[rt.win + pos];
    }
    void push(const Node& nd) { stack.push_back(nd); if (stack.size() > maxDepth) maxDepth = stack.size(); }
    bool saturated(const Root&, uint32_t, uint32_t) const { return false; }   // (the device's shortcut for k-mers at MAX: never needed for the result)
    void note_step(uint32_t, uint32_t) {}
    bool any(bool b) const { return b; }
    void note_chunk() {}
    void note_run() {}
    void note_wave(int) {}
    void note_item(uint32_t) {}
    uint32_t leafSum = 0;
    void leaf(const Root&, uint32_t, uint32_t flo, uint32_t w) { if (scatter) { for (uint32_t r = 0; r < w; ++r) scatter_at(saArr[flo + r]); return; } leafSum += w; }
    void leaf_flush(const Root& rt, uint32_t kmer)
    {
        uint32_t count = leafSum; leafSum = 0;
        if (!count) return;
        uint32_t pos = rt.win + (rt.strand ? rt.n - 1 - kmer : kmer);
        uint32_t add = count < 0xFFFFu ? count : 0xFFFFu;
        uint64_t v = (uint64_t)(*acc)[pos] + add;
        (*acc)[pos] = v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v;
    }
    uint32_t C(uint32_t c) const { return ix->C[c]; }
};

// SA range of an ACGT string (most significant symbol first in idx) by right extensions from the root: what the device reads from
// its table of all J-mers (gm_kernels.h: qmer_table_kernel)
template <int WPP>
static void table_entry(const HostIndex<WPP>& ix, uint64_t idx, uint32_t J, uint32_t& flo, uint32_t& rlo, uint32_t& w)
{
    constexpr uint32_t SPB = BlockGeom<WPP>::SPB, WPB = BlockGeom<WPP>::WPB;
    flo = 0; rlo = 0; w = (uint32_t)ix.n;
    for (uint32_t i = 0; i < J && w; ++i) {
        const uint32_t c = (uint32_t)(idx >> (2u * (J - 1u - i))) & 3u;
        uint32_t rl[NLET], rh[NLET];
        block_rank<WPP>(ix.blk[1].data() + (size_t)(rlo / SPB) * WPB, rlo % SPB, rl);
        block_rank<WPP>(ix.blk[1].data() + (size_t)((rlo + w) / SPB) * WPB, (rlo + w) % SPB, rh);
        uint32_t tot = 0, below = 0;
        for (uint32_t x = 0; x < NLET; ++x) { const uint32_t cx = rh[x] - rl[x]; tot += cx; if (x < c) below += cx; }
        flo += (w - tot) + below; rlo = ix.C[c] + rl[c]; w = rh[c] - rl[c];
    }
}

// all roots of a plan through one environment; jumpCap > 0: regular blocks whose jump region holds no N start from the patterns
template <int WPP, class Env>
static void search_plan(const MapPlan& plan, const HostIndex<WPP>& ix, Env& env, uint32_t K, uint32_t E, uint64_t rows, uint32_t verifyT, uint32_t jumpCap, uint64_t* nPatterns)
{
    std::vector<JumpSearch> jumps(plan.nSearches);
    const uint32_t L = K - plan.stepSize + 1;
    if (jumpCap && E >= 1) {
        for (uint32_t J = std::min(jumpCap, L - 1u); J >= 1; --J) {   // one J for every search: the largest whose pattern lists stay small
            bool ok = true;
            for (uint32_t s = 0; s < plan.nSearches && ok; ++s) ok = oss_jump_patterns(E, plan.table[(size_t)(plan.stepSize - 1) * 8 + s], L, J, 4096, &jumps[s]) && !jumps[s].pat.empty();   // (the rule of gm_api.hip: prepare_search)
            if (ok) break;
            for (auto& j : jumps) j = JumpSearch();
        }
    }
    // what a lane reads: rotation words and groups (gm_oss.h: oss_make_items); with g_jumpGroups every group two patterns can form
    std::vector<uint64_t> gmasks;
    std::vector<SearchItems> items(plan.nSearches);
    for (uint32_t s = 0; s < plan.nSearches; ++s)
        if (jumps[s].J) oss_make_items(jumps[s], E, g_jumpGroups ? 1 : 0, jumps[s].regionA + jumps[s].J + 2u <= L, 0.5, 0.05, &gmasks, &items[s], g_jumpGroups > 1 ? 0xFFu : 1u);
    uint64_t roots = plan.numRoots();
    uint32_t rpb = plan.nSearches * plan.nStrands;
    // the rule of gm_api.hip (prepare_search): every text symbol an item may look at lies inside its row's record
    uint32_t maxA0 = 0;
    for (uint32_t n = 1; n <= plan.stepSize; ++n) for (uint32_t s = 0; s < plan.nSearches; ++s) maxA0 = std::max(maxA0, n - 1u + oss_start(plan.table[(size_t)(n - 1) * 8 + s]));
    const bool fastOK = g_fastVerify && K <= 32u && K + plan.stepSize - 1u <= FV_MAXW && maxA0 <= (uint32_t)CTX_LEFT;
    auto walk = [&](Node nd, const Root& rt) {
        bool have = true;
        for (;;) {
            if (!have) { if (env.stack.empty()) break; nd = env.stack.back(); env.stack.pop_back(); have = true; }
            if (Env::RANGE_ADD && g_selfHit && nd.w == 1u && rt.strand == 0u && meta_errs(nd.meta) == 0u && nd.rlo != ~0u) {   // gm_kernels.h: self hits
                bool anyN = false;
                for (uint32_t i = 0; i < K + rt.n - 1u; ++i) anyN |= env.text_char(rt, i) >= SYM_N;
                if (!anyN) {
                    uint32_t smin, smax;
                    if (self_hit_kmers(nd.meta, rt, K, smin, smax)) { if constexpr (Env::RANGE_ADD) env.leaf_range(rt, smin, smax); }
                    env.selfHits++; have = false; continue;
                }
            }
            if (env.saArr && verifyT && nd.w <= verifyT) {   // the device defers these to a wave-wide verification round
                for (uint32_t r2 = 0; r2 < nd.w; ++r2) {
                    if (fastOK) { const typename Env::MaskItem mi = env.mask_item(nd.flo + r2, nd.meta, rt); verify_with(mi, nd.meta, rt, K, E, env); }
                    else verify_item(nd.flo + r2, nd.meta, rt, K, E, env);
                }
                env.verified += nd.w; have = false; continue;
            }
            if (meta_mode(nd.meta) == M_SPLIT) { Node left; split_node(nd, left, K); env.push(left); }
            lane_step(nd, have, rt, K, E, env);
        }
    };
    bool anyJump = false;
    for (uint32_t s = 0; s < plan.nSearches; ++s) anyJump = anyJump || jumps[s].J != 0;
    if (g_expand && anyJump) {
        // ---- phase A / phase B (gm_expand.h): what expand_kernel and the walker do, in the device's order ----
        const uint32_t J = jumps[0].J;
        // the call's items back to back, the searches' records as gm_api.hip (prepare_search) lays them out
        std::vector<uint32_t> pat, first(plan.nSearches), count(plan.nSearches), nbWord(plan.nSearches, 0u);
        std::vector<uint32_t> ex(plan.nSearches), ey(plan.nSearches), ez(plan.nSearches);
        const bool canFilter = g_nbFilter != 0 && env.saArr && env.textSent;
        for (uint32_t s = 0; s < plan.nSearches; ++s) {
            first[s] = (uint32_t)pat.size(); count[s] = (uint32_t)items[s].items.size();
            uint32_t e16[GROUP_MAX_LAYOUTS], run = first[s];
            for (uint32_t L2 = 0; L2 < GROUP_MAX_LAYOUTS; ++L2) { run += L2 < items[s].seg.size() ? items[s].seg[L2] : 0u; e16[L2] = run; }
            ex[s] = e16[0] | e16[1] << 16; ey[s] = e16[2] | e16[3] << 16; ez[s] = e16[4] | e16[5] << 16;
            if (canFilter) {
                const uint32_t nr = std::min<uint32_t>(NB_SYMS, L - jumps[s].regionA - J), nl = std::min<uint32_t>(NB_SYMS, jumps[s].regionA);
                for (uint32_t i = 0; i < nr; ++i) nbWord[s] |= 1u << (2u * i);
                for (uint32_t i = 0; i < nl; ++i) nbWord[s] |= 1u << (16u + 2u * i);
                nbWord[s] |= nr << 12 | nl << 28 | 1u << 31;
            }
            pat.insert(pat.end(), items[s].items.begin(), items[s].items.end());
        }
        const std::vector<uint32_t> wmap = make_wmap(plan.nStrands, plan.nSearches, first.data(), count.data());
        const std::vector<uint32_t> shifts = group_layout_shifts(J);
        struct HostTab {
            const std::vector<uint32_t>* shifts; const std::vector<uint64_t>* masks;
            uint32_t layout(uint32_t il) const { return il < shifts->size() ? ((*shifts)[il] | il << 5 | (1u + 16u * il) << 13) : 0u; }
            uint64_t mask(uint32_t id) const { return id < masks->size() ? (*masks)[id] : 0ull; }
        } tab{&shifts, &gmasks};
        struct EmuMem {
            const uint8_t* text; uint64_t avail;
            uint64_t word(uint64_t i) const { uint64_t v = 0; for (uint32_t k = 0; k < 16u; ++k) { const uint64_t p = 16u * i + k; const uint64_t c = p < avail ? text[p] : 0u; v |= (c & 15u) << (4u * k); } return v; }
            void pair(uint64_t i, uint64_t& lo, uint64_t& hi) const { lo = word(i); hi = word(i + 1); }
        } mem{env.text, env.textAvail ? env.textAvail : env.textLenSlice};
        // the 4th word of a table entry (gm_kernels.h: qmer_table_kernel): the text next to the occurrence(s) of a one- / two-row entry
        auto sent = [&](int64_t p) -> uint32_t { return (p < 0 || p >= (int64_t)ix.n) ? (uint32_t)SYM_SENT : (uint32_t)(*env.textSent)[p]; };
        auto entry_nb = [&](uint32_t flo, uint32_t w) -> uint32_t {
            uint32_t nb = 0;
            if (!canFilter) return 0u;
            if (w == 1u) {
                const int64_t t = env.saArr[flo];
                uint32_t nr = 0, nl = 0;
                while (nr < NB_SYMS && sent(t + J + nr) < SYM_N) { nb |= sent(t + J + nr) << (2u * nr); ++nr; }
                while (nl < NB_SYMS && sent(t - 1 - (int64_t)nl) < SYM_N) { nb |= sent(t - 1 - (int64_t)nl) << (16u + 2u * nl); ++nl; }
                nb |= nr << 12 | nl << 28 | 0x80008000u;
            }
            if (w == 2u) for (uint32_t r2 = 0; r2 < 2u; ++r2) {
                const int64_t t = env.saArr[flo + r2];
                uint32_t x = 0, nr = 0, nl = 0;
                while (nr < NB_SYMS2 && sent(t + J + nr) < SYM_N) { x |= sent(t + J + nr) << (2u * nr); ++nr; }
                while (nl < NB_SYMS2 && sent(t - 1 - (int64_t)nl) < SYM_N) { x |= sent(t - 1 - (int64_t)nl) << (6u + 2u * nl); ++nl; }
                nb |= (x | nr << 12 | nl << 14) << (16u * r2);
            }
            return nb;
        };
        struct Pkt { Node nd; uint32_t win, nss; std::vector<uint64_t> w; };
        std::vector<Pkt> lists[3];

        const uint64_t nBlocks = plan.useList ? plan.blocks.size() : plan.numBlocks;
        auto packet = [&](uint32_t cls, const Node& nd, const Root& rt) {
            Pkt p; p.nd = nd; p.win = rt.win; p.nss = pkt_root_word(rt.n, rt.strand, rt.search);
            const uint32_t words = 2u * pkt_chunks_for(K, plan.stepSize);
            for (uint32_t j = 0; j < words; ++j) p.w.push_back(nib64(mem, (uint64_t)rt.win + 16u * j));
            lists[cls].push_back(p); g_packets[cls]++;
        };
        // mode 0: every pattern of every root in one pass; 1: the patterns without a substitution, a work item per root (the J-mer's own table
        // entry); 2: everything else (gm_expand.h: expand_strip_exact).  g_expand == 2 runs 1 and 2 one after the other, like gm_api.hip: run_expand.
        const uint32_t exactItem = (uint32_t)pat.size();
        pat.push_back(0u);
        std::vector<uint32_t> wmap1;
        for (uint32_t st = 0; st < plan.nStrands; ++st) for (uint32_t s2 = 0; s2 < plan.nSearches; ++s2) {
            bool hasExact = false;
            for (uint32_t d : jumps[s2].pat) hasExact = hasExact || (d & 7u) == 0u;
            wmap1.push_back(wmap_pack(s2, st, exactItem) | (hasExact ? 0u : WMAP_ROOT_ONLY));
        }
        auto phaseA = [&](uint32_t mode) {
        const std::vector<uint32_t>& wm_ = mode == 1u ? wmap1 : wmap;
        const uint32_t ipbm = (uint32_t)wm_.size();
        for (uint64_t b = 0; b < nBlocks; ++b) for (uint32_t q = 0; q < ipbm; ++q) {
            const uint32_t wm = wm_[q], search = wm & 7u, strand = (wm >> 3) & 1u, jp = wm >> 8;
            Root rt;
            if (plan.useList) { rt.win = (uint32_t)MapPlan::block_pos(plan.blocks[b]); rt.n = MapPlan::block_n(plan.blocks[b]); }
            else { rt.win = (uint32_t)(b * plan.stepSize); rt.n = (uint32_t)std::min<uint64_t>(plan.stepSize, plan.numKmers - rt.win); }
            rt.strand = strand; rt.search = search; rt.rec = plan.table[(size_t)(rt.n - 1) * 8 + search];
            const JumpSearch& js = jumps[search];
            const uint32_t W = K + rt.n - 1u;
            XRoot xr; xr.jb = xr.jn = xr.ext = 0u; xr.bad = 1u;
            if (rt.n == plan.stepSize) xr = expand_root(mem, rt.win, W, strand, rt.n - 1u + js.regionA, J, nbWord[search], items[search].ext);
            if (xr.bad) {   // an odd block shape, or an N inside the J-mer: the root walks the tree from its root (its first item's lane says so; with two passes the first)
                if (mode == 1u || (mode == 0u && jp == first[search])) packet(0u, root_node(rt, (uint32_t)rows), rt);
                continue;
            }
            const uint32_t jd = pat[jp];
            XItem it = expand_item(jd, jump_item_flags(jp, ex[search], ey[search], ez[search]), xr, tab);
            if (it.state == 2u) {
                // the word of the group's bitmap, built here by asking the index (the device reads it from its bitmaps); kind from the plane
                const uint32_t kind = (jd >> (it.sh + 3u)) & 1u, e0 = (xr.ext >> (JF_EXT_SHIFT + 2u)) & 3u, e1 = (xr.ext >> JF_EXT_SHIFT) & 3u;
                const uint32_t pre = rot_add(xr.jb, it.gcur);
                if (group_word(pre, it.sh) != it.widx) g_hangs += 1000000;
                uint64_t word = 0;
                for (uint32_t c = 0; c < 64u; ++c) {
                    const uint64_t cand = (pre & ~(63u << it.sh)) | c << it.sh;
                    uint32_t f, r2, w;
                    if (kind) table_entry<WPP>(ix, cand << 4 | e0 << 2 | e1, J + 2u, f, r2, w); else table_entry<WPP>(ix, cand, J, f, r2, w);
                    if (w) word |= 1ull << c;
                }
                expand_word(it, word, jd, xr, tab);
            }
            if (mode == 2u) expand_strip_exact(it);
            if (wm & WMAP_ROOT_ONLY) it.state = 0u;
            const uint32_t off = rt.n - 1u;
            const uint32_t jm0 = meta_pack((js.meta0 & 0x1FFu) + off, ((js.meta0 >> 9) & 0x1FFu) + off, js.meta0 >> 18, 0u, M_OSS);
            const XItem it0 = it;
            for (uint32_t kk = 0, nrot = expand_count(it0); kk < nrot; ++kk) {
                // the device deals the rotations of 64 items out to its lanes (expand_nth: the k-th surviving rotation); the item's own iterator must agree
                const uint32_t rw = expand_nth(it0.gcur, it0.sh, it0.state == 1u, it0.alive, kk);
                { uint32_t rw2 = 0; if (!expand_next(it, rw2) || rw2 != rw) g_hangs += 1000000; }
                if (mode == 1u && rw != 0u) g_hangs += 1000000;
                if (mode == 2u && rw == 0u) g_hangs += 1000000;
                uint32_t flo, rlo, w;
                table_entry<WPP>(ix, rot_add(xr.jb, rw), J, flo, rlo, w);
                if (nPatterns) ++*nPatterns;
                const XNode x = expand_filter(flo, rlo, w, entry_nb(flo, w), rw, jm0, xr.jn, nbWord[search], E, (uint32_t)g_nbFilter, env.saArr ? verifyT : 0u);
                if (!x.take) { if (w) g_packets[3]++; continue; }
                Node nd; nd.flo = x.flo; nd.rlo = x.rlo; nd.w = x.w; nd.meta = x.meta;
                packet(expand_class(x.errs), nd, rt);
            }
        }
        };
        auto walkLists = [&]() {
        for (uint32_t cls = 0; cls < 3u; ++cls) {
            for (const Pkt& p : lists[cls]) {
                Root rt; rt.win = p.win; rt.n = p.nss & 0xFFu; rt.strand = (p.nss >> 8) & 1u; rt.search = (p.nss >> 9) & 7u;
                rt.rec = plan.table[(size_t)(rt.n - 1) * 8 + rt.search];
                env.pwin = p.w.data();
                walk(p.nd, rt);
            }
            lists[cls].clear();
        }
        };
        if (g_expand >= 2) { phaseA(1u); walkLists(); phaseA(2u); walkLists(); }
        else { phaseA(0u); walkLists(); }
        env.pwin = nullptr;
        return;
    }
    for (uint64_t id = 0; id < roots; ++id) {
        uint64_t b = id / rpb; uint32_t r = (uint32_t)(id % rpb);
        Root rt;
        if (plan.useList) { rt.win = (uint32_t)MapPlan::block_pos(plan.blocks[b]); rt.n = MapPlan::block_n(plan.blocks[b]); }
        else { rt.win = (uint32_t)(b * plan.stepSize); rt.n = (uint32_t)std::min<uint64_t>(plan.stepSize, plan.numKmers - rt.win); }
        rt.strand = r / plan.nSearches;
        rt.search = r % plan.nSearches;
        rt.rec = plan.table[(size_t)(rt.n - 1) * 8 + rt.search];
        const JumpSearch& js = jumps[rt.search];
        bool jumped = false;
        if (js.J && rt.n == plan.stepSize) {
            const uint32_t a0 = rt.n - 1u + js.regionA;
            uint32_t base2 = 0; bool bad = false;
            for (uint32_t i = 0; i < js.J; ++i) { const uint32_t c = env.text_char(rt, a0 + i); if (c >= SYM_N) bad = true; base2 = base2 << 2 | (c & 3u); }
            if (!bad) {
                jumped = true;
                auto lookup = [&](uint32_t idx, uint32_t nsub) {
                    Node nd;
                    table_entry<WPP>(ix, idx, js.J, nd.flo, nd.rlo, nd.w);
                    if (nPatterns) ++*nPatterns;
                    if (!nd.w) return;
                    const uint32_t off = rt.n - 1u;
                    nd.meta = meta_pack((js.meta0 & 0x1FFu) + off, ((js.meta0 >> 9) & 0x1FFu) + off, js.meta0 >> 18, nsub, M_OSS);
                    walk(nd, rt);
                };
                const SearchItems& it = items[rt.search];
                // the two letters behind the J-mer (groups of kind 1); a needle N there ends every pattern that may not err any more
                uint32_t e0 = SYM_N, e1 = SYM_N;
                if (it.ext) { e0 = env.text_char(rt, a0 + js.J); e1 = env.text_char(rt, a0 + js.J + 1u); }
                const bool extValid = e0 < SYM_N && e1 < SYM_N;
                const std::vector<uint32_t> shifts = group_layout_shifts(js.J);
                // the word of a group's bitmap, built here by asking the index (the device reads it from its bitmaps)
                auto group_bits = [&](uint32_t pre, uint32_t shift, uint32_t kind) {
                    uint64_t word = 0;
                    for (uint32_t c = 0; c < 64u && (kind == 0u || extValid); ++c) {
                        const uint64_t cand = (pre & ~(63u << shift)) | c << shift;
                        uint32_t f, r, w;
                        if (kind) table_entry<WPP>(ix, cand << 4 | e0 << 2 | e1, js.J + 2u, f, r, w); else table_entry<WPP>(ix, cand, js.J, f, r, w);
                        if (w) word |= 1ull << c;
                    }
                    return word;
                };
                if (g_stateMachine) {
                    // The lane's pattern-fetch state machine itself (gm_oss.h: jump_decide -- the code part B of the kernel's loop runs), one
                    // "iteration" at a time with the device's timing: a word asked for in one iteration is looked at in the next, a table
                    // entry read in one iteration becomes a node in the next.  A root that has not finished its items after a generous
                    // number of iterations is a hang of exactly the kind that reached the GPU in round 4: reported, not waited for.
                    struct HostTab {
                        const std::vector<uint32_t>* shifts; const std::vector<uint64_t>* masks;
                        uint32_t layout(uint32_t il) const { return il < shifts->size() ? ((*shifts)[il] | il << 5 | (1u + 16u * il) << 13) : 0u; }   // (planes: any distinct numbers)
                        uint64_t mask(uint32_t id) const { return id < masks->size() ? (*masks)[id] : 0ull; }
                    } tab{&shifts, &gmasks};
                    uint32_t e16[GROUP_MAX_LAYOUTS], run = 0;
                    for (uint32_t L2 = 0; L2 < GROUP_MAX_LAYOUTS; ++L2) { run += L2 < it.seg.size() ? it.seg[L2] : 0u; e16[L2] = run; }
                    const uint32_t ex = e16[0] | e16[1] << 16, ey = e16[2] | e16[3] << 16, ez = e16[4] | e16[5] << 16;
                    uint32_t fs = 2u | jump_item_flags(0u, ex, ey, ez), jd = it.items[0], jpp = 1u | (uint32_t)it.items.size() << 16, gcur = 0;
                    if (it.ext && extValid) fs |= JF_EXTOK | (e0 << 2 | e1) << JF_EXT_SHIFT;
                    unsigned long long galive = 0, pw = 0;
                    uint32_t entryIdx = 0, entryErrs = 0;
                    const size_t cap = 8 * it.items.size() + 64 * 70 * it.items.size() + 16;
                    size_t iter = 0;
                    for (;; ++iter) {
                        if (iter > cap) { g_hangs++; break; }
                        if (fs & JF_ENTRY) { fs &= ~JF_ENTRY; lookup(entryIdx, entryErrs); }            // part A: the entry has arrived
                        const uint32_t fsBefore = fs;
                        const JumpStep D = jump_decide(fs, jd, gcur, galive, pw, base2, tab);
                        if (D.want) { const uint32_t jp = jpp & 0xFFFFu; if (jp < (jpp >> 16)) { jd = it.items[jp]; jpp += 1u; fs |= jump_item_flags(jp, ex, ey, ez); } }
                        if (D.asked) {
                            // (an asked-for word leaves jd alone: it still holds the group item)
                            const uint32_t il = (fsBefore >> JF_IL_SHIFT) & 7u, ly = tab.layout(il), sh = ly & 31u, kind = (jd >> (sh + 3u)) & 1u;
                            // the word's address as the kernel would form it: plane of the kind (and of the two letters), word of the layout
                            if (kind ? D.wsel != ((ly >> 13) & 255u) + (e0 << 2 | e1) : D.wsel != ((ly >> 5) & 255u)) g_hangs += 1000000;
                            const uint32_t pre = rot_add(base2, jd & ~(63u << sh));
                            if (group_word(pre, sh) != D.widx) g_hangs += 1000000;
                            pw = group_bits(pre, sh, kind);
                            fs |= JF_WORD;
                        }
                        if (D.go) { entryIdx = rot_add(base2, D.rw); entryErrs = rot_errors(D.rw); fs |= JF_ENTRY; }
                        if (jump_done(fs, galive)) break;
                    }
                } else
                for (size_t q = 0; q < it.items.size(); ++q) {
                    const uint32_t d = it.items[q];
                    if (q >= it.groups()) { lookup(rot_add(base2, d), rot_errors(d)); continue; }
                    // a group: the word of its bitmap, in rotation space, masked; only patterns that pass are looked up
                    size_t lay = 0, cum = it.seg[0];
                    while (q >= cum) cum += it.seg[++lay];
                    const uint32_t shift = shifts[lay], own = (d >> shift) & 63u, kind = own >> 3, rotw = d & ~(63u << shift);
                    const uint32_t pre = rot_add(base2, rotw);
                    uint64_t alive = word_to_rotations(group_bits(pre, shift, kind), (base2 >> shift) & 63u) & gmasks[own & 7u];
                    while (alive) {
                        const uint32_t rw = rotw | (uint32_t)__builtin_ctzll(alive) << shift; alive &= alive - 1ull;
                        lookup(rot_add(base2, rw), rot_errors(rw));
                    }
                }
            }
        }
        if (!jumped) walk(root_node(rt, (uint32_t)rows), rt);
    }
}

// K > MAX_K: the per-node code of the long k-mer kernel (gm_longk_step.h), every root through a plain LIFO
template <int WPP, class Env>
static void search_plan_long(const MapPlan& plan, Env& env, uint32_t K, uint32_t E, uint64_t rows, uint32_t verifyT)
{
    typedef LNodeT<uint32_t> LN;
    const uint32_t rpb = plan.nSearches * plan.nStrands;
    std::vector<LN> st;
    for (uint64_t id = 0; id < plan.numRoots(); ++id) {
        const uint64_t b = id / rpb; const uint32_t r = (uint32_t)(id % rpb);
        Root rt;
        if (plan.useList) { rt.win = (uint32_t)MapPlan::block_pos(plan.blocks[b]); rt.n = MapPlan::block_n(plan.blocks[b]); }
        else { rt.win = (uint32_t)(b * plan.stepSize); rt.n = (uint32_t)std::min<uint64_t>(plan.stepSize, plan.numKmers - rt.win); }
        rt.strand = r / plan.nSearches; rt.search = r % plan.nSearches; rt.rec = OssRecord{0, 0, 0, 0};
        const OssRecordL& rec = plan.tableL[(size_t)(rt.n - 1) * 8 + rt.search];
        LN nd = long_root_node<uint32_t>(rt.n, rec, (uint32_t)rows);
        bool have = true;
        for (;;) {
            if (!have) { if (st.empty()) break; nd = st.back(); st.pop_back(); have = true; }
            const uint32_t w0 = nd.w;
            long_node(nd, have, rt, rec, K, E, env.saArr ? verifyT : 0u, 0xFFFFFFFFu, env,
                      [&](uint32_t pos) { return env.text_char(rt, pos); },
                      [&](const LN& x) { st.push_back(x); if (st.size() > env.maxDepth) env.maxDepth = st.size(); });
            if (env.saArr && verifyT && w0 <= verifyT) env.verified += w0;
        }
    }
}

template <int WPP, bool NL>
static int run(const uint8_t* bf, const uint8_t* br, uint64_t rows, uint32_t nseqTotal, const uint8_t* text, uint64_t textLen,
               const uint64_t* seqCum, uint32_t nseqLocal, uint32_t K, uint32_t E, uint32_t infix, int revcompl, int valueBits,
               const uint64_t* intervals, uint64_t nIntervals, void* out, uint64_t* stats, const uint32_t* sa, uint32_t verifyT,
               const uint8_t* allCodes, const uint64_t* allCum, uint32_t jumpCap)
{
    MapPlan plan;
    int rc = make_map_plan(K, E, infix, revcompl, textLen, intervals, nIntervals, &plan, 0, g_ossWeights);
    if (rc) return rc;
    HostIndex<WPP> ix; ix.build(bf, br, rows, nseqTotal);
    std::vector<uint32_t> acc(textLen ? textLen : 1, 0);
    EmuEnv<WPP, NL, true> env; env.ix = &ix; env.text = text; env.K = K; env.acc = &acc; env.textLenSlice = textLen;
    std::vector<uint32_t> diff(textLen + 1, 0);
    if (!plan.useList) env.diff = &diff;
    std::vector<uint8_t> textS;
    if (sa && (verifyT || NL)) {   // sentinel text: sequence s occupies [cum[s] + s, cum[s+1] + s), sentinel after it
        textS.resize(rows);
        for (uint32_t q = 0; q < nseqTotal; ++q) {
            for (uint64_t i = allCum[q]; i < allCum[q + 1]; ++i) textS[i + q] = allCodes[i];
            textS[allCum[q + 1] + q] = (uint8_t)SYM_SENT;
        }
        env.saArr = sa; env.textSent = &textS;
        // (the slice is either a view into the whole text -- what follows it is readable, as on the device -- or a copy of its own)
        env.textAvail = (text >= allCodes && text < allCodes + allCum[nseqTotal]) ? allCum[nseqTotal] - (uint64_t)(text - allCodes) : textLen;
    }
    uint32_t bound = stack_bound(E, plan.stepSize);
    uint64_t nPatterns = 0;
    if (K > MAX_K) { env.diff = nullptr; search_plan_long<WPP>(plan, env, K, E, rows, verifyT); }   // (verified runs k-mer by k-mer, as the device does there)
    else search_plan<WPP>(plan, ix, env, K, E, rows, verifyT, jumpCap, &nPatterns);
    uint64_t corrRoots = 0;
    if (NL && E >= 1) {
        // correction pass: the text windows with N as needles, full semantics (N children followed), every located occurrence
        // inside the slice gets one hit (gm_api.hip does the same with ScatterEnv)
        if (!sa) return -101;
        const uint64_t allLen = allCum[nseqTotal];
        std::vector<std::pair<uint64_t, uint64_t>> runs;
        for (uint64_t i = 0; i < allLen; ) { if (allCodes[i] != SYM_N) { ++i; continue; } uint64_t e = i; while (e < allLen && allCodes[e] == SYM_N) ++e; runs.emplace_back(i, e); i = e; }
        std::vector<uint64_t> cumv(allCum, allCum + nseqTotal + 1), iv;
        n_window_intervals(runs, cumv, K, E, iv);
        if (!iv.empty()) {
            MapPlan cplan;
            rc = make_map_plan(K, E, infix, revcompl, allLen, iv.data(), iv.size() / 2, &cplan, 0, g_ossWeights);
            if (rc) return rc;
            EmuEnv<WPP, false> cenv; cenv.ix = &ix; cenv.text = allCodes; cenv.K = K; cenv.acc = &acc;
            cenv.saArr = sa; cenv.textSent = &textS; cenv.textAvail = allLen;
            if (plan.useList) cenv.selBlocks = &plan.blocks;
            cenv.scatter = true; cenv.cumAll = allCum; cenv.nSeqAll = nseqTotal; cenv.sliceBegin = (uint64_t)(text - allCodes); cenv.sliceLen = textLen;
            search_plan<WPP>(cplan, ix, cenv, K, E, rows, verifyT, 0, nullptr);
            corrRoots = cplan.numRoots();
            env.steps += cenv.steps; env.verified += cenv.verified; env.maxDepth = std::max(env.maxDepth, cenv.maxDepth);
        }
    }
    uint32_t maxv = valueBits == 8 ? 255u : 65535u;
    uint32_t run = 0;
    for (uint64_t j = 0; j < textLen; ++j) {
        if (j % plan.stepSize == 0) run = 0;               // gm_kernels.h: finalize_diff_kernel
        run += diff[j];
        const uint64_t tot = (uint64_t)acc[j] + (env.diff ? run : 0u);
        uint32_t v = tot < maxv ? (uint32_t)tot : maxv;
        if (valueBits == 8) ((uint8_t*)out)[j] = (uint8_t)v; else ((uint16_t*)out)[j] = (uint16_t)v;
    }
    for (uint32_t s = 1; s <= nseqLocal; ++s) {   // resetLimits, algo.hpp:10-22
        uint64_t lim = std::min<uint64_t>(K, seqCum[s] - seqCum[s - 1] + 1);
        for (uint64_t j = 1; j < lim; ++j) { if (valueBits == 8) ((uint8_t*)out)[seqCum[s] - j] = 0; else ((uint16_t*)out)[seqCum[s] - j] = 0; }
    }
    if (stats) { stats[0] = env.maxDepth; stats[1] = bound; stats[2] = env.steps; stats[3] = env.verified; stats[4] = nPatterns; stats[5] = corrRoots; stats[6] = env.selfHits; }
    return 0;
}


// gm_oss.h self-check (no index involved): for every search of (K, E) under the library's block shape and jump length J, the items of
// oss_make_items (plain / every possible group / groups by the occurrence rule) must expand to exactly the patterns of oss_jump_patterns --
// same substituted J-mers with the same error counts -- for `reps` random needles; groups of kind 1 may only hold patterns that have spent
// the whole budget; word_to_rotations / rot_add must agree with their definitions.  Returns 0, or a code that names the first failure.
extern "C" int gm_emu_check_items2(uint32_t K, uint32_t E, uint32_t J, uint32_t ossWeights, uint32_t reps, uint32_t seed, uint64_t* stats, uint32_t layouts);
extern "C" int gm_emu_check_items(uint32_t K, uint32_t E, uint32_t J, uint32_t ossWeights, uint32_t reps, uint32_t seed, uint64_t* stats)
{
    return gm_emu_check_items2(K, E, J, ossWeights, reps, seed, stats, 1u);
}
// layouts: bit L = groups of layout L may be of kind 0 (1: the LOW / MID rule of round 4; 0xFF: every layout)
extern "C" int gm_emu_check_items2(uint32_t K, uint32_t E, uint32_t J, uint32_t ossWeights, uint32_t reps, uint32_t seed, uint64_t* stats, uint32_t layouts)
{
    uint64_t x = 88172645463325252ull ^ ((uint64_t)seed << 32 | seed);
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int t = 0; t < 200; ++t) {
        const uint64_t w = rnd(); const uint32_t low6 = (uint32_t)rnd() & 63u;
        const uint64_t r = word_to_rotations(w, low6);
        for (uint32_t rot = 0; rot < 64u; ++rot) {
            const uint32_t c0 = ((low6 >> 4) + (rot >> 4)) & 3u, c1 = (((low6 >> 2) & 3u) + ((rot >> 2) & 3u)) & 3u, c2 = ((low6 & 3u) + (rot & 3u)) & 3u, b = c0 << 4 | c1 << 2 | c2;
            if ((rot_add(low6, rot) & 63u) != b) return 1;
            if (((r >> rot) & 1ull) != ((w >> b) & 1ull)) return 2;
        }
        const uint32_t idx = (uint32_t)rnd();
        if (jump_swap_mid(jump_swap_mid(idx)) != idx || (jump_swap_mid(idx) & 63u) != ((idx >> 6) & 63u)) return 3;
    }
    const uint32_t infix = tuned_infix_length(K, E);
    MapPlan plan;
    if (infix == 0 || make_map_plan(K, E, infix, 1, 100000, nullptr, 0, &plan, 0, ossWeights)) return -1;
    if (J >= plan.infix) return -2;
    uint64_t nPat = 0, nItems = 0, nGroups = 0;
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<uint64_t> masks;
        for (uint32_t s = 0; s < plan.nSearches; ++s) {
            JumpSearch js;
            if (!oss_jump_patterns(E, plan.table[(size_t)(plan.stepSize - 1) * 8 + s], plan.infix, J, 4096, &js) || js.pat.empty()) continue;
            SearchItems it;
            oss_make_items(js, E, mode, js.regionA + J + 2u <= plan.infix, 0.51, 0.044, &masks, &it, layouts);
            if (it.groups() > it.items.size() || masks.size() > GROUP_MAX_MASKS) return 4;
            if (mode == 2) { nPat += js.pat.size(); nItems += it.items.size(); nGroups += it.groups(); }
            const std::vector<uint32_t> shifts = group_layout_shifts(J);
            for (uint32_t rep = 0; rep < reps; ++rep) {
                const uint32_t base = J == 16u ? (uint32_t)rnd() : (uint32_t)rnd() & ((1u << (2u * J)) - 1u);
                std::vector<uint64_t> A, B;   // (substituted J-mer, errors) of either form
                for (uint32_t d : js.pat) A.push_back((uint64_t)jump_apply(base, d, J) << 8 | (d & 7u));
                for (size_t q = 0; q < it.items.size(); ++q) {
                    const uint32_t d = it.items[q];
                    if (q >= it.groups()) { B.push_back((uint64_t)rot_add(base, d) << 8 | rot_errors(d)); continue; }
                    size_t lay = 0, cum = it.seg[0];
                    while (q >= cum) cum += it.seg[++lay];
                    const uint32_t sh = shifts[lay], own = (d >> sh) & 63u, rotw = d & ~(63u << sh);
                    if ((own >> 3) && lay >= 2) return 7;   // kind 1 exists for LOW and MID only
                    // word / bit of the layout's bitmap: all 64 members share the word, and their bits are their three characters
                    for (uint32_t c = 0; c < 64u; ++c) {
                        const uint32_t cand = (rot_add(base, rotw) & ~(63u << sh)) | c << sh;
                        if (group_word(cand, sh) != group_word(rot_add(base, rotw), sh) || ((cand >> sh) & 63u) != c) return 8;
                    }
                    if (sh == 0u && group_word(base, 0u) != base >> 6) return 9;
                    if (sh == 6u && group_word(base, 6u) != jump_swap_mid(base) >> 6) return 9;
                    if ((own & 7u) >= masks.size()) return 5;
                    for (uint32_t rot = 0; rot < 64u; ++rot) if ((masks[own & 7u] >> rot) & 1ull) {
                        const uint32_t rw = rotw | rot << sh;
                        if ((own >> 3) && rot_errors(rw) != E) return 6;   // kind 1 is only valid without budget
                        B.push_back((uint64_t)rot_add(base, rw) << 8 | rot_errors(rw));
                    }
                }
                std::sort(A.begin(), A.end()); std::sort(B.begin(), B.end());
                if (A != B) return 10 + mode;
            }
        }
    }
    if (stats) { stats[0] = nPat; stats[1] = nItems; stats[2] = nGroups; }
    return 0;
}

// gm_engine.h self-check (no index involved): fv_masks against the definition, symbol by symbol.  Random windows of W <= FV_MAXW symbols at
// every alignment inside three 16-byte chunks, either strand, needles with N, records with N and sentinels, every anchor a0 <= CTX_LEFT;
// positions whose text symbol lies behind the record are not compared (the host never enables the fast path where they could be needed).
// Also scan_side over the masks against a plain loop.  Returns 0, or a code that names the first failure.
extern "C" int gm_emu_check_fv_masks(uint32_t reps, uint32_t seed)
{
    uint64_t x = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed << 32 | seed);
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    struct NoteEnv { void note_chunk() {} void note_wave(int) {} } nenv;
    for (uint32_t rep = 0; rep < reps; ++rep) {
        uint8_t chunkSym[96], rec[CTX_SYMS];
        for (auto& v : chunkSym) { const uint64_t t = rnd() % 40; v = t == 0 ? 4 : (uint8_t)(t & 3u); }
        for (auto& v : rec) { const uint64_t t = rnd() % 50; v = t == 0 ? 4 : t == 1 ? 5 : (uint8_t)(t & 3u); }
        const uint32_t W = 1u + (uint32_t)(rnd() % FV_MAXW), wo = (uint32_t)(rnd() & 31u), strand = (uint32_t)(rnd() & 1u), a0 = (uint32_t)(rnd() % (CTX_LEFT + 1));
        if (rep & 1u) for (uint32_t i = 0; i < W && i + CTX_LEFT - a0 < (uint32_t)CTX_SYMS; ++i) {   // mostly-matching pairs: the interesting case
            const uint8_t nd = strand ? chunkSym[wo + W - 1 - i] : chunkSym[wo + i];
            const uint8_t want = strand ? (nd < 4 ? 3 - nd : nd) : nd;
            if (rnd() % 8) rec[i + CTX_LEFT - a0] = want;
        }
        uint32_t c[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, r[7] = {0, 0, 0, 0, 0, 0, 0};
        const uint32_t chunks = wo + W > 64u ? 3u : 2u;
        for (uint32_t k = 0; k < 32u * chunks; ++k) c[k >> 3] |= (uint32_t)chunkSym[k] << (4u * (k & 7u));
        for (uint32_t i = 0; i < (uint32_t)CTX_SYMS; ++i) r[i >> 3] |= (uint32_t)rec[i] << (4u * (i & 7u));
        for (int nl = 0; nl < 2; ++nl) {
            uint64_t mm = 0, st = 0;
            if (nl) fv_masks<true>(c, wo, W, strand, r, a0, mm, st); else fv_masks<false>(c, wo, W, strand, r, a0, mm, st);
            uint64_t emm = 0, est = 0, known = 0;
            for (uint32_t i = 0; i < W; ++i) {
                const uint32_t ri = i + CTX_LEFT - a0;
                if (ri >= (uint32_t)CTX_SYMS) continue;
                known |= 1ull << i;
                const uint8_t raw = strand ? chunkSym[wo + W - 1 - i] : chunkSym[wo + i];
                const uint8_t nd = strand ? (raw < 4 ? 3 - raw : raw) : raw, tx = rec[ri];
                const bool stop = tx == 5 || (nl && tx == 4);
                if (stop) est |= 1ull << i;
                if (stop || nd != tx || nd == 4) emm |= 1ull << i;
            }
            if ((mm & known) != emm) return 1 + nl;
            if ((st & known) != est) return 3 + nl;
            if (st & ~mm) return 5;
            // scans over the masks
            MaskItemT<uint32_t> it; it.p0 = 0; it.mm = emm; it.st = est;
            Root rt; rt.win = 0; rt.n = 1; rt.strand = 0; rt.search = 0; rt.rec = OssRecord{0, 0, 0, 0};
            for (int t = 0; t < 4; ++t) {
                const bool down = (rnd() & 1u) != 0;
                const uint32_t q0 = (uint32_t)(rnd() % W), need = down ? (uint32_t)(rnd() % (q0 + 2)) : (uint32_t)(rnd() % (W - q0 + 1)), budget = (uint32_t)(rnd() % 5);
                uint32_t cnt = 0, pos[4] = {9999, 9999, 9999, 9999};
                const uint32_t got = scan_side(nenv, rt, it, 0u, q0, down, need, budget, cnt, pos);
                uint32_t ecnt = 0, epos[4] = {9999, 9999, 9999, 9999}, egot = need;
                for (uint32_t i = 0; i < need; ++i) {
                    const uint32_t p = down ? q0 - i : q0 + i;
                    if (!((emm >> p) & 1ull)) continue;
                    if ((est >> p) & 1ull) { egot = i; break; }
                    if (ecnt == budget) { egot = i; break; }
                    if (ecnt < 4) epos[ecnt] = i + 1;
                    ++ecnt;
                }
                if (got != egot || cnt != ecnt) return 6;
                for (uint32_t j = 0; j < cnt && j < 4; ++j) if (pos[j] != epos[j]) return 7;
            }
        }
    }
    return 0;
}

// gm_host.h: n_window_intervals on a host text (runs of N found here): returns the number of [begin, end) pairs written (at most cap)
extern "C" uint64_t gm_emu_n_window_intervals(const uint8_t* codes, const uint64_t* cum, uint32_t nSeq, uint32_t K, uint32_t E, uint64_t* out, uint64_t cap)
{
    const uint64_t n = cum[nSeq];
    std::vector<std::pair<uint64_t, uint64_t>> runs;
    for (uint64_t i = 0; i < n; ) { if (codes[i] != SYM_N) { ++i; continue; } uint64_t e = i; while (e < n && codes[e] == SYM_N) ++e; runs.emplace_back(i, e); i = e; }
    std::vector<uint64_t> cumv(cum, cum + nSeq + 1), iv;
    n_window_intervals(runs, cumv, K, E, iv);
    for (size_t k = 0; k < iv.size() && k < 2 * cap; ++k) out[k] = iv[k];
    return iv.size() / 2;
}

// jumpCap: longest jump (0 = the plain tree walk from the root); nless != 0: the main pass never follows the text letter N and the
// correction pass adds the occurrences with N in the text (needs sa, allCodes, allCum).  stats: 7 entries.
extern "C" int gm_emu_map2(int wpp, const uint8_t* bf, const uint8_t* br, uint64_t rows, uint32_t nseqTotal, const uint8_t* text,
                           uint64_t textLen, const uint64_t* seqCum, uint32_t nseqLocal, uint32_t K, uint32_t E, int32_t xo,
                           int32_t infixOverride, int revcompl, int valueBits, const uint64_t* intervals, uint64_t nIntervals,
                           void* out, uint64_t* stats, const uint32_t* sa, uint32_t verifyT, const uint8_t* allCodes, const uint64_t* allCum,
                           uint32_t jumpCap, int nless)
{
    uint32_t infix = infixOverride > 0 ? (uint32_t)infixOverride : (xo >= 0 ? default_infix_length(K, E, xo) : tuned_infix_length(K, E));
    if (infix == 0) return PLAN_BAD_OVERLAP;
    memset(out, 0, textLen * (valueBits / 8));
#define GM_EMU_ARGS bf, br, rows, nseqTotal, text, textLen, seqCum, nseqLocal, K, E, infix, revcompl, valueBits, intervals, nIntervals, out, stats, sa, verifyT, allCodes, allCum, jumpCap
    switch (wpp) {
        case 1: return nless ? run<1, true>(GM_EMU_ARGS) : run<1, false>(GM_EMU_ARGS);
        case 3: return nless ? run<3, true>(GM_EMU_ARGS) : run<3, false>(GM_EMU_ARGS);
        case 9: return nless ? run<9, true>(GM_EMU_ARGS) : run<9, false>(GM_EMU_ARGS);
    }
#undef GM_EMU_ARGS
    return -100;
}

extern "C" int gm_emu_map(int wpp, const uint8_t* bf, const uint8_t* br, uint64_t rows, uint32_t nseqTotal, const uint8_t* text,
                          uint64_t textLen, const uint64_t* seqCum, uint32_t nseqLocal, uint32_t K, uint32_t E, int32_t xo,
                          int32_t infixOverride, int revcompl, int valueBits, const uint64_t* intervals, uint64_t nIntervals,
                          void* out, uint64_t* stats, const uint32_t* sa, uint32_t verifyT, const uint8_t* allCodes, const uint64_t* allCum)
{
    uint64_t st[7] = {0, 0, 0, 0, 0, 0, 0};
    const int rc = gm_emu_map2(wpp, bf, br, rows, nseqTotal, text, textLen, seqCum, nseqLocal, K, E, xo, infixOverride, revcompl, valueBits, intervals, nIntervals,
                               out, st, sa, verifyT, allCodes, allCum, 0, 0);
    if (stats) for (int i = 0; i < 4; ++i) stats[i] = st[i];
    return rc;
}
