// gm_expand.h -- phase A of the split search: the jump patterns of a root enumerated by lanes of their own (host/device header).
//
// Round 6.  Until round 5 ONE persistent loop did two jobs per lane: it turned the jump patterns of the lane's root into search nodes
// (item -> bitmap word -> table entries -> neighbour filter: gm_kernels.h, parts A / B of search_body) and it walked the subtrees below
// those nodes.  The two jobs want different things -- the first is a stream of independent random reads with almost no state, the second a
// state machine with stacks and needle windows in LDS -- and sharing a loop cost both: half of the lanes held no node in an iteration
// because they were waiting for a word or an entry, and every iteration walked through ~190 VALU instructions of pattern code.
//
// The split: phase A (expand_kernel, gm_kernels.h) gives every (root, item) pair a lane of its own.  The lane reads its item's bitmap
// word, looks the surviving substituted J-mers up in the table of all J-mers, applies the neighbour filters and appends what is left to a
// list of self-contained NODE PACKETS in HBM (the node, its root and the root's needle window).  Phase B (search_kernel with Env::NODES)
// is the persistent walker of before, but it draws PACKETS instead of roots and knows nothing about patterns.
//
// Everything a lane decides in phase A is in this header so that the CPU harness (tests/emu) runs the very same code against the
// oracle before a GPU does.  Semantics kept: /root/reference/src/find2_index_approx.hpp:223-369 (which substitutions a search allows in its
// first J characters: gm_oss.h), /root/reference/src/algo.hpp:165-218 (what happens below the node: unchanged, phase B).
#pragma once
#include "gm_engine.h"

namespace gm {

constexpr uint32_t NB_SYMS = 6;       // neighbour symbols per side carried by one-row q-mer table entries (qmer_table_kernel)
constexpr uint32_t NB_SYMS2 = 3;      // ... and per side and row by two-row entries

// A node packet: (2 + pktChunks) x 16 bytes.
//   unit 0: the node {fwd lo, rev lo, width, meta}                                   (NodeT<uint32_t>)
//   unit 1: {window origin (slice position of the block's first k-mer), n | strand << 8 | search << 9, chunk of phase A that wrote it, stamp}
//   units 2..: the root's needle window, 4 bits per symbol, symbol 0 of the window in nibble 0 (no offset: the walker stages it as it is)
// `stamp` numbers the slice (a packet slot that phase A did not fill in this slice holds an older stamp or zero), `chunk` lets the walker
// drop the packets of work that did not fit and is redone by the next slice.
constexpr uint32_t PKT_HEADER_UNITS = 2;
GM_HD uint32_t pkt_units(uint32_t pktChunks) { return PKT_HEADER_UNITS + pktChunks; }
GM_HD uint32_t pkt_chunks_for(uint32_t K, uint32_t stepSize) { return (K + stepSize - 1u + 31u) / 32u; }
GM_HD uint32_t pkt_root_word(uint32_t n, uint32_t strand, uint32_t search) { return n | strand << 8 | search << 9; }

// work item q of a k-mer block -> {search | strand << 3 | item number among the call's items << 8}
GM_HD uint32_t wmap_pack(uint32_t search, uint32_t strand, uint32_t jp) { return search | strand << 3 | jp << 8; }
constexpr uint32_t WMAP_ROOT_ONLY = 16u;   // first pass of two: the search has no pattern without a substitution (a lower bound inside its first J characters): its work
                                           // item only speaks for the roots that walk the tree from its root

// 16 symbols of the 4-bit text starting at symbol p (any alignment).  Mem::pair(i, lo, hi): 64-bit words i and i + 1 of the packed text.
template <class Mem> GM_HD uint64_t nib64(const Mem& mem, uint64_t p)
{
    uint64_t lo, hi;
    mem.pair(p >> 4, lo, hi);
    const uint32_t r = (uint32_t)(p & 15u) * 4u;
    return r ? (lo >> r) | (hi << (64u - r)) : lo;
}

// what a root contributes to every one of its items
struct XRoot {
    uint32_t jb;    // index of the needle's J-mer in the table of all J-mers
    uint32_t jn;    // the needle's characters next to the J-mer, packed like the 4th word of a one-row table entry (bit 15: the filter applies)
    uint32_t ext;   // JF_EXTOK | letters << JF_EXT_SHIFT: the two letters behind the J-mer (groups of kind 1), 0 when one of them is N
    uint32_t bad;   // an N inside the J-mer: the root walks the tree from its root
};

// g: text symbol of the window's first symbol (slice begin + window origin); W = K + n - 1; a0: window coordinate of the J-mer's first
// character (n - 1 + regionA); nbWord: 4th word of the search's jump record; wantExt: the search has groups of kind 1
// (the arithmetic of search_body's stage 2, with the window read from the packed text instead of LDS)
template <class Mem>
GM_HD XRoot expand_root(const Mem& mem, uint64_t g, uint32_t W, uint32_t strand, uint32_t a0, uint32_t J, uint32_t nbWord, bool wantExt)
{
    XRoot r; r.jn = 0u; r.ext = 0u;
    const uint32_t ni = strand ? W - a0 - J : a0;     // window position of the J-mer's lowest text symbol
    const uint64_t v = nib64(mem, g + ni);
    const uint64_t qmask = J >= 16u ? ~0ull : ((1ull << (4u * J)) - 1ull);
    r.bad = (v & qmask & 0xCCCCCCCCCCCCCCCCull) != 0ull ? 1u : 0u;
    uint64_t t = v & qmask & 0x3333333333333333ull;
    t = (t | (t >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    t = (t | (t >> 4)) & 0x00FF00FF00FF00FFull;
    t = (t | (t >> 8)) & 0x0000FFFF0000FFFFull;
    t = (t | (t >> 16)) & 0x00000000FFFFFFFFull;
    const uint32_t lo2 = (uint32_t)t;                                   // sum of c_i << 2i, i in text order
    const uint32_t m2 = J >= 16u ? 0xFFFFFFFFu : ((1u << (2u * J)) - 1u);
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t rv = __builtin_bitreverse32(lo2);
#else
    uint32_t rv = lo2;
    rv = ((rv >> 1) & 0x55555555u) | ((rv & 0x55555555u) << 1);
    rv = ((rv >> 2) & 0x33333333u) | ((rv & 0x33333333u) << 2);
    rv = ((rv >> 4) & 0x0F0F0F0Fu) | ((rv & 0x0F0F0F0Fu) << 4);
    rv = __builtin_bswap32(rv);
#endif
    rv = ((rv & 0xAAAAAAAAu) >> 1) | ((rv & 0x55555555u) << 1);         // the 2-bit groups in reverse order
    r.jb = strand ? (~lo2 & m2) : (J ? rv >> (32u - 2u * J) : 0u);
    if (r.bad) return r;
    const bool filter = (nbWord >> 31) != 0u;
    if (!filter && !wantExt) return r;
    // the symbols next to the J-mer in the window: `up` = the 16 behind its highest text symbol, `low` = the (up to) 6 in front of its lowest
    const uint64_t up = nib64(mem, g + ni + J);
    const uint32_t d = ni < NB_SYMS ? ni : NB_SYMS;
    const uint64_t low = nib64(mem, g + ni - d);      // symbol ni - 1 - i sits in nibble d - 1 - i
    // needle(a0 + J + i): forward strand the i-th symbol above, reverse strand the complement of the i-th symbol below; needle(a0 - 1 - i) the
    // other way round.  Six symbols per side at once (no loop over the neighbours: phase A is paid per (root, item)):
    //   ab: nibble i = the i-th symbol above; be: nibble i = the i-th symbol below (N where the window has none)
    const uint32_t ab = (uint32_t)up & 0xFFFFFFu;
    const uint32_t be = (d ? (uint32_t)(nib_reverse16(low) >> (4u * (16u - d))) : 0u) | (0x444444u << (4u * d) & 0xFFFFFFu);
    const uint32_t rt6 = strand ? (uint32_t)nib_complement(be) & 0xFFFFFFu : ab, lf6 = strand ? (uint32_t)nib_complement(ab) & 0xFFFFFFu : be;
    auto pack2 = [](uint32_t x) {   // six nibbles (letters) -> six 2-bit groups
        x &= 0x333333u; x = (x | x >> 2) & 0x0F0F0Fu;
        return (x & 0xFu) | ((x >> 8) & 0xFu) << 4 | ((x >> 16) & 0xFu) << 8;
    };
    if (filter) {
        const uint32_t nr = (nbWord >> 12) & 7u, nl = (nbWord >> 28) & 7u;
        const uint32_t mr = (1u << (4u * nr)) - 1u, ml = (1u << (4u * nl)) - 1u;
        const uint32_t notLetter = ((rt6 & mr) | (lf6 & ml)) & 0x444444u;
        r.jn = notLetter ? 0u : (pack2(rt6 & mr) | pack2(lf6 & ml) << 16 | 0x8000u);   // a needle N mismatches everything: such roots take no shortcut
    }
    if (wantExt) {
        const uint32_t e0 = rt6 & 15u, e1 = (rt6 >> 4) & 15u;
        if ((e0 | e1) < SYM_N) r.ext = JF_EXTOK | (e0 << 2 | e1) << JF_EXT_SHIFT;
    }
    return r;
}

// one item of a search in the hands of its lane
struct XItem {
    unsigned long long alive;   // rotations of the group that exist and are still to be looked up
    uint32_t gcur, sh;          // the group's own rotations, the bit offset of its three characters
    uint32_t state;             // 0 nothing (left), 1 a plain pattern to look up, 2 the group's word is wanted, 3 rotations of `alive`
    uint32_t widx, wsel;        // state 2: word `widx` of bitmap plane `wsel`
};
// jd: the item; flags: jump_item_flags of its number (gm_oss.h).  Tab as for jump_decide.
template <class Tab> GM_HD XItem expand_item(uint32_t jd, uint32_t flags, const XRoot& r, const Tab& tab)
{
    XItem it; it.alive = 0ull; it.gcur = jd; it.sh = 0u; it.state = 1u; it.widx = it.wsel = 0u;
    if (!(flags & JF_GROUP)) return it;
    const uint32_t ly = tab.layout((flags >> JF_IL_SHIFT) & 7u);
    const uint32_t sh = ly & 31u, kind = (jd >> (sh + 3u)) & 1u;
    it.sh = sh; it.gcur = jd & ~(63u << sh);
    if (kind && !(r.ext & JF_EXTOK)) { it.state = 0u; return it; }   // a needle N behind the J-mer: no pattern without budget can match
    it.widx = group_word(rot_add(r.jb, it.gcur), sh);
    it.wsel = kind ? ((ly >> 13) & 255u) + ((r.ext >> JF_EXT_SHIFT) & 15u) : (ly >> 5) & 255u;
    it.state = 2u;
    return it;
}
template <class Tab> GM_HD void expand_word(XItem& it, unsigned long long pw, uint32_t jd, const XRoot& r, const Tab& tab)
{
    it.alive = word_to_rotations(pw, (r.jb >> it.sh) & 63u) & tab.mask((jd >> it.sh) & 7u);
    it.state = 3u;
}
// the next rotation word of the item, if any
GM_HD bool expand_next(XItem& it, uint32_t& rw)
{
    if (it.state == 1u) { rw = it.gcur; it.state = 0u; return true; }
    if (it.state != 3u || it.alive == 0ull) return false;
    rw = it.gcur | ctz64(it.alive) << it.sh;
    it.alive &= it.alive - 1ull;
    return true;
}

// Two passes (gm_api.hip: run_expand): the patterns WITHOUT a substitution of every root first -- a work item per root, the J-mer's own
// table entry -- and their packets walked; then everything else, for the blocks whose k-mers are not all at MAX already.  On a genome the
// k-mers inside young repeat families have thousands of exact copies: what the first pass counts for them ends the second pass before
// it reads a single bitmap word.  This takes the pattern without a substitution out of an item of the second pass.
GM_HD void expand_strip_exact(XItem& it)
{
    if (it.gcur != 0u) return;                       // some character outside the group is substituted
    if (it.state == 1u) it.state = 0u;               // the plain pattern without a substitution
    else if (it.state == 3u) it.alive &= ~1ull;      // rotation 0 of the group's own three characters
}

// position of the k-th set bit of m (k = 0: the lowest; k < popcount(m)): bisection, no loop over the bits
GM_HD uint32_t nth_set_bit(unsigned long long m, uint32_t k)
{
    const uint32_t lo = (uint32_t)m, cl = popc32(lo);
    uint32_t x = lo, pos = 0;
    if (k >= cl) { k -= cl; x = (uint32_t)(m >> 32); pos = 32u; }
    uint32_t off = 0;
    for (uint32_t w = 16u; w >= 1u; w >>= 1) {
        const uint32_t c = popc32((x >> off) & ((1u << w) - 1u));
        if (k >= c) { k -= c; off += w; }
    }
    return pos + off;
}
// rotation word number k of an item (expand_kernel deals the rotations of 64 items out to the lanes of the wavefront: lane j takes the j-th)
GM_HD uint32_t expand_count(const XItem& it) { return it.state == 1u ? 1u : it.state == 3u ?
#if defined(__HIP_DEVICE_COMPILE__)
    (uint32_t)__popcll(it.alive)
#else
    (uint32_t)__builtin_popcountll(it.alive)
#endif
    : 0u; }
GM_HD uint32_t expand_nth(uint32_t gcur, uint32_t sh, bool plain, unsigned long long alive, uint32_t k) { return plain ? gcur : gcur | nth_set_bit(alive, k) << sh; }

// A table entry {fwd lo, rev lo, width, neighbour word} of the J-mer with rotation word rw applied -> a node at depth J, or nothing.
// jm0: meta of the node at depth J without errors; h: 4th word of the search's jump record (which neighbours count).
// (search_body part A of round 5, line by line: the one-row and the two-row neighbour filter)
struct XNode { uint32_t take, flo, rlo, w, meta, errs; };
GM_HD XNode expand_filter(uint32_t eFlo, uint32_t eRlo, uint32_t eW, uint32_t eNb, uint32_t rw, uint32_t jm0, uint32_t jn, uint32_t h, uint32_t E, uint32_t nbFilter, uint32_t verifyT)
{
    XNode x; x.errs = rot_errors(rw); x.flo = eFlo; x.rlo = eRlo; x.w = eW; x.meta = jm0 | x.errs << META_ERRS_SHIFT;
    bool take = eW != 0u;
    if (take && eW == 1u && (jn & eNb & 0x8000u) != 0u) {
        // The substituted J-mer occurs once.  Whatever this node could still find lies at that one place and contains the whole infix, so the
        // infix characters next to the J-mer must agree with the text there up to the errors the pattern has left; a text N or a sequence
        // end within them ends it too (N-less pass).
        const uint32_t df = eNb ^ jn;
        const uint32_t differ = (df | df >> 1) & h & 0x05550555u;
        take = ((eNb >> 12) & 7u) >= ((h >> 12) & 7u) && ((eNb >> 28) & 7u) >= ((h >> 28) & 7u) && x.errs + popc32(differ) <= E;
    }
    if (take && eW == 2u && (jn & 0x8000u) != 0u && nbFilter == 1u) {
        // The substituted J-mer occurs TWICE: the same test with 3 + 3 neighbours for either row.  Neither passes: no node.  One passes: the
        // node is that row alone, and a lone row of which only the forward position is known is never stepped -- it goes to the
        // verification queue (rows-only node, rlo = all ones).
        const uint32_t needR = ((h >> 12) & 7u) < NB_SYMS2 ? ((h >> 12) & 7u) : NB_SYMS2, needL = ((h >> 28) & 7u) < NB_SYMS2 ? ((h >> 28) & 7u) : NB_SYMS2;
        const uint32_t mR = h & 0x15u, mL = (h >> 16) & 0x15u, budget = E - x.errs;
        bool pass[2];
        for (uint32_t r = 0; r < 2u; ++r) {
            const uint32_t y = (eNb >> (16u * r)) & 0xFFFFu;
            const uint32_t dr = (y ^ jn) & 0x3Fu, dl = ((y >> 6) ^ (jn >> 16)) & 0x3Fu;
            const uint32_t mism = popc32((dr | dr >> 1) & mR) + popc32((dl | dl >> 1) & mL);
            pass[r] = ((y >> 12) & 3u) >= needR && (y >> 14) >= needL && mism <= budget;
        }
        take = pass[0] || pass[1];
        if (take && pass[0] != pass[1] && verifyT != 0u) { x.flo += pass[1] ? 1u : 0u; x.rlo = ~0u; x.w = 1u; }
    }
    x.take = take ? 1u : 0u;
    return x;
}

// which of the walker's three lists a node goes to: the walker draws the nodes of patterns without a substitution first, then one, then the
// rest -- the counters that prune the later ones (min(total, MAX): a block whose k-mers are all at MAX needs no further hit) are mostly
// final by the time they are drawn
GM_HD uint32_t expand_class(uint32_t errs) { return errs < 2u ? errs : 2u; }

// (host) the work items of one k-mer block: every (strand, search, item) of the regular block shape, strand-major.
// firstItem[s] / nItems[s]: where the items of search s sit among the call's items
inline std::vector<uint32_t> make_wmap(uint32_t nStrands, uint32_t nSearches, const uint32_t* firstItem, const uint32_t* nItems)
{
    std::vector<uint32_t> m;
    for (uint32_t st = 0; st < nStrands; ++st)
        for (uint32_t s = 0; s < nSearches; ++s)
            for (uint32_t j = 0; j < nItems[s]; ++j) m.push_back(wmap_pack(s, st, firstItem[s] + j));
    return m;
}

}  // namespace gm
