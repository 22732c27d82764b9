// gm_oss.h -- Optimum Search Schemes (Hamming) as data, plus the per-block-shape length table.
//
// Scheme constants are those of /root/reference/src/find2_index_approx.hpp:67-134 (pi = block order,
// l/u = cumulative lower/upper error bounds).  Block lengths follow
// _optimalSearchSchemeComputeFixedBlocklengthGM / SetBlockLengthGM / InitGM (:139-176).
// The reference recomputes lengths per block on the CPU (std::vector, src/algo.hpp:248-249); here the
// host tabulates one 16-byte record per (k-mers-in-block n, search s) and a lane keeps the record of
// its current root in four VGPRs.
#pragma once
#include "gm_common.h"
#include <vector>

namespace gm {

constexpr int OSS_MAXB = 6;
constexpr int OSS_MAXS = 7;   // table stride per n is 8 records

struct OssSearch { uint8_t nb; uint8_t pi[OSS_MAXB], l[OSS_MAXB], u[OSS_MAXB]; };
struct OssScheme { uint8_t ns; OssSearch s[OSS_MAXS]; };

inline const OssScheme& oss_scheme(uint32_t E)   // host only: lanes read packed OssRecords
{
    static const OssScheme S[5] = {
        {1, {{1, {1}, {0}, {0}}}},
        {2, {{2, {1, 2}, {0, 0}, {0, 1}},
             {2, {2, 1}, {0, 1}, {0, 1}}}},
        {3, {{4, {1, 2, 3, 4}, {0, 0, 1, 1}, {0, 0, 2, 2}},
             {4, {3, 2, 1, 4}, {0, 0, 0, 0}, {0, 1, 1, 2}},
             {4, {4, 3, 2, 1}, {0, 0, 0, 2}, {0, 1, 2, 2}}}},
        {4, {{5, {1, 2, 3, 4, 5}, {0, 0, 0, 0, 3}, {0, 1, 2, 3, 3}},
             {5, {2, 3, 4, 5, 1}, {0, 0, 0, 2, 2}, {0, 1, 2, 2, 3}},
             {5, {3, 4, 5, 2, 1}, {0, 0, 1, 1, 1}, {0, 1, 1, 3, 3}},
             {5, {5, 4, 3, 2, 1}, {0, 0, 0, 0, 0}, {0, 0, 3, 3, 3}}}},
        {7, {{6, {1, 2, 3, 4, 5, 6}, {0, 0, 0, 0, 0, 4}, {0, 2, 3, 3, 4, 4}},
             {6, {3, 4, 5, 6, 2, 1}, {0, 0, 0, 1, 4, 4}, {0, 0, 1, 1, 4, 4}},
             {6, {2, 3, 4, 5, 6, 1}, {0, 0, 0, 0, 0, 0}, {0, 2, 2, 3, 3, 4}},
             {6, {3, 2, 4, 5, 6, 1}, {0, 1, 1, 1, 1, 1}, {0, 1, 2, 3, 3, 4}},
             {6, {4, 3, 2, 5, 6, 1}, {0, 0, 2, 2, 2, 2}, {0, 0, 2, 3, 3, 4}},
             {6, {4, 3, 2, 5, 6, 1}, {0, 1, 2, 2, 2, 2}, {0, 1, 2, 3, 3, 4}},
             {6, {6, 5, 4, 3, 2, 1}, {0, 0, 0, 0, 3, 3}, {0, 0, 4, 4, 4, 4}}}},
    };
    return S[E];
}

// One search of one block shape, packed:
//   x: cumulative block lengths bl[0..3] (8 bit each, search order)
//   y: bl[4] | bl[5]<<8 | startPos<<16 | nb<<24
//   z: l[0..5] 3 bit each (bits 0..17) | goRight[0..5] (bits 18..23; block bi is searched left-to-right)
//   w: u[0..5] 3 bit each
struct OssRecord { uint32_t x, y, z, w; };

GM_HD uint32_t oss_bl(const OssRecord& r, uint32_t bi)
{
    uint32_t lo = bi < 4 ? r.x : r.y;
    return (lo >> (8u * (bi & 3u))) & 0xFFu;
}
GM_HD uint32_t oss_start(const OssRecord& r) { return (r.y >> 16) & 0xFFu; }
GM_HD uint32_t oss_nb(const OssRecord& r) { return r.y >> 24; }
GM_HD uint32_t oss_l(const OssRecord& r, uint32_t bi) { return (r.z >> (3u * bi)) & 7u; }
GM_HD uint32_t oss_u(const OssRecord& r, uint32_t bi) { return (r.w >> (3u * bi)) & 7u; }
GM_HD uint32_t oss_right(const OssRecord& r, uint32_t bi) { return (r.z >> (18u + bi)) & 1u; }

// infixLen = length of the common infix of the block's k-mers (the reference's local `overlap`,
// src/algo.hpp:246).  Returns false if the infix is shorter than the number of scheme blocks.
// blockLens (optional): lengths of blocks 1..nb in left-to-right order, summing to infixLen.  The scheme covers every
// error distribution exactly once for ANY positive block lengths (l/u constrain errors per block, not per position);
// the reference always uses equal lengths (:167-172).
// layout of search s over an infix of infixLen characters: cumulative block lengths in search order, start position, packed bounds
inline bool oss_layout(uint32_t E, uint32_t s, uint32_t infixLen, const uint32_t* blockLens, uint32_t bl[OSS_MAXB], uint32_t* startOut, uint32_t* zOut, uint32_t* wOut, uint32_t* blocksOut)
{
    const OssSearch& S = oss_scheme(E).s[s];
    uint32_t blocks = S.nb;
    if (infixLen < blocks) return false;
    uint32_t base = infixLen / blocks, rest = infixLen - blocks * base;   // :167-172
    uint32_t cum = 0, start = 0;
    for (uint32_t i = 0; i < (uint32_t)OSS_MAXB; ++i) bl[i] = 0;
    for (uint32_t i = 0; i < blocks; ++i) {
        uint32_t len = blockLens ? blockLens[S.pi[i] - 1] : base + ((uint32_t)(S.pi[i] - 1) < rest ? 1u : 0u);   // :145 blocklength[pi[i]-1]
        cum += len;
        bl[i] = cum;
        if (S.pi[i] < S.pi[0]) start += len;                               // :158-160
    }
    uint32_t z = 0, w = 0;
    for (uint32_t i = 0; i < blocks; ++i) {
        z |= (uint32_t)S.l[i] << (3u * i);
        w |= (uint32_t)S.u[i] << (3u * i);
        // direction of block i: the first block always goes right (:441); later ones by pi order (:274,:321)
        uint32_t right = (i == 0) ? 1u : (S.pi[i] > S.pi[i - 1] ? 1u : 0u);
        z |= right << (18u + i);
    }
    *startOut = start; *zOut = z; *wOut = w; *blocksOut = blocks;
    return true;
}

inline bool oss_make_record(uint32_t E, uint32_t s, uint32_t infixLen, OssRecord* out, const uint32_t* blockLens = nullptr)
{
    uint32_t bl[OSS_MAXB], start, z, w, blocks;
    if (infixLen > 255u || !oss_layout(E, s, infixLen, blockLens, bl, &start, &z, &w, &blocks)) return false;
    OssRecord r;
    r.x = bl[0] | bl[1] << 8 | bl[2] << 16 | bl[3] << 24;
    r.y = bl[4] | bl[5] << 8 | start << 16 | blocks << 24;
    r.z = z; r.w = w;
    *out = r;
    return true;
}

// The same search for k-mers longer than MAX_K (gm_longk.h): 16-bit lengths, 24 bytes, read from memory by the lane (not kept in registers).
//   z, w as in OssRecord (l / goRight, u)
struct OssRecordL { uint16_t bl[OSS_MAXB]; uint16_t start, nb; uint32_t z, w; };
GM_HD uint32_t oss_bl(const OssRecordL& r, uint32_t bi) { return r.bl[bi]; }
GM_HD uint32_t oss_start(const OssRecordL& r) { return r.start; }
GM_HD uint32_t oss_nb(const OssRecordL& r) { return r.nb; }
GM_HD uint32_t oss_l(const OssRecordL& r, uint32_t bi) { return (r.z >> (3u * bi)) & 7u; }
GM_HD uint32_t oss_u(const OssRecordL& r, uint32_t bi) { return (r.w >> (3u * bi)) & 7u; }
GM_HD uint32_t oss_right(const OssRecordL& r, uint32_t bi) { return (r.z >> (18u + bi)) & 1u; }
inline bool oss_make_record_long(uint32_t E, uint32_t s, uint32_t infixLen, OssRecordL* out, const uint32_t* blockLens = nullptr)
{
    uint32_t bl[OSS_MAXB], start, z, w, blocks;
    if (infixLen > 0xFFFFu || !oss_layout(E, s, infixLen, blockLens, bl, &start, &z, &w, &blocks)) return false;
    OssRecordL r;
    for (int i = 0; i < OSS_MAXB; ++i) r.bl[i] = (uint16_t)bl[i];
    r.start = (uint16_t)start; r.nb = (uint16_t)blocks; r.z = z; r.w = w;
    *out = r;
    return true;
}

// ---- jump patterns: the top of the search tree as direct table lookups ------------------------------------------------------
// The first J characters of a search (in search order) are a contiguous substring of the needle window.  Every path of
// _optimalSearchSchemeChildrenGM / ...ExactGM (find2_index_approx.hpp:223-369) through these J characters is "the needle
// substring with some characters substituted", and which substitution sets are allowed depends only on the scheme's cumulative
// bounds l/u -- not on the index.  So instead of walking the top of the tree node by node (at e = 2 on a 3 Gbp text that is
// ~70 % of all steps: every string of <= 14 characters exists), the lane enumerates the PATTERNS of its search and reads the
// SA ranges of the substituted J-mers from the table of all J-mers (one 16-byte read each); the tree walk starts at depth J.
// Only substitutions by A,C,G,T are enumerated (the table holds no N): kernels that jump run with Env::NLESS (gm_engine.h).
//
// descriptor (u32): bits 0..2 number of substitutions (<= 4); substitution k at bits 3 + 6k: offset of the character inside the
// J-mer in needle order (4 bits), rotation r in 1..3 (2 bits): the substituted letter is (needle letter + r) & 3.
struct JumpSearch {
    uint32_t J = 0;        // characters covered (0: no jump for this search)
    uint32_t regionA = 0;  // window coordinate of the J-mer's first character, minus (n - 1)   [the infix starts at n - 1]
    uint32_t meta0 = 0;    // (a - (n-1)) | (bx - (n-1)) << 9 | t << 18 of the node at depth J (errs and mode are added by the lane)
    std::vector<uint32_t> pat;
};

// (host) Patterns of search s for an infix of length L split by rec; J characters.  Returns false if more than maxPatterns exist.
inline bool oss_jump_patterns(uint32_t E, const OssRecord& rec, uint32_t L, uint32_t J, size_t maxPatterns, JumpSearch* out)
{
    out->J = 0; out->pat.clear();
    if (J == 0 || J >= L || J > 16u) return false;   // (offsets inside the J-mer have 4 bits)
    // the exact path through the first J characters: position, block and "block ends here" of every character
    uint32_t a = oss_start(rec), bx = a, t = 0;    // coordinates relative to the infix start
    uint32_t pos[16], blk[16]; bool endsBlock[16];
    for (uint32_t i = 0; i < J; ++i) {
        const bool right = oss_right(rec, t) != 0;
        pos[i] = right ? bx : a - 1u;
        blk[i] = t;
        if (right) ++bx; else --a;
        endsBlock[i] = (bx - a == oss_bl(rec, t));
        if (endsBlock[i]) ++t;
    }
    out->regionA = a; out->meta0 = a | bx << 9 | t << 18; out->J = J;
    // depth-first over the characters: match, or substitute while the block's upper bound allows; lower bound at block ends
    struct Frame { uint32_t i, errs, desc; };
    std::vector<Frame> st; st.push_back({0, 0, 0});
    while (!st.empty()) {
        const Frame f = st.back(); st.pop_back();
        if (f.i == J) { out->pat.push_back(f.desc | f.errs); if (out->pat.size() > maxPatterns) { out->J = 0; out->pat.clear(); return false; } continue; }
        const uint32_t b = blk[f.i], u = oss_u(rec, b), l = oss_l(rec, b);
        // pushed in reverse so that patterns come out in the order of the tree walk (match first)
        if (f.errs + 1u <= u && f.errs < 4u && !(endsBlock[f.i] && f.errs + 1u < l))
            for (uint32_t r = 3; r >= 1; --r) st.push_back({f.i + 1u, f.errs + 1u, f.desc | ((pos[f.i] - a) | r << 4) << (3u + 6u * f.errs)});
        if (!(endsBlock[f.i] && f.errs < l)) st.push_back({f.i + 1u, f.errs, f.desc});
    }
    return true;
}

// ---- items of a search: rotation words, and groups of patterns answered by ONE word of an existence bitmap --------------------
// What a lane reads is not the descriptor above but its ROTATION WORD: 2-bit field at index bits 2(J-1-off) = r, i.e. the rotations laid
// out like the table index itself, so that "apply the substitutions" is one carry-less field-wise addition (rot_add) and "count them"
// one population count (rot_errors) -- no loop over the substitutions in the kernel.
//
// On a genome about half of the substituted J-mers of a search do not occur at all (3 Gbp, J = 16: the average 16-mer has 0.72
// occurrences), and most of those that do occur are chance hits which the neighbour filter drops right after their 16-byte table entry
// has been read -- an HBM fetch and a turn of the lane each.  Bitmaps answer for 64 patterns at once, with ONE 8-byte read:
//   kind 0  "does the J-mer occur"                                              4^J bits      (from the table of all J-mers)
//   kind 1  "does the J-mer occur FOLLOWED BY the needle's next two letters"    16 x 4^J bits (from the text; 3 Gbp: 4 % of the 18-mers occur)
// Kind 1 is a necessary condition for every pattern that has spent the whole error budget (the rest of the infix must follow
// exactly) when two more infix characters lie to the right of the J-mer.  64 patterns that differ only in three adjacent characters form
// a GROUP; two layouts: LOW = the last three characters of the J-mer (index bits 0..5), MID = the three before them (bits 6..11; its
// bitmap is indexed with those two 6-bit fields swapped, jump_swap_mid).
//   group item (u32): the rotations of all other characters | in the 6 bits of the group's own characters: mask id (bits 0..2), kind (bit 3)
//   mask (u64): bit (r0 << 4 | r1 << 2 | r2) set = the pattern that additionally rotates the group's three characters by r0, r1, r2
//               belongs to the search.
// The lane reads the group's word, brings it into "rotation space" (word_to_rotations) and walks the set bits of word & mask: only those
// patterns are looked up in the table.  A search's items are its LOW groups, then its MID groups, then its plain patterns.
// Round 5: a group may sit at ANY three adjacent characters ("layout" = the bit offset `sh` of their 6 bits inside the J-mer index): the
// one-substitution patterns of the searches that start on the RIGHT of the infix (e = 2: searches 2 and 3) touch the first eight
// characters of their J-mer, which neither LOW (sh = 0) nor MID (sh = 6) covers -- they were 50 of the 62 plain table reads of a
// K = 30 e = 2 block.  Word index and bit of a J-mer in the bitmap of layout sh: group_word / (idx >> sh) & 63 (sh = 0 and 6 are the LOW and
// MID of round 4, bit for bit).  Kind 1 exists for LOW and MID only (16 bitmaps each); the other layouts have a kind-0 bitmap of their own.
// A call uses at most GROUP_MAX_LAYOUTS layouts; a search's items come layout by layout (segments), then its plain patterns.
constexpr uint32_t GROUP_SYMS = 3;
constexpr uint32_t GROUP_MAX_MASKS = 8;
constexpr uint32_t GROUP_MAX_LAYOUTS = 6;
// the layouts a call with jumps of J characters may use, in the order of its segments: LOW, MID, then from the left end of the J-mer
inline std::vector<uint32_t> group_layout_shifts(uint32_t J)
{
    std::vector<uint32_t> sh;
    if (J <= GROUP_SYMS) return sh;
    sh.push_back(0u);
    if (J >= 2u * GROUP_SYMS) sh.push_back(6u);
    for (int32_t v = 2 * (int32_t)J - 6; v > 6 && sh.size() < GROUP_MAX_LAYOUTS; v -= 6) sh.push_back((uint32_t)v);
    return sh;
}
inline uint32_t rot_errors_host(uint32_t rw) { return (uint32_t)__builtin_popcount((rw | rw >> 1) & 0x55555555u); }

struct SearchItems {
    std::vector<uint32_t> items;   // the groups of layout 0, of layout 1, ..., then the plain rotation words
    std::vector<uint32_t> seg;     // groups per layout (same order as group_layout_shifts)
    uint32_t groups() const { uint32_t g = 0; for (uint32_t v : seg) g += v; return g; }
    uint32_t patterns = 0;         // patterns covered (== JumpSearch::pat.size())
    bool ext = false;              // some group is of kind 1: the lane needs the two letters behind the J-mer
};

inline uint32_t oss_rotation_word(uint32_t d, uint32_t J)
{
    uint32_t rw = 0;
    for (uint32_t k = 0; k < (d & 7u); ++k) { const uint32_t f = (d >> (3u + 6u * k)) & 63u; rw |= (f >> 4) << (2u * (J - 1u - (f & 15u))); }
    return rw;
}

// (host) Items of one search of a call with E errors.
//   group:  0 plain patterns only, 1 groups wherever two patterns share a word, 2 groups where they are expected to save table reads
//           (occur0 / occur1 = share of the J-mers / of the (J+2)-mers that occur in the text)
//   ext:    kind 1 bitmaps are at hand and two more infix characters lie to the right of the J-mer
// masks: the distinct masks of the call so far (shared by its searches).  A group whose mask would be the ninth is not formed.
inline void oss_make_items(const JumpSearch& js, uint32_t E, int group, bool ext, double occur0, double occur1, std::vector<uint64_t>* masks, SearchItems* out,
                           uint32_t kind0Layouts = 1u)   // kind0Layouts: bit L set = layout L has a kind-0 bitmap at hand (LOW always has)
{
    const std::vector<uint32_t> shifts = group_layout_shifts(js.J);
    out->items.clear(); out->patterns = (uint32_t)js.pat.size(); out->seg.assign(shifts.size(), 0u); out->ext = false;
    std::vector<uint32_t> rws;
    for (uint32_t d : js.pat) rws.push_back(oss_rotation_word(d, js.J));
    if (!group || shifts.empty() || !masks) { out->items = rws; return; }
    struct G { uint32_t key; uint64_t mask; uint32_t kind; };
    std::vector<uint64_t> m2 = *masks;
    bool fail = false;
    std::vector<uint32_t> rest = rws, grouped;
    for (size_t L = 0; L < shifts.size(); ++L) {
        const uint32_t shift = shifts[L];
        const bool k0 = ((kind0Layouts >> L) & 1u) != 0u, k1 = ext && L < 2u;   // kind 1 (two more letters) exists for LOW and MID
        if (!k0 && !k1) continue;
        // members that may share a word: same rotations outside the layout's three characters, same kind
        std::vector<G> gs;
        for (uint32_t rw : rest) {
            const uint32_t key = rw & ~(63u << shift), kind = (k1 && rot_errors_host(rw) == E) ? 1u : 0u;
            size_t g = 0;
            while (g < gs.size() && !(gs[g].key == key && gs[g].kind == kind)) ++g;
            if (g == gs.size()) gs.push_back(G{key, 0, kind});
            gs[g].mask |= 1ull << ((rw >> shift) & 63u);
        }
        std::vector<uint32_t> left;
        for (const G& g : gs) {
            const int m = __builtin_popcountll(g.mask);
            bool take = m >= 2 && (g.kind ? k1 : k0);
            // one word + the members that pass it, against one table read per member
            if (take && group != 1) take = m * (1.0 - (g.kind ? occur1 : occur0)) >= 1.5;
            if (take) {
                size_t id = 0;
                while (id < m2.size() && m2[id] != g.mask) ++id;
                if (id == m2.size()) { if (m2.size() == GROUP_MAX_MASKS) take = false; else m2.push_back(g.mask); }   // (out of masks: its members stay single)
                if (take) { grouped.push_back(g.key | ((uint32_t)id | g.kind << 3) << shift); out->seg[L]++; out->ext = out->ext || g.kind != 0u; }
            }
            if (!take) for (uint32_t r = 0; r < 64u; ++r) if ((g.mask >> r) & 1ull) left.push_back(g.key | r << shift);
        }
        rest = left;
    }
    out->items = grouped;
    out->items.insert(out->items.end(), rest.begin(), rest.end());
    if (fail || out->items.size() > 0xFFFFu) { out->items = rws; out->seg.assign(shifts.size(), 0u); out->ext = false; return; }
    *masks = m2;
}

// the J-mer index with a descriptor's substitutions applied (descriptor form: tests and the emulator's reference path)
GM_HD uint32_t jump_apply(uint32_t idx, uint32_t d, uint32_t J)
{
    for (uint32_t k = 0; k < (d & 7u); ++k) {
        const uint32_t f = (d >> (3u + 6u * k)) & 63u, sh = 2u * (J - 1u - (f & 15u)), old = (idx >> sh) & 3u;
        idx ^= (old ^ ((old + (f >> 4)) & 3u)) << sh;
    }
    return idx;
}
// the same with a rotation word: every 2-bit letter of idx advanced by the rotation in the same field, without carries between fields
GM_HD uint32_t rot_add(uint32_t idx, uint32_t rw)
{
    return (((idx & 0x33333333u) + (rw & 0x33333333u)) & 0x33333333u) | (((idx & 0xCCCCCCCCu) + (rw & 0xCCCCCCCCu)) & 0xCCCCCCCCu);
}
GM_HD uint32_t rot_errors(uint32_t rw)   // substitutions of a rotation word
{
    const uint32_t nz = (rw | rw >> 1) & 0x55555555u;
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(nz);
#else
    return (uint32_t)__builtin_popcount(nz);
#endif
}

// index of a MID group's bitmaps: the J-mer index with its two lowest 6-bit fields swapped (the group's characters become the bit number)
GM_HD uint32_t jump_swap_mid(uint32_t idx) { return (idx & ~0xFFFu) | (idx & 63u) << 6 | ((idx >> 6) & 63u); }
// word of a J-mer in the bitmap of the layout whose three characters sit at bit offset sh of the index (its bit: (idx >> sh) & 63): the
// index with those six bits taken out.  sh = 0: idx >> 6 (LOW); sh = 6: jump_swap_mid(idx) >> 6 (MID).
GM_HD uint32_t group_word(uint32_t idx, uint32_t sh) { return (uint32_t)(((uint64_t)idx >> (sh + 6u)) << sh) | (idx & ((1u << sh) - 1u)); }

// Bitmap word of the 64 J-mers that share the first J - 3 characters: bit (c0 << 4 | c1 << 2 | c2) = the J-mer ending in letters c0 c1 c2
// occurs.  Returns the same bits indexed by ROTATIONS relative to the needle's last three letters low6 = n0 << 4 | n1 << 2 | n2:
// bit (r0 << 4 | r1 << 2 | r2) = the J-mer ending in (n0 + r0) & 3, (n1 + r1) & 3, (n2 + r2) & 3 occurs.
GM_HD uint64_t word_to_rotations(uint64_t w, uint32_t low6)
{
    const uint32_t s0 = 16u * ((low6 >> 4) & 3u), s1 = 4u * ((low6 >> 2) & 3u), s2 = low6 & 3u;
    if (s0) w = (w >> s0) | (w << (64u - s0));                                         // r0: rotate the four 16-bit quarters
    const uint64_t lo1 = 0x0001000100010001ull * ((1ull << (16u - s1)) - 1ull);        // r1: the four nibbles of every quarter
    w = ((w >> s1) & lo1) | ((w << (16u - s1)) & ~lo1);
    const uint64_t lo2 = 0x1111111111111111ull * ((1ull << (4u - s2)) - 1ull);         // r2: the four bits of every nibble
    w = ((w >> s2) & lo2) | ((w << (4u - s2)) & ~lo2);
    return w;
}

// ---- the pattern-fetch state machine of a lane (gm_kernels.h: search_body, part B; tests/emu runs the same functions) ---------------------
// fs of a lane with pattern work = 2 | flags: table entry in flight, bitmap word in flight, jd holds an item, that item is a group; layout
// of the group item in jd; half the bit offset of the layout the live rotations (galive) belong to; the two letters behind the J-mer and
// whether both are letters.  The one GPU-only bug of round 4 lived in exactly this logic (a flag that outlived its group: the lane asked
// for the same word for ever); it is header code now so that the CPU harness walks every root through it, with an iteration bound.
constexpr uint32_t JF_ENTRY = 4u, JF_WORD = 8u, JF_ITEM = 16u, JF_GROUP = 32u;
constexpr uint32_t JF_IL_SHIFT = 6u, JF_IL_MASK = 7u << 6;       // layout of the group item in jd
constexpr uint32_t JF_CLS_SHIFT = 9u, JF_CLS_MASK = 15u << 9;    // half the bit offset of the layout the live rotations belong to
constexpr uint32_t JF_EXT_SHIFT = 13u, JF_EXTOK = 1u << 17;      // the two letters behind the J-mer; both are letters
// ends of the layouts' groups among a search's items: six 16-bit numbers in x, y, z (gm_api.hip: jinfo2); GROUP_MAX_LAYOUTS = a plain pattern
GM_HD uint32_t item_layout(uint32_t jp, uint32_t ex, uint32_t ey, uint32_t ez)
{
    return (jp >= (ex & 0xFFFFu)) + (jp >= (ex >> 16)) + (jp >= (ey & 0xFFFFu)) + (jp >= (ey >> 16)) + (jp >= (ez & 0xFFFFu)) + (jp >= (ez >> 16));
}
// flags of a freshly loaded item number jp
GM_HD uint32_t jump_item_flags(uint32_t jp, uint32_t ex, uint32_t ey, uint32_t ez)
{
    const uint32_t il = item_layout(jp, ex, ey, ez);
    return JF_ITEM | (il < GROUP_MAX_LAYOUTS ? JF_GROUP | il << JF_IL_SHIFT : 0u);
}
struct JumpStep {          // what a lane does in this iteration
    bool want;             // the item in jd has been used up: load the next one (if any) and set jump_item_flags
    bool asked;            // request word `widx` of bitmap plane `wsel`; the caller sets JF_WORD
    bool go;               // read the table entry of the J-mer rot_add(jb, rw); the caller sets JF_ENTRY (errors of the pattern: rot_errors(rw))
    uint32_t widx, wsel, rw;
};
// Tab::layout(il) -> {bit offset: 5 bits, kind-0 plane: 8, first kind-1 plane: 8}; Tab::mask(id) -> the group mask
template <class Tab>
GM_HD JumpStep jump_decide(uint32_t& fs, uint32_t jd, uint32_t& gcur, unsigned long long& galive, unsigned long long pw, uint32_t jb, const Tab& tab)
{
    JumpStep D; D.want = D.asked = D.go = false; D.widx = D.wsel = D.rw = 0u;
    if ((fs & (JF_ITEM | JF_WORD | JF_GROUP)) == (JF_ITEM | JF_GROUP)) {   // a group item: request its word now
        const uint32_t ly = tab.layout((fs >> JF_IL_SHIFT) & 7u);
        const uint32_t sh = ly & 31u, kind = (jd >> (sh + 3u)) & 1u;
        if (kind && !(fs & JF_EXTOK)) {   // a needle N behind the J-mer: no pattern without budget can match
            fs &= ~(JF_ITEM | JF_GROUP | JF_IL_MASK);
            D.want = true;
        } else {
            D.widx = group_word(rot_add(jb, jd & ~(63u << sh)), sh);
            // the bitmaps are one array of planes: kind 0 LOW | kind 1 LOW, 16 letter pairs | kind 1 MID, 16 letter pairs | kind 0 of the other layouts
            D.wsel = kind ? ((ly >> 13) & 255u) + ((fs >> JF_EXT_SHIFT) & 15u) : (ly >> 5) & 255u;
            D.asked = true;
        }
    } else if (galive == 0ull && (fs & JF_WORD)) {   // the word of group jd has arrived: its patterns that pass
        const uint32_t sh = tab.layout((fs >> JF_IL_SHIFT) & 7u) & 31u;
        galive = word_to_rotations(pw, (jb >> sh) & 63u) & tab.mask((jd >> sh) & 7u);
        gcur = jd & ~(63u << sh);
        fs = (fs & ~(JF_WORD | JF_ITEM | JF_GROUP | JF_IL_MASK | JF_CLS_MASK)) | (sh >> 1) << JF_CLS_SHIFT;
        D.want = true;
    }
    if (!(fs & JF_ENTRY)) {
        if (galive != 0ull) {   // the next pattern of the current group that passed
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t bit = (uint32_t)__ffsll((long long)galive) - 1u;
#else
            const uint32_t bit = (uint32_t)__builtin_ctzll(galive);
#endif
            D.rw = gcur | bit << (((fs >> JF_CLS_SHIFT) & 15u) * 2u);
            galive &= galive - 1ull;
            D.go = true;
        } else if ((fs & (JF_ITEM | JF_WORD | JF_GROUP)) == JF_ITEM && !D.want && !D.asked) {   // a plain pattern
            D.rw = jd; D.go = true;
            fs &= ~JF_ITEM;
            D.want = true;
        }
    }
    return D;
}
// after the caller has applied the step (item loaded, JF_WORD / JF_ENTRY set): is the root's pattern work over?
GM_HD bool jump_done(uint32_t fs, unsigned long long galive) { return !(fs & (JF_ENTRY | JF_WORD | JF_ITEM)) && galive == 0ull; }

}  // namespace gm
