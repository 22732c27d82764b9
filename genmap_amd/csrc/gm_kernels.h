// gm_kernels.h -- device kernels of the mappability hot path (HIP, gfx950 only).
//
//   search_kernel<WPP>   persistent wavefronts; every lane owns one search node at a time and advances it
//                        with gm::lane_step (gm_engine.h).  Work items are (k-mer block, strand, OSS search)
//                        roots drawn from a global counter in chunks, handed out inside the wavefront with
//                        a ballot/mbcnt rank -- the "warp-ballot work queue" of the search tree.  Lanes keep
//                        their pending nodes in a lane-private LIFO in HBM/L2 (16-byte nodes, one
//                        global_store_dwordx4 / global_load_dwordx4 each).
//                        With Env::NODES (round 6) the same loop is the WALKER of the split search: lanes draw node packets
//                        that phase A wrote (one round trip each: header into registers, needle window by LDS DMA).
//   expand_kernel        phase A of the split search (gm_expand.h): one lane per (k-mer block, strand, search, item) turns the
//                        jump patterns of a root into node packets -- bitmap word, table entries dealt out to all lanes of the
//                        wavefront, neighbour filters -- in three lists by the pattern's substitutions; expand_slice_begin /
//                        _commit_kernel cut a call into slices that fit the packet buffers (decided on the device).
//   finalize_kernel      acc (u32, saturating-safe) -> c[] as uint8/uint16:  min(total, MAX)
//                        (std::min(count + hits, max_val), /root/reference/src/algo.hpp:36,48,191)
//   reset_limits_kernel  resetLimits, /root/reference/src/algo.hpp:10-22
//
// Roofline: integer/pointer chasing, bound by random HBM line reads; no MFMA anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include "gm_engine.h"
#include "gm_expand.h"

namespace gm {

// ---- phase A / phase B of the split search (gm_expand.h): control block of one slice, device memory ------------------------------------
// Phase A appends node packets to two buffers: X holds the packets of patterns without a substitution from its front and those with one
// substitution from its back, Y those with two or more.  A wavefront reserves REGIONS of packets (one device-scope atomic per region and
// class) and fills them privately; what it leaves unfilled keeps an older stamp and is skipped by the walker.  A reservation that does not
// fit marks the wavefront's chunk of work as failed: the walker drops the packets of chunks >= failFrom, the next slice starts there.
struct ExpandCtl {
    unsigned long long nextChunk;        // work counter of phase A
    unsigned long long tailsX;           // packets reserved in X: class 0 | class 1 << 32
    unsigned long long reserved0;        // (the walker's counters are the stripes below)
    unsigned long long chunkBegin, chunkEnd;
    uint32_t tailY, failFrom;
    uint32_t valid[3];                   // per class: end of the last reservation that fitted (fit is monotone in the order of the atomics)
    uint32_t stamp;
    uint32_t t[3];                       // what the walker draws: valid[] at the end of phase A
    uint32_t pad;
    // The walker's work counters.  Packets are drawn in pools of WORK_CHUNK_NODES; pool number c * XSTRIPES + j comes from stripe j's
    // counter.  One counter would be asked 13 M times a second at K=30 e=2 -- all a single address serves -- and larger pools leave the
    // wavefronts of a slice finishing far apart (2048 per pool: 137 -> 238 ms); a wavefront starts at its own stripe and moves on to the
    // next when one has run out, so the stripes also even each other out at the end of a slice.
    unsigned long long walk[16 * 16];    // stripe j at walk[16 j]: 128 bytes apart
};
constexpr uint32_t XSTRIPES = 16;
struct ExpandProgress {
    unsigned long long committed;        // chunks of the call that are done (searched, or handed to a walker launch queued behind)
    unsigned long long totalChunks;
    uint32_t stampCounter, slices, lastChunks, lastX, lastY, pad;
};
constexpr uint32_t XREGION = 1024;       // packets per reservation (a returning device-scope atomic on ONE address: the part serves ~15 M of them per second --
                                         // measured: phase A with chunks of 4 / 18 / 72 blocks 335 / 132 / 75 ms, profiles/r06 -- so every counter here is drawn from rarely)

struct SearchArgs {
    const uint32_t* blk[2];     // rank blocks: [0] forward BWT (extend left), [1] reverse BWT (extend right)
    uint64_t C[NLET + 1];       // (rows are row_t = uint32_t, or uint64_t in the wide geometry WPP = 2; the arguments carry 64 bits)
    uint64_t nRows;
    const uint8_t* text;        // slice base, one code per byte
    uint32_t* acc;              // per slice position, zeroed by the caller (two planes of accPlane entries with StoreEnv)
    uint32_t* diff;             // CountEnv: difference plane of the verified runs (nullptr: runs are added k-mer by k-mer into acc).  A run of
                                // k-mers [lo, hi] of ONE block adds +1 at lo and -1 at hi + 1 (nothing when hi is the block's last k-mer);
                                // finalize_diff_kernel sums the plane inside each block and adds it to acc.  Regular partition only.
    uint64_t accPlane;
    uint32_t maxVal;            // 255 or 65535: the result is min(total, maxVal), so saturated k-mers need no further hits
    uint32_t addCap;            // a single add into acc is clamped to this (min(maxVal, 65535): the result is min(total, maxVal) either way)
    uint32_t noWrap;            // 1: the host has shown that no accumulator of this call can reach 2^32 (gm_api.hip: acc_cannot_wrap) -- the adds
                                // are fire-and-forget; 0: every add returns the old value and a (theoretical) wrap sets the sticky top bit
    uint32_t K, E;
    uint32_t stepSize, nSearches, rootsPerBlock;
    uint64_t numKmers;
    uint64_t blockBegin;        // first block of this shard
    uint64_t numRoots;          // roots of this shard
    const uint2* blockList;     // (first k-mer, count) per block, or nullptr for the arithmetic partition
    const uint4* table;         // OssRecord[(n-1)*8 + s]
    uint4* stack;               // lane-private LIFOs: stack[lane * stackDepth + i]
    uint32_t stackDepth;            // total entries a lane may stack (LDS part + spill part)
    uint32_t spillDepth;            // entries per lane in `stack`
    unsigned long long* workCounter;
    uint32_t* errorFlag;            // sticky: bit 0 lane stack overflow, bit 1 a wavefront ran past guardCap (a hung search ends as GM_ERR_INTERNAL)
    uint32_t guardCap;              // a wavefront gives up when its guard counter passes this: the counter counts loop iterations and falls back to
    uint32_t guardKeep;             // (counter & guardKeep) whenever a lane held a node or something was verified -- 0: CONSECUTIVE idle iterations are
                                    // bounded (knob "stall_cap", default 2^22), ~0: all iterations are (knob "iter_cap": tests force the bound)
    unsigned long long* counters;   // [0] node steps, [1] distinct rank lines (only with GM_COUNTERS)
    // ---- locate path (csv, --exclude-pseudo; /root/reference/src/algo.hpp:311-387) ----
    const void* sa;                 // forward suffix array (sentinel-text positions, row_t each), sampling rate 1; nullptr when sampled
    const uint2* saMark;            // sampled suffix array: per 32 rows {mark bits, samples before the word} ...
    const void* saSamples;          // ... and the values of the marked rows (row_t each)
    const uint64_t* cumGlobal;      // sentinel-free cumulative sequence lengths of the WHOLE index, nSeqGlobal + 1
    uint32_t nSeqGlobal;
    const uint32_t* seqFile;        // fasta id per global sequence (mappingSeqIdFile, src/mappability.hpp:230-248)
    const uint8_t* rowFile;         // optional: fasta id of the sequence that suffix-array row r lies in (built once per index and file assignment): --exclude-pseudo
                                    // reads ONE byte per located row instead of suffix array entry -> sequence lookup -> seqFile
    uint32_t* fileBits;             // [pos * wordsPerKmer + w]: set of fasta ids seen for the k-mer at pos
    uint32_t wordsPerKmer;
    uint64_t posBase;               // window origin of cnt2 / offs / emit arrays (slice position)
    uint64_t windowLen;
    uint32_t* cnt2;                 // [strand * windowLen + pos - posBase]: occurrences (pass 1) / cursor (pass 2)
    const uint64_t* offs;           // exclusive scan of cnt2 (pass 2)
    uint64_t* emit;                 // packed (seqNo << 32 | seqPos) per occurrence
    // ---- verification of narrow nodes (gm_engine.h: verify_item) ----
    const uint8_t* textS;           // sentinel text (one code per byte, 5 = sentinel), nRows bytes
    const uint4* ctx;               // optional: per forward SA row one 32-byte record {SA[row], 56 symbols around it} (CTX_* below)
    uint32_t verifyT;               // nodes with range width <= verifyT are resolved by verification (0 = off)
    uint32_t verifyTExt;            // ... and this wide once the infix is complete (extension phase)
    uint32_t patBatch;              // jump patterns: parts A / B of the loop run when this many idle lanes wait for their pattern turn (or no lane holds a node); 1: every iteration
    uint32_t fetchBatch;            // roots are drawn when this many lanes are idle (or nothing else is left): the fetch code runs per batch
    uint32_t satMinW;               // saturation is looked up (a global read per covered k-mer) only for nodes at least this wide
    uint32_t probation;             // a single-row node that has spent every error is stepped this many times before it is verified
    uint32_t verifyCost;            // ... when width * verifyCost <= estimated rank steps left below the node
    uint32_t fastVerify;            // 1: every item of a verification round is settled from ONE round of reads (record + needle window, gm_engine.h: fv_masks);
                                    //    the host sets it when K <= 32, the window fits FV_MAXW symbols and no node can be verified right of the record's anchor range
    uint32_t nbFilter;              // != 0: one-row table entries are compared with the needle's next characters before they become nodes; 1: two-row entries too
    uint32_t selfHit;               // 1: a single error-free row on the forward strand is the window's own location -- no lookup at all
    // ---- LDS staging (per wavefront): verification queue | top of the lane stacks | packed needle windows ----
    const uint4* text4;             // whole text, 4 bits per symbol (32 symbols per 16-byte chunk), sentinel-free
    const uint4* text2;             // win2: the text at 2 bits per symbol (64 symbols per chunk; an N is stored as A) ...
    const uint8_t* nflag;           // ... and one bit per chunk of it: the chunk holds an N (roots whose window touches such a chunk read their needle from text4: text_char)
    uint32_t win2;                  // 1: roots stage their needle windows from text2 (long windows: two chunks less per lane, the LDS of two stack levels)
    uint64_t textBegin;             // slice offset inside the text (symbols)
    uint32_t vqCap;                 // queue entries per wavefront
    uint32_t verifyRows;            // rows of one node queued per iteration (1 or 2)
    uint32_t ldsDepth;              // stack entries per lane kept in LDS (deeper ones spill to `stack`)
    uint32_t winChunks;             // 16-byte chunks per lane for the needle window
    uint32_t entrySlots;            // != 0: 4 KB of LDS per block hold the table entries in flight of the jump patterns (one slot per lane)
    uint32_t lqCap;                 // leaf queue entries per wavefront (locating leaf policies: FileSetEnv, OccEmitEnv; 0 elsewhere)
    // ---- q-mer range tables: the first (always exact) OSS block of a root starts from a lookup instead of q steps ----
    const uint4* qtabA;             // {fwd lo, rev lo, width, 0} of every ACGT string of length q (two tables at most per call)
    const uint4* qtabB;
    uint32_t qlenPacked[2];         // 8 bits per search: table prefix length q_s (0 = no table for that search), at most 16
    uint32_t qselMask;              // bit s: search s uses qtabB
    uint32_t startPacked[2];        // 8 bits per search: startPos of the regular block shape (n == stepSize)
    uint32_t skipDup;               // 1: the range-hi block is not loaded when it is the range-lo block (saves a translation per shared step)
    uint32_t coop;                  // 1: rank blocks are read by groups of lanes (rank2_coop); 32- and 64-byte blocks
    uint32_t steal;                 // > 0: lanes without work take the bottom of a neighbour's stack (work sharing inside the wavefront),
                                    //      an exchange runs when at least this many lanes are idle
    uint32_t chunkBlocks, chunkStride, chunkIndex;   // != 0: this call owns the chunks c = chunkIndex (mod chunkStride) of chunkBlocks blocks each
    // ---- jump patterns (gm_oss.h): kernels compiled with Env::JUMPS start every regular root at depth jumpJ ----
    const uint32_t* patterns;       // descriptors of every search, back to back
    const uint4* jinfo;             // per search {first pattern | patterns << 16, meta at depth J relative to n - 1, first descriptor, 0}
    const uint4* jtab;              // table of all J-mers
    const unsigned long long* jbits;   // bitmaps of the groups of patterns, (1 + 16 + 16) x jbitsWords words: "the J-mer occurs" | "... followed by letters x y" (16 pairs) | the same in the MID layout
    uint64_t jbitsWords;            // 4^J / 64
    const uint4* jinfo2;            // per search: where the groups of each layout end among its items (16 bits each, x lo, x hi, y lo, ...), w: groups of kind 1?
    uint32_t layShift[8];           // layouts of the call's groups (gm_oss.h: group_layout_shifts): bit offset of the three characters in the J-mer index,
    uint32_t layPlane0[8];          // plane of the layout's kind-0 bitmap inside jbits,
    uint32_t layPlane1[8];          // first of the 16 planes of its kind-1 bitmaps (LOW and MID only)
    unsigned long long gmask[8];    // masks of the groups (gm_oss.h: GROUP_MAX_MASKS)
    uint32_t jumpJ;                 // 0: no jumps in this call (every root starts at the tree's root)
    uint32_t jumpAPacked[2];        // 8 bits per search: window coordinate of the J-mer's first character, minus (n - 1)
    // ---- correction pass (ScatterEnv): occurrences are added at THEIR OWN position if the main pass computes that position ----
    uint64_t sliceBegin, sliceLen;  // the slice of the main pass inside the whole text
    uint64_t ownBegin, ownEnd;      // slice positions the call owns (a shard's range), and of those the chunks chunkIndex (mod chunkStride)
    uint32_t ownChunkLen;           // positions per chunk (0: no chunks)
    const uint2* selBlocks;         // a selection's blocks, sorted by position (nullptr: every position is computed)
    uint32_t nSelBlocks;
    // ---- k-mers longer than MAX_K (gm_longk.h) ----
    const void* tableL;             // OssRecordL[(n-1)*8 + s]
    // ---- phase A / phase B of the split search (gm_expand.h; ExpandCtl above) ----
    uint4* pktX; uint4* pktY;       // node packets, (2 + pktChunks) x 16 bytes each
    uint32_t capX, capY;            // packets the buffers hold
    uint32_t pktChunks;             // 16-byte chunks of needle window per packet
    ExpandCtl* xctl;
    const uint32_t* wmapRoots;      // phase A: (strand, search) of root q of a block (the work map of the first of two passes)
    uint32_t rootsPerBlockA;        // strands x searches
    uint32_t xshare;                // 1: the root contexts of a group of blocks are computed once, by a lane per ROOT, and read by the roots' items from LDS
    uint32_t xmode;                 // phase A: 0 every pattern of every root (one pass), 1 the patterns without a substitution (a work item per root), 2 the rest, for blocks not at MAX yet
    const uint32_t* wmap;           // work item of a block -> search | strand << 3 | item << 8 (gm_expand.h: make_wmap)
    uint32_t itemsPerBlock;         // work items per k-mer block
    uint32_t expandBlocks;          // k-mer blocks per chunk of phase A
    uint64_t numBlocksCall;         // k-mer blocks of the call (numRoots / rootsPerBlock)
    uint32_t ldsPad;                // measurement: bytes of LDS requested on top of what a block uses (knob "lds_pad"; travels with the call: two host threads may drive two indexes)
    uint32_t satDrawW;              // walker: a drawn node at least this wide is dropped when all k-mers of its block are at MAX already
};

// which positions of a range belong to the calling shard (interleaved chunks of `len` positions); len == 0: all of them
struct ChunkSel { uint32_t len, stride, index; };
// The calling shard's positions of [0, n) as ranges: range q = [b, e).  Kernels over positions walk blockIdx.y over these ranges
// and blockIdx.x / threads inside one, so that a shard touches only its own chunks, without a division per position.
__device__ __forceinline__ uint64_t own_ranges(const ChunkSel& c, uint64_t n)
{
    if (c.len == 0u) return n ? 1u : 0u;
    const uint64_t chunks = (n + c.len - 1) / c.len;                       // chunks of [0, n), the last one possibly short
    return chunks > c.index ? (chunks - c.index + c.stride - 1) / c.stride : 0u;
}
__device__ __forceinline__ void own_range(const ChunkSel& c, uint64_t n, uint64_t q, uint64_t& b, uint64_t& e)
{
    if (c.len == 0u) { b = 0; e = n; return; }
    b = (q * c.stride + c.index) * (uint64_t)c.len;
    e = b + c.len < n ? b + c.len : n;
}
// saturating sums of packed unsigned bytes / halfwords (a plane holds min(count, MAX) of one strand)
__device__ __forceinline__ uint32_t sat_add_u8x4(uint32_t x, uint32_t y)
{
    const uint32_t sum = ((x & 0x7F7F7F7Fu) + (y & 0x7F7F7F7Fu)) ^ ((x ^ y) & 0x80808080u);
    const uint32_t c = ((x & y) | ((x | y) & ~sum)) & 0x80808080u;        // carry out of each byte
    return sum | ((c >> 7) * 0xFFu);
}
__device__ __forceinline__ uint32_t sat_add_u16x2(uint32_t x, uint32_t y)
{
    const uint32_t sum = ((x & 0x7FFF7FFFu) + (y & 0x7FFF7FFFu)) ^ ((x ^ y) & 0x80008000u);
    const uint32_t c = ((x & y) | ((x | y) & ~sum)) & 0x80008000u;
    return sum | ((c >> 15) * 0xFFFFu);
}


// sentinel-text position -> (seqNo, seqPos); sequence s starts at cum[s] + s
// (seqPos is reported in 32 bits: csv / --exclude-pseudo need every single sequence to be shorter than 2^32, gm_api.hip checks)
__device__ __forceinline__ uint2 locate_position(const uint64_t* __restrict__ cum, uint32_t nSeq, uint64_t p)
{
    if (nSeq <= 32u) {
        // few sequences (a handful of genomes: what --exclude-pseudo and csv are used on): one pass over the sequence starts with
        // wave-uniform addresses (scalar loads) instead of log2(nSeq) DEPENDENT vector loads per located row -- the drain of the leaf
        // queue spent most of its time in that chain (profiles/r03/c5_block_shape.txt)
        uint32_t s = 0; uint64_t base = 0;
#pragma unroll 4
        for (uint32_t i = 1; i < nSeq; ++i) { const uint64_t st = cum[i] + i; if (st <= p) { s = i; base = st; } }
        return make_uint2(s, (uint32_t)(p - base));
    }
    uint32_t lo = 0, hi = nSeq;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cum[mid] + mid <= p) lo = mid; else hi = mid; }
    return make_uint2(lo, (uint32_t)(p - (cum[lo] + lo)));
}

// search nodes and queue entries in memory: one 16-byte unit for 32-bit rows, two for 64-bit rows
template <typename R> struct NodeIO;
template <> struct NodeIO<uint32_t> {
    static constexpr uint32_t NU = 1;
    static __device__ __forceinline__ void store(uint4* p, uint32_t stride, const NodeT<uint32_t>& nd) { p[0] = make_uint4(nd.flo, nd.rlo, nd.w, nd.meta); (void)stride; }
    static __device__ __forceinline__ NodeT<uint32_t> load(const uint4* p, uint32_t stride) { (void)stride; const uint4 v = p[0]; NodeT<uint32_t> nd; nd.flo = v.x; nd.rlo = v.y; nd.w = v.z; nd.meta = v.w; return nd; }
    // verification queue entry: row, meta, window origin, n | strand << 8 | search << 9
    static __device__ __forceinline__ void store_item(uint4* p, uint32_t row, uint32_t meta, uint32_t win, uint32_t nss) { p[0] = make_uint4(row, meta, win, nss); }
    static __device__ __forceinline__ void load_item(const uint4* p, uint32_t& row, uint32_t& meta, uint32_t& win, uint32_t& nss) { const uint4 v = p[0]; row = v.x; meta = v.y; win = v.z; nss = v.w; }
    // q-mer table entry {fwd lo, rev lo, width}
    static __device__ __forceinline__ void load_qentry(const uint4* tab, uint32_t idx, uint32_t& flo, uint32_t& rlo, uint32_t& w) { const uint4 v = tab[idx]; flo = v.x; rlo = v.y; w = v.z; }
    static __device__ __forceinline__ void load_qentry(const uint4* tab, uint32_t idx, uint32_t& flo, uint32_t& rlo, uint32_t& w, uint32_t& nb) { const uint4 v = tab[idx]; flo = v.x; rlo = v.y; w = v.z; nb = v.w; }
    // nb: the text next to the string's ONLY occurrence (qmer_table_kernel), 0 for every other entry
    static __device__ __forceinline__ void store_qentry(uint4* tab, uint32_t idx, uint32_t flo, uint32_t rlo, uint32_t w, uint32_t nb = 0u) { tab[idx] = make_uint4(flo, rlo, w, nb); }
};
template <> struct NodeIO<uint64_t> {
    static constexpr uint32_t NU = 2;
    static __device__ __forceinline__ void store(uint4* p, uint32_t stride, const NodeT<uint64_t>& nd)
    {
        p[0] = make_uint4((uint32_t)nd.flo, (uint32_t)(nd.flo >> 32), (uint32_t)nd.rlo, (uint32_t)(nd.rlo >> 32));
        p[stride] = make_uint4((uint32_t)nd.w, (uint32_t)(nd.w >> 32), nd.meta, 0u);
    }
    static __device__ __forceinline__ NodeT<uint64_t> load(const uint4* p, uint32_t stride)
    {
        const uint4 a = p[0], b = p[stride];
        NodeT<uint64_t> nd; nd.flo = (uint64_t)a.y << 32 | a.x; nd.rlo = (uint64_t)a.w << 32 | a.z; nd.w = (uint64_t)b.y << 32 | b.x; nd.meta = b.z; return nd;
    }
    static __device__ __forceinline__ void store_item(uint4* p, uint64_t row, uint32_t meta, uint64_t win, uint32_t nss)
    {
        p[0] = make_uint4((uint32_t)row, (uint32_t)(row >> 32), meta, nss); p[1] = make_uint4((uint32_t)win, (uint32_t)(win >> 32), 0u, 0u);
    }
    static __device__ __forceinline__ void load_item(const uint4* p, uint64_t& row, uint32_t& meta, uint64_t& win, uint32_t& nss)
    {
        const uint4 a = p[0], b = p[1]; row = (uint64_t)a.y << 32 | a.x; meta = a.z; nss = a.w; win = (uint64_t)b.y << 32 | b.x;
    }
    // q-mer table entry of the wide geometry: 16 bytes like the narrow one -- {fwd lo, rev lo, width} as 40-bit numbers (an index has fewer
    // than 2^40 rows): x, y, z = their low words, w = their bits 32..39 in bytes 0, 1, 2.  (Round 4 kept three 64-bit numbers in 32 bytes: two
    // loads per root, and a table of all 16-mers would have taken 137 GB -- the table stopped at 14 characters.)
    static __device__ __forceinline__ void load_qentry(const uint4* tab, uint32_t idx, uint64_t& flo, uint64_t& rlo, uint64_t& w)
    {
        const uint4 a = tab[idx];
        flo = (uint64_t)(a.w & 0xFFu) << 32 | a.x; rlo = (uint64_t)((a.w >> 8) & 0xFFu) << 32 | a.y; w = (uint64_t)((a.w >> 16) & 0xFFu) << 32 | a.z;
    }
    static __device__ __forceinline__ void load_qentry(const uint4* tab, uint32_t idx, uint64_t& flo, uint64_t& rlo, uint64_t& w, uint32_t& nb) { load_qentry(tab, idx, flo, rlo, w); nb = 0u; }
    static __device__ __forceinline__ void store_qentry(uint4* tab, uint32_t idx, uint64_t flo, uint64_t rlo, uint64_t w, uint32_t = 0u)
    {
        tab[idx] = make_uint4((uint32_t)flo, (uint32_t)rlo, (uint32_t)w, ((uint32_t)(flo >> 32) & 0xFFu) | ((uint32_t)(rlo >> 32) & 0xFFu) << 8 | ((uint32_t)(w >> 32) & 0xFFu) << 16);
    }
};

// k-mer starts (block coordinates) still covered by a node
__device__ __forceinline__ void covered_kmers(uint32_t meta, uint32_t n, uint32_t K, uint32_t& smin, uint32_t& smax)
{
    const uint32_t a = meta_a(meta), bx = meta_bx(meta), t = meta_t(meta), md = meta_mode(meta);
    if (md == M_OSS) { smin = 0u; smax = n - 1u; }
    else if (md == M_EXT_R) { smin = t - K; smax = a; }
    else if (md == M_EXT_L) { smin = bx - K; smax = t; }
    else { smin = bx - K; smax = a; }
}

// Verification record of a suffix-array row (built once per index when HBM allows, gm_api.hip: make_ctx): word 0 = SA[row] = p0,
// words 1..7 = the 56 symbols textS[p0 - CTX_LEFT .. p0 - CTX_LEFT + 55], 4 bits each (symbol i in bits 4(i%8) of word 1 + i/8).
// One aligned 32-byte read replaces the dependent pair "SA entry, then text around it" (two to three random requests).
// (CTX_LEFT = 24, CTX_SYMS = 56: gm_engine.h, next to the fast verification that reads whole windows from a record)

// (NB_SYMS, NB_SYMS2: neighbour symbols per side carried by one- / two-row q-mer table entries -- gm_expand.h)
constexpr uint32_t STEAL_LEVELS = 16;  // a lane gives away at most this many bottom entries before its stack has run empty once

// fetch state of a lane with jump patterns: fs = 2 | flags.  Table entry in flight, bitmap word in flight, jd holds an item, that item is a
// group / a MID group, the current set of live rotations belongs to a MID group; bits 8..11 the two letters behind the J-mer, bit 12: both are letters
// (JF_* flags of the pattern-fetch state, item_layout, jump_decide: gm_oss.h -- shared with the CPU harness)
constexpr uint32_t WORK_CHUNK = 256;   // roots taken from the global counter per atomic
constexpr uint32_t WORK_CHUNK_NODES = 256;    // node packets per pool of the walker of the split search (striped counters: ExpandCtl::walk)
constexpr uint32_t VERIFY_TMAX = 16;   // widest range resolved by verification
constexpr uint32_t VERIFY_ROWS = 2;    // rows of one node queued per iteration at most (SearchArgs::verifyRows; the rest waits on the lane's stack)

template <int WPP> struct EnvBase {
    static constexpr bool EXACT_ONLY = false;   // StoreEnv: the kernel only ever runs with E = 0
    static constexpr bool NLESS = false;        // the text letter N is never followed (gm_engine.h); a correction pass adds those occurrences
    static constexpr bool JUMPS = false;        // regular roots start from the jump patterns of their search (gm_oss.h)
    static constexpr bool LEAFQ = false;        // leaves are queued per wavefront and located 64 rows at a time (LeafQueueEnv)
    static constexpr bool SELF_HIT = false;     // frequency policies: a lone error-free row on the forward strand is the window itself
    static constexpr bool RANGE_ADD = false;    // the policy takes a run of hit k-mers whole (leaf_range) instead of one leaf_at per k-mer
    static constexpr bool LEGACY_LOOP = false;  // the loop order of round 3 (root draw in front of the verification, staged root context): StoreEnv
    static constexpr bool NODES = false;        // the walker of the split search: lanes draw node packets written by phase A (gm_expand.h) instead of roots
    typedef typename BlockGeom<WPP>::row_t row_t;
    typedef NodeT<row_t> Node;
    typedef RootT<row_t> Root;
    typedef NodeIO<row_t> IO;
    const SearchArgs& A;
    uint4* stk;          // global spill area of this lane
    uint4* lstk;         // LDS: [depth][lane] of this wavefront, already offset by the lane
    const uint8_t* lwin; // LDS: [chunk][lane] 16-byte chunks of this lane's packed window, already offset by the lane
    uint32_t woff;       // nibble offset of the window inside its first chunk
    uint32_t sp;         // entries on the lane's stack; they live at levels [sbase, sbase + sp)
    uint32_t sbase;      // raised when a neighbour takes the bottom entry (work sharing), back to 0 when the stack runs empty
    uint32_t K;
#ifdef GM_COUNTERS
    uint32_t steps = 0, lines = 0, stOss = 0, stExt = 0, stExtW1 = 0, stExtW4 = 0, stOssW1 = 0, pushes = 0, vItems = 0, vItemsOss = 0, vChunks = 0, jumps = 0, jumpDrops = 0;
    uint32_t whit[17] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // (16: part B of the jump patterns; not exported)
    mutable uint32_t locRows = 0, lfSteps = 0;   // rows located (one suffix-array or mark-word read each), LF steps of sampled walks
    uint32_t jumpWords = 0, jumpDrops2 = 0;       // words of the existence bitmap read for groups of jump patterns; rows of two-row entries ended by the neighbour filter
    uint32_t maxSp = 0, selfHits = 0, runs = 0;  // deepest lane stack, self hits, verified runs of k-mers handed to the leaf policy
    // a wavefront passed here (counted by its first enabled lane): the dynamic cost of a region = passes x its instructions
    __device__ __forceinline__ void note_wave(int i) { const unsigned long long m = __ballot(true); if (__lane_id() == (unsigned)(__ffsll((long long)m) - 1)) whit[i]++; }
    __device__ __forceinline__ void note_chunk() { vChunks++; }
    __device__ __forceinline__ void note_run() { runs++; }
    __device__ __forceinline__ void note_item(uint32_t mode) { vItems++; vItemsOss += (mode == M_OSS); }
    __device__ __forceinline__ void note_step(uint32_t mode, row_t w)
    {
        if (mode == M_OSS) { stOss++; stOssW1 += (w == 1u); } else { stExt++; stExtW1 += (w == 1u); stExtW4 += (w > 1u && w <= 4u); }
    }
#else
    __device__ __forceinline__ void note_step(uint32_t, row_t) {}
    __device__ __forceinline__ void note_chunk() {}
    __device__ __forceinline__ void note_run() {}
    __device__ __forceinline__ void note_item(uint32_t) {}
    __device__ __forceinline__ void note_wave(int) {}
#endif
    __device__ __forceinline__ EnvBase(const SearchArgs& a, uint4* s, uint32_t k) : A(a), stk(s), lstk(nullptr), lwin(nullptr), woff(0), sp(0), sbase(0), K(k) {}
    __device__ __forceinline__ Node pop()
    {
        --sp;
        const uint32_t lv = sbase + sp;
        const Node nd = lv < A.ldsDepth ? IO::load(lstk + (size_t)lv * IO::NU * 64u, 64u) : IO::load(stk + (size_t)(lv - A.ldsDepth) * IO::NU * 64u, 64u);
        if (sp == 0u) sbase = 0u;
        return nd;
    }
    __device__ __forceinline__ row_t slice_pos(const Root& rt, uint32_t kmer) const { return rt.win + (rt.strand ? rt.n - 1u - kmer : kmer); }

    __device__ __forceinline__ void rank2(uint32_t right, row_t lo, row_t hi, row_t rl[NLET], row_t rh[NLET])
    {
        constexpr uint32_t SPB = BlockGeom<WPP>::SPB, WPB = BlockGeom<WPP>::WPB;
        constexpr int NV = (BlockGeom<WPP>::HDRW + 3 * WPP + 3) / 4;   // uint4 loads that cover header + planes
        const uint32_t* base = right ? A.blk[1] : A.blk[0];
        const row_t bl = lo / SPB, bh = hi / SPB;
        const uint4* pl = reinterpret_cast<const uint4*>(base + (size_t)bl * WPB);
        const uint4* ph = reinterpret_cast<const uint4*>(base + (size_t)bh * WPB);
        uint32_t wl[NV * 4], wh[NV * 4];
#pragma unroll
        for (int j = 0; j < NV; ++j) { uint4 v = pl[j]; wl[4 * j] = v.x; wl[4 * j + 1] = v.y; wl[4 * j + 2] = v.z; wl[4 * j + 3] = v.w; }
        // range lo and range hi usually share a block once the range is narrow: one request instead of two
        // always two loads: when lo and hi share a block the second one is an L1 hit, which is cheaper than a divergent
        // branch around it (r01h: +5..9 %); the miss count (algorithmic lines) is unchanged
        if (A.skipDup) {   // wave-uniform choice
            // (the placeholder is opaque to the optimiser: otherwise it turns "placeholder, conditional load, select" into
            // "copy of the lo block, conditional load", which waits for the first load before it issues the second)
            const uint32_t z = opaque_zero();
#pragma unroll
            for (int j = 0; j < NV * 4; ++j) wh[j] = z;
            if (bl != bh) {   // lanes whose range ends in the block it starts in issue no second request
#pragma unroll
                for (int j = 0; j < NV; ++j) { uint4 v = ph[j]; wh[4 * j] = v.x; wh[4 * j + 1] = v.y; wh[4 * j + 2] = v.z; wh[4 * j + 3] = v.w; }
            }
            uint32_t same = bl == bh ? 1u : 0u;
            asm volatile("" : "+v"(same));
#pragma unroll
            for (int j = 0; j < NV * 4; ++j) wh[j] = same ? wl[j] : wh[j];
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) { uint4 v = ph[j]; wh[4 * j] = v.x; wh[4 * j + 1] = v.y; wh[4 * j + 2] = v.z; wh[4 * j + 3] = v.w; }
        }
        block_rank<WPP>(wl, (uint32_t)(lo - bl * SPB), rl);
        block_rank<WPP>(wh, (uint32_t)(hi - bh * SPB), rh);
#ifdef GM_COUNTERS
        steps += 1; lines += 1 + (bl != bh);
#endif
    }
    // ---- cooperative form of rank2: G = BYTES/16 adjacent lanes read ONE block with ONE 16-byte load each ----
    // Beyond ~4 GiB of randomly read footprint the memory system charges per lane-load (address translation), not per block
    // or byte (profiles/r02/r02a_gather2_*: 64-B blocks read by one lane 19 G/s, by four lanes 48 G/s).  The lanes of a
    // group take turns as owner: in round r every lane of the group loads its 16-byte piece of the two blocks owner r asked
    // for; a butterfly of DPP exchanges then hands every owner the pieces of its own blocks.  Must be called with all 64
    // lanes enabled; lanes without a query pass lo = hi = 0.
    static __device__ __forceinline__ uint32_t opaque_zero() { uint32_t z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return z; }
    template <int CTRL> static __device__ __forceinline__ uint32_t dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }   // every lane has a source inside its quad: bound_ctrl spares the zeroing of the destination
    template <int CTRL> static __device__ __forceinline__ uint4 dpp4(const uint4& v) { return make_uint4(dpp<CTRL>(v.x), dpp<CTRL>(v.y), dpp<CTRL>(v.z), dpp<CTRL>(v.w)); }
    static __device__ __forceinline__ uint4 sel4(bool c, const uint4& a, const uint4& b) { return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
    // exchange step of the transposition: lanes that differ in bit `bit` of their group index swap lo's upper with hi's lower register
    template <int CTRL> static __device__ __forceinline__ void exchange(bool up, uint4& low, uint4& high)
    {
        const uint4 recv = dpp4<CTRL>(sel4(up, low, high));
        low = sel4(up, recv, low);
        high = sel4(up, high, recv);
    }
    template <int G> static __device__ __forceinline__ void transpose(uint32_t j, uint4 (&V)[G])
    {
        static_assert(G == 2 || G == 4, "groups of 2 or 4 lanes");
        const bool b0 = (j & 1u) != 0u;
        if (G == 2) { exchange<0xB1>(b0, V[0], V[1]); return; }              // quad_perm [1,0,3,2]
        exchange<0xB1>(b0, V[0], V[1]); exchange<0xB1>(b0, V[2 % G], V[3 % G]);
        const bool b1 = (j & 2u) != 0u;
        exchange<0x4E>(b1, V[0], V[2 % G]); exchange<0x4E>(b1, V[1], V[3 % G]);   // quad_perm [2,3,0,1]
    }
    template <int R> __device__ __forceinline__ void coop_round(uint32_t bl, uint32_t bhx, uint32_t j, uint4& L, uint4& H, uint32_t& dup, uint32_t z)
    {
        constexpr int G = (5 + 3 * WPP + 3) / 4;
        constexpr int BC = G == 2 ? (R == 0 ? 0xA0 : 0xF5) : R * 0x55;   // broadcast of lane R of the group: quad_perm [R,R,R,R] / [0,0,2,2] / [1,1,3,3]
        const uint32_t obl = dpp<BC>(bl), obhx = dpp<BC>(bhx);
        const uint32_t obh = obhx & 0x7FFFFFFFu;
        const uint4* pb = reinterpret_cast<const uint4*>((obhx >> 31) ? A.blk[1] : A.blk[0]);
        L = pb[(size_t)obl * G + j];
        H = make_uint4(z, z, z, z);                    // opaque placeholder, see rank2
        if (obl != obh) H = pb[(size_t)obh * G + j];   // group-uniform: nobody touches a block twice
        uint32_t d = obl == obh ? 1u : 0u;
        asm volatile("" : "+v"(d));   // opaque: the final select must not be folded into the branch above (see rank2)
        dup |= d << R;
    }
    __device__ __forceinline__ void rank2_coop(uint32_t right, uint32_t lo, uint32_t hi, uint32_t rl[NLET], uint32_t rh[NLET])   // 32-bit rows only
    {
        constexpr uint32_t SPB = BlockGeom<WPP>::SPB;
        constexpr int G = (5 + 3 * WPP + 3) / 4;
        const uint32_t j = threadIdx.x & (uint32_t)(G - 1);
        const uint32_t bl = lo / SPB, bh = hi / SPB;
        const uint32_t bhx = bh | right << 31;   // block indexes stay below 2^27
        uint4 L[G], H[G];
        uint32_t dup = 0;
        const uint32_t z = opaque_zero();
        coop_round<0>(bl, bhx, j, L[0], H[0], dup, z);
        coop_round<1>(bl, bhx, j, L[1], H[1], dup, z);
        if (G == 4) { coop_round<2 % G>(bl, bhx, j, L[2 % G], H[2 % G], dup, z); coop_round<3 % G>(bl, bhx, j, L[3 % G], H[3 % G], dup, z); }
#pragma unroll
        for (int r = 0; r < G; ++r) H[r] = sel4(((dup >> r) & 1u) != 0u, L[r], H[r]);
        transpose<G>(j, L);
        transpose<G>(j, H);
        uint32_t wl[G * 4], wh[G * 4];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            wl[4 * k] = L[k].x; wl[4 * k + 1] = L[k].y; wl[4 * k + 2] = L[k].z; wl[4 * k + 3] = L[k].w;
            wh[4 * k] = H[k].x; wh[4 * k + 1] = H[k].y; wh[4 * k + 2] = H[k].z; wh[4 * k + 3] = H[k].w;
        }
        block_rank<WPP>(wl, lo - bl * SPB, rl);
        block_rank<WPP>(wh, hi - bh * SPB, rh);
#ifdef GM_COUNTERS
        if (lo | hi) { steps += 1; lines += 1 + (bl != bh); }
#endif
    }
    // the same for 64-bit rows (wide geometry): 64-byte blocks read by groups of FOUR lanes; block numbers take two registers (the
    // direction travels in bit 31 of the upper one: a block number stays below 2^34)
    template <int R> __device__ __forceinline__ void coop_round_wide(uint32_t blLo, uint32_t blHi, uint32_t bhLo, uint32_t bhHiX, uint32_t j, uint4& L, uint4& H, uint32_t& dup, uint32_t z)
    {
        constexpr int BC = R * 0x55;   // quad_perm [R,R,R,R]
        const uint32_t oblLo = dpp<BC>(blLo), oblHi = dpp<BC>(blHi), obhLo = dpp<BC>(bhLo), obhHiX = dpp<BC>(bhHiX);
        const uint64_t obl = (uint64_t)oblHi << 32 | oblLo, obh = (uint64_t)(obhHiX & 0x7FFFFFFFu) << 32 | obhLo;
        const uint4* pb = reinterpret_cast<const uint4*>((obhHiX >> 31) ? A.blk[1] : A.blk[0]);
        L = pb[obl * 4u + j];
        H = make_uint4(z, z, z, z);
        if (obl != obh) H = pb[obh * 4u + j];
        uint32_t d = obl == obh ? 1u : 0u;
        asm volatile("" : "+v"(d));
        dup |= d << R;
    }
    __device__ __forceinline__ void rank2_coop(uint32_t right, uint64_t lo, uint64_t hi, uint64_t rl[NLET], uint64_t rh[NLET])   // 64-bit rows
    {
        constexpr uint32_t SPB = BlockGeom<WPP>::SPB;
        constexpr int G = 4;
        static_assert(BlockGeom<WPP>::BYTES == 64u, "four lanes per 64-byte block");
        const uint32_t j = threadIdx.x & 3u;
        const uint64_t bl = lo / SPB, bh = hi / SPB;
        const uint32_t blLo = (uint32_t)bl, blHi = (uint32_t)(bl >> 32), bhLo = (uint32_t)bh, bhHiX = (uint32_t)(bh >> 32) | right << 31;
        uint4 L[G], H[G];
        uint32_t dup = 0;
        const uint32_t z = opaque_zero();
        coop_round_wide<0>(blLo, blHi, bhLo, bhHiX, j, L[0], H[0], dup, z);
        coop_round_wide<1>(blLo, blHi, bhLo, bhHiX, j, L[1], H[1], dup, z);
        coop_round_wide<2>(blLo, blHi, bhLo, bhHiX, j, L[2], H[2], dup, z);
        coop_round_wide<3>(blLo, blHi, bhLo, bhHiX, j, L[3], H[3], dup, z);
#pragma unroll
        for (int r = 0; r < G; ++r) H[r] = sel4(((dup >> r) & 1u) != 0u, L[r], H[r]);
        transpose<G>(j, L);
        transpose<G>(j, H);
        uint32_t wl[G * 4], wh[G * 4];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            wl[4 * k] = L[k].x; wl[4 * k + 1] = L[k].y; wl[4 * k + 2] = L[k].z; wl[4 * k + 3] = L[k].w;
            wh[4 * k] = H[k].x; wh[4 * k + 1] = H[k].y; wh[4 * k + 2] = H[k].z; wh[4 * k + 3] = H[k].w;
        }
        block_rank<WPP>(wl, (uint32_t)(lo - bl * SPB), rl);
        block_rank<WPP>(wh, (uint32_t)(hi - bh * SPB), rh);
#ifdef GM_COUNTERS
        if (lo | hi) { steps += 1; lines += 1 + (bl != bh); }
#endif
    }
    // (the walker of the split search never stages 2-bit windows: CountEnv<WPP, 2> instantiates this with WIN2_POSSIBLE = false and keeps the loop it was tuned with)
    template <bool WIN2_POSSIBLE = true> __device__ __forceinline__ uint32_t text_char_impl(const Root& rt, uint32_t pos) const
    {
        const uint32_t W = K + rt.n - 1u;
        const uint32_t p = rt.strand ? (W - 1u - pos) : pos;
        uint32_t c;
        if (WIN2_POSSIBLE && A.win2) {   // (wave-uniform) windows at 2 bits per symbol; bit 7 of woff: the window touches a chunk with an N -- the 4-bit text in HBM knows where
            if (woff & 128u) {
                const uint64_t g = A.textBegin + (uint64_t)rt.win + p;
                c = (reinterpret_cast<const uint8_t*>(A.text4)[g >> 1] >> (((uint32_t)g & 1u) * 4u)) & 15u;
            } else {
                const uint32_t s2 = (woff & 63u) + p;
                c = (lwin[(s2 >> 6) * 1024u + ((s2 & 63u) >> 2)] >> ((s2 & 3u) * 2u)) & 3u;
            }
        } else {
            const uint32_t nib = woff + p;
            const uint32_t b = lwin[(nib >> 5) * 1024u + ((nib & 31u) >> 1)];   // chunk stride = 64 lanes x 16 bytes
            c = (b >> ((nib & 1u) * 4u)) & 15u;
        }
        return rt.strand ? complement(c) : c;
    }
    __device__ __forceinline__ uint32_t text_char(const Root& rt, uint32_t pos) const { return text_char_impl<true>(rt, pos); }
    __device__ __forceinline__ void push(const Node& nd)
    {
        const uint32_t lv = sbase + sp;
#ifdef GM_COUNTERS
        pushes++; maxSp = lv + 1u > maxSp ? lv + 1u : maxSp;
#endif
        // wave-uniform fast path (scalar branch, no exec-mask juggling): nobody in the wavefront is past the LDS levels
        if (__ballot(lv >= A.ldsDepth) == 0ull) { IO::store(lstk + (size_t)lv * IO::NU * 64u, 64u, nd); ++sp; return; }
        note_wave(15);
        if (lv < A.ldsDepth) { IO::store(lstk + (size_t)lv * IO::NU * 64u, 64u, nd); ++sp; }
        else if (lv < A.stackDepth) { IO::store(stk + (size_t)(lv - A.ldsDepth) * IO::NU * 64u, 64u, nd); ++sp; }
        else atomicOr(A.errorFlag, 1u);   // never expected: depth = stack_bound(E, stepSize) (+ STEAL_LEVELS with work sharing)
    }
    __device__ __forceinline__ void drain(bool) {}
    __device__ __forceinline__ void on_root() {}
    __device__ __forceinline__ uint32_t root_hits() const { return 0u; }     // travels with stolen work (CountEnv: gate of the saturation check)
    __device__ __forceinline__ void set_root_hits(uint32_t) {}
    __device__ __forceinline__ bool saturated(const Root&, uint32_t, uint32_t) const { return false; }
    __device__ __forceinline__ row_t C(uint32_t c) const { return (row_t)A.C[c]; }
    __device__ __forceinline__ bool any(bool b) const { return __ballot(b) != 0ull; }
    __device__ __forceinline__ row_t sa(row_t row) const { return reinterpret_cast<const row_t*>(A.sa)[row]; }
    // text position of a forward SA row for csv / --exclude-pseudo: one read with the full array; with a sampled one, LF steps
    // (one rank block each: the symbol in front of the suffix and its rank) until a marked row (src/seqan_libdivsufsort.h:129-143)
    __device__ __forceinline__ row_t locate(row_t row) const
    {
#ifdef GM_COUNTERS
        locRows++;
#endif
        if (A.sa) return sa(row);
        constexpr uint32_t SPB = BlockGeom<WPP>::SPB, WPB = BlockGeom<WPP>::WPB, H = BlockGeom<WPP>::HDRW;
        row_t r = row; uint32_t k = 0;
        for (;;) {
            const uint2 m = A.saMark[(size_t)(r >> 5)];
            const uint32_t bit = 1u << ((uint32_t)r & 31u);
            if (m.x & bit) return (row_t)(reinterpret_cast<const row_t*>(A.saSamples)[m.y + (uint32_t)__popc(m.x & (bit - 1u))] + k);
            const uint32_t* blk = A.blk[0] + (size_t)(r / SPB) * WPB;
            const uint32_t off = (uint32_t)(r % SPB), w = off >> 5, t = off & 31u;
            const uint32_t c = ((blk[H + w] >> t) & 1u) | (((blk[H + WPP + w] >> t) & 1u) << 1) | (((blk[H + 2 * WPP + w] >> t) & 1u) << 2);
            // a sentinel in front of the suffix is never reached (sequence starts are sampled) and a walk is shorter than the
            // largest sampling rate: anything else is a damaged index file, which must not hang the device
            if (c >= NLET || k >= 64u) return (row_t)0;
            row_t rk[NLET];
            block_rank<WPP>(blk, off, rk);
            row_t nx = 0;
#pragma unroll
            for (uint32_t i = 0; i < NLET; ++i) nx = c == i ? (row_t)(A.C[i] + rk[i]) : nx;   // selects, not an indexed array (scratch)
            r = nx;
            ++k;
#ifdef GM_COUNTERS
            lfSteps++;
#endif
        }
    }
    struct Item { row_t p0; uint32_t w[7]; };
    __device__ __forceinline__ Item item(row_t row) const
    {
        Item it;
        if (A.ctx) {   // wave-uniform
            const uint4 c0 = A.ctx[(size_t)row * 2], c1 = A.ctx[(size_t)row * 2 + 1];
            it.p0 = c0.x; it.w[0] = c0.y; it.w[1] = c0.z; it.w[2] = c0.w; it.w[3] = c1.x; it.w[4] = c1.y; it.w[5] = c1.z; it.w[6] = c1.w;
        } else {
            it.p0 = sa(row);
#pragma unroll
            for (int k = 0; k < 7; ++k) it.w[k] = 0u;
        }
        return it;
    }
    // fast verification (gm_engine.h): the record and the chunks of the 4-bit text that hold the needle window, requested together
    typedef MaskItemT<row_t> MaskItem;
    template <bool NL> __device__ __forceinline__ MaskItem mask_item(row_t row, uint32_t meta, const Root& rt) const
    {
        const uint64_t g = A.textBegin + rt.win;
        const uint32_t wo = (uint32_t)(g & 31u), W = K + rt.n - 1u;
        const uint4* src = A.text4 + (g >> 5);
        const uint4 c0 = A.ctx[(size_t)row * 2], c1 = A.ctx[(size_t)row * 2 + 1];
        const uint4 n0 = src[0], n1 = src[1];
        uint4 n2 = make_uint4(0u, 0u, 0u, 0u);
        if (wo + W > 64u) n2 = src[2];              // (the text has 20 chunks of padding behind it)
        const uint32_t c[12] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w};
        const uint32_t r[7] = {c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        MaskItem it; it.p0 = (row_t)c0.x;
        fv_masks<NL>(c, wo, W, rt.strand, r, meta_a(meta), it.mm, it.st);
        return it;
    }
    // 8 consecutive symbols of the record starting at symbol index s (0 <= s <= 48), one per byte
    static __device__ __forceinline__ uint64_t ctx8(const Item& it, uint32_t s)
    {
        const uint32_t k = s >> 3;
        // words k and k + 1 (word 7 does not exist: never needed for s <= 48 unless the shift is 0, then its bits are not used)
        const uint32_t a01 = (k & 1u) ? it.w[1] : it.w[0], a23 = (k & 1u) ? it.w[3] : it.w[2], a45 = (k & 1u) ? it.w[5] : it.w[4], a6 = it.w[6];
        const uint32_t b01 = (k & 1u) ? it.w[2] : it.w[1], b23 = (k & 1u) ? it.w[4] : it.w[3], b45 = (k & 1u) ? it.w[6] : it.w[5];
        const uint32_t lo = (k & 4u) ? ((k & 2u) ? a6 : a45) : ((k & 2u) ? a23 : a01);
        const uint32_t hi = (k & 4u) ? ((k & 2u) ? 0u : b45) : ((k & 2u) ? b23 : b01);
        const uint32_t sh = (s & 7u) * 4u;
        const uint32_t v = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
        // 8 nibbles -> 8 bytes
        uint32_t x0 = v & 0xFFFFu, x1 = v >> 16;
        x0 = (x0 | (x0 << 8)) & 0x00FF00FFu; x0 = (x0 | (x0 << 4)) & 0x0F0F0F0Fu;
        x1 = (x1 | (x1 << 8)) & 0x00FF00FFu; x1 = (x1 | (x1 << 4)) & 0x0F0F0F0Fu;
        return (uint64_t)x1 << 32 | x0;
    }
    // eight consecutive bytes starting at p (any alignment): two aligned 64-bit loads and a funnel shift
    static __device__ __forceinline__ uint64_t load8_up(const uint8_t* p)
    {
        const uintptr_t u = reinterpret_cast<uintptr_t>(p);
        // ONE 16-byte load from the 8-aligned address below p (a lane-load is the unit the memory system charges for
        // beyond the TLB reach, profiles/r02/r02a_gather2_*): two 64-bit halves, funnel-shifted
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2), aligned(8)));
        const u64x2 v = *reinterpret_cast<const u64x2*>(u & ~static_cast<uintptr_t>(7));
        const uint32_t sh = (uint32_t)(u & 7u) * 8u;
        const uint64_t lo = v.x, hi = v.y;
        return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
    }
    static __device__ __forceinline__ uint64_t load8_down(const uint8_t* p) { return __builtin_bswap64(load8_up(p - 7)); }   // p[0], p[-1], ..
    static __device__ __forceinline__ uint64_t complement8(uint64_t x)   // A<->T, C<->G, N stays N, per byte
    {
        const uint64_t n4 = x & 0x0404040404040404ull;
        return (x ^ 0x0303030303030303ull) ^ ((n4 >> 1) | (n4 >> 2));
    }
    __device__ __forceinline__ uint64_t needle8(const Root& rt, uint32_t q, bool down) const
    {
        if (!rt.strand) { const uint8_t* p = A.text + (size_t)rt.win + q; return down ? load8_down(p) : load8_up(p); }
        const uint8_t* p = A.text + (size_t)rt.win + (K + rt.n - 2u - q);   // needle(q) = comp(text[win + W - 1 - q])
        return complement8(down ? load8_up(p) : load8_down(p));
    }
    __device__ __forceinline__ uint64_t text8(const Item& it, int32_t off, bool down) const
    {
        if (A.ctx) {   // wave-uniform: the record holds symbols p0 - CTX_LEFT .. p0 - CTX_LEFT + 55
            const int32_t s = off + CTX_LEFT - (down ? 7 : 0);
            if (s >= 0 && s <= CTX_SYMS - 8) {
                const uint64_t v = ctx8(it, (uint32_t)s);
                return down ? __builtin_bswap64(v) : v;
            }
        }
        const uint8_t* p = A.textS + ((long long)it.p0 + off);   // 512 sentinel bytes of padding on both sides
        return down ? load8_down(p) : load8_up(p);
    }
};

// leaf policy 1: frequency only -- hits[a-ab] = min(countOccurrences(it) + hits[a-ab], max) (algo.hpp:48,191)
// MODE 0: the plain tree walk; 1: roots start from their jump patterns (N-less pass); 2: the walker of the split search (N-less pass too: its
// packets come from the jump patterns, enumerated by phase A)
template <int WPP, int MODE = 0> struct CountEnv : EnvBase<WPP> {
    using EnvBase<WPP>::A;
    typedef typename EnvBase<WPP>::row_t row_t;
    typedef typename EnvBase<WPP>::Root Root;
    static constexpr bool NLESS = MODE != 0, JUMPS = MODE == 1, NODES = MODE == 2;   // the tables of the jump hold A,C,G,T strings only
    // Verified hits come as RUNS of k-mers of one block (gm_engine.h: verify_item): with the difference plane a run costs two
    // fire-and-forget atomics whatever its length instead of one returning device-scope atomic per k-mer (3.09 Gbp K=100 e=1: a third
    // of all memory requests of the pass were those atomics, WRITE_SIZE 98x the result, profiles/r03/final/pmc_by_config.txt).
    // That also makes the self hit pay at e >= 1 (one atomic for the whole block instead of n; r03 measured -5..-10 % with n).
    static constexpr bool RANGE_ADD = true, SELF_HIT = true;
    uint32_t leafSum = 0;
    uint32_t rootHits = 0;   // hits this lane has added for its current root (gates the saturation check)
    __device__ __forceinline__ CountEnv(const SearchArgs& a, uint4* s, uint32_t k) : EnvBase<WPP>(a, s, k) {}
    __device__ __forceinline__ uint32_t text_char(const Root& rt, uint32_t pos) const { return this->template text_char_impl<MODE != 2>(rt, pos); }
    __device__ __forceinline__ void on_root() { rootHits = 0; }
    __device__ __forceinline__ uint32_t root_hits() const { return rootHits; }
    __device__ __forceinline__ void set_root_hits(uint32_t v) { rootHits = v; }
    // c[] = min(total, MAX) (algo.hpp:36,48,191): once every k-mer a node still covers has reached MAX, nothing below the
    // node can change the result.  Checked only after this lane alone has produced MAX hits for the root (repeats).
    __device__ __forceinline__ bool saturated(const Root& rt, uint32_t smin, uint32_t smax) const
    {
        // (the walker of the split search sees the nodes of a root one by one, in any lane: what the root has found so far is in acc, not in
        //  this lane's count -- it looks whenever the caller asks, and the callers ask for wide nodes only)
        if (!NODES && rootHits < A.maxVal) return false;
        const row_t lo = rt.win + (rt.strand ? rt.n - 1u - smax : smin);
        const uint32_t cnt = smax - smin + 1u;
        for (uint32_t i = 0; i < cnt; ++i) if (__hip_atomic_load(&A.acc[lo + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < A.maxVal) return false;
        return true;
    }
    __device__ __forceinline__ void leaf(const Root&, uint32_t, row_t, row_t w) { leafSum = (row_t)leafSum + w > 0xFFFFFFFFull ? 0xFFFFFFFFu : leafSum + (uint32_t)w; }
    __device__ __forceinline__ void leaf_flush(const Root& rt, uint32_t kmer)
    {
        const uint32_t count = leafSum; leafSum = 0;
        if (!count) return;
        rootHits = rootHits + count < rootHits ? 0xFFFFFFFFu : rootHits + count;
        const row_t pos = this->slice_pos(rt, kmer);
        add_acc(&A.acc[pos], count < A.addCap ? count : A.addCap);
    }
    // One add into an accumulator.  A RETURNING device-scope atomic is a round trip through the fabric that the whole wavefront waits
    // for (s_waitcnt vmcnt(0) right behind it) -- in the step of nearly every iteration at e >= 1, for the sake of a wrap-around that the
    // host can rule out for all but absurd (K, E): then the add is fire-and-forget.
    __device__ __forceinline__ void add_acc(uint32_t* p, uint32_t add)
    {
        if (A.noWrap) { __hip_atomic_fetch_add(p, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }   // wave-uniform
        const uint32_t old = atomicAdd(p, add);
        if (old > 0xFFFFFFFFu - add) atomicOr(p, 0x80000000u);   // sticky saturation on (theoretical) wrap
    }
    __device__ __forceinline__ void leaf_at(const Root& rt, uint32_t kmer, row_t)
    {
        add_acc(&A.acc[this->slice_pos(rt, kmer)], 1u);
        // (verified hits are not added to rootHits: the verifying lane is not the root's lane)
    }
    // k-mers s0..s1 of the block each gain one occurrence
    __device__ __forceinline__ void leaf_range(const Root& rt, uint32_t s0, uint32_t s1)
    {
        const row_t lo = rt.win + (rt.strand ? rt.n - 1u - s1 : s0), hi = lo + (s1 - s0);   // slice positions of the run
        if (A.diff) {   // wave-uniform
            // the sums are taken modulo 2^32 inside one block: a position's true count is below 2^32 (rows of the index)
            __hip_atomic_fetch_add(&A.diff[lo], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (hi + 1u < rt.win + rt.n) __hip_atomic_fetch_add(&A.diff[hi + 1u], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (row_t p = lo; p <= hi; ++p) add_acc(&A.acc[p], 1u);
        }
    }
};

// leaf policy 1b: frequency for E = 0 -- with a single exact search every (k-mer, strand) pair is reached by at most one
// leaf or one verified row, so the count is written with a plain store into a per-strand plane (8-bit for -fs, else 16-bit) (plane[strand * accPlane +
// pos], counts clamp at 65535) instead of a device-scope atomic (which is a fabric transaction on a multi-XCD part); finalize adds the planes.
template <int WPP, typename TPlane> struct StoreEnv : EnvBase<WPP> {
    using EnvBase<WPP>::A;
    typedef typename EnvBase<WPP>::row_t row_t;
    typedef typename EnvBase<WPP>::Root Root;
    static constexpr bool EXACT_ONLY = true;    // launched for E = 0 only (gm_api.hip: `store`)
    static constexpr bool SELF_HIT = true;
    // Round 4 reordered the loop for the kernels with jump patterns (every load of an iteration issued back to back at its end, the root
    // context fetched in place, the OSS record from LDS in stage 2, argument words picked by selects).  The e = 0 kernel has no pattern
    // reads to line up with and lost 3.3 % to it on one box (51.0 -> 52.7 ms, profiles/r04/final/ab_round_start_vs_commits.txt): it keeps
    // the order of round 3 -- draw in front of the verification, a staged root context, the record read with the draw, indexed argument
    // words (whose vector load parks the wavefront until its window has landed: profiles/r03/README_experiments.txt, item 10).
#ifdef GM_NO_LEGACY_LOOP   // (A/B builds)
    static constexpr bool LEGACY_LOOP = false;
#else
    static constexpr bool LEGACY_LOOP = WPP != 2;
#endif   // (the wide e = 0 kernels sit at the 128-VGPR cap: the staged context would spill)
    static constexpr uint32_t CAP = sizeof(TPlane) == 1 ? 0xFFu : 0xFFFFu;   // min(MAX, f + r) == min(MAX, min(MAX, f) + min(MAX, r))
    __device__ __forceinline__ StoreEnv(const SearchArgs& a, uint4* s, uint32_t k) : EnvBase<WPP>(a, s, k) {}
    __device__ __forceinline__ void leaf(const Root& rt, uint32_t kmer, row_t, row_t w)
    {
        reinterpret_cast<TPlane*>(A.acc)[(size_t)rt.strand * A.accPlane + this->slice_pos(rt, kmer)] = (TPlane)(w < CAP ? w : CAP);
    }
    __device__ __forceinline__ void leaf_flush(const Root&, uint32_t) {}
    __device__ __forceinline__ void leaf_at(const Root& rt, uint32_t kmer, row_t)
    {
        reinterpret_cast<TPlane*>(A.acc)[(size_t)rt.strand * A.accPlane + this->slice_pos(rt, kmer)] = (TPlane)1u;
    }
};

// Leaves of the locating policies (getOccurrences of every matching range, algo.hpp:328-345) are not walked by the lane that
// found them -- a repeat leaf of 500 rows would hold the other 63 lanes for 500 dependent reads -- but queued per wavefront in LDS
// as {first row, rows, target}; when enough have gathered, the whole wavefront expands the queue row by row: a prefix sum over the
// entries' widths, every lane finds the entry its row belongs to and locates that one row (consecutive rows of a leaf are
// consecutive suffix-array entries: coalesced reads), then Derived::row_action(target, index inside the leaf, seqNo, seqPos).
constexpr uint32_t LQ_DRAIN = 48;         // entries that trigger a drain (the queue holds SearchArgs::lqCap)
constexpr uint32_t LQ_PIECE = 1u << 24;   // a leaf wider than this is queued in pieces: 64 entries' rows fit 32 bits
template <int WPP, class Derived> struct LeafQueueEnv : EnvBase<WPP> {
    using EnvBase<WPP>::A;
    typedef typename EnvBase<WPP>::row_t row_t;
    static constexpr bool LEAFQ = true;
    uint4* lq = nullptr;          // LDS: entries {row lo, rows, target lo, target hi (56 bits) | row bits 32..39 << 24}
    uint32_t* lqCtl = nullptr;    // LDS: [0] entries pushed, [1..64] scratch of the expansion (exclusive prefix sums)
    __device__ __forceinline__ LeafQueueEnv(const SearchArgs& a, uint4* s, uint32_t k) : EnvBase<WPP>(a, s, k) {}
    // one row of a queued leaf: by default "locate it, then Derived::row_action(target, index inside the leaf, (seqNo, seqPos))"
    __device__ __forceinline__ void row_located(uint64_t target, uint32_t r, row_t row)
    {
        static_cast<Derived*>(this)->row_action(target, r, locate_position(A.cumGlobal, A.nSeqGlobal, this->locate(row)));
    }
    // called by one lane (any control flow): returns false when the queue is full -- the caller then walks the leaf itself
    __device__ __forceinline__ bool enqueue(row_t flo, uint32_t w, uint64_t target)
    {
        if (A.lqCap == 0u) return false;   // (gm_longk.h: no queue at all -- no counter to bump either)
        const uint32_t slot = atomicAdd(&lqCtl[0], 1u);
        if (slot >= A.lqCap) return false;
        lq[slot] = make_uint4((uint32_t)flo, w, (uint32_t)target, (uint32_t)(target >> 32) | (uint32_t)((uint64_t)flo >> 32) << 24);
        return true;
    }
    // called by the whole wavefront at a uniform point of the loop
    __device__ __forceinline__ void drain(bool finishing)
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t pushed = __hip_atomic_load(&lqCtl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)pushed);
        if (cnt == 0u || (!finishing && cnt < LQ_DRAIN)) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t n = cnt < A.lqCap ? cnt : A.lqCap;
        for (uint32_t base = 0; base < n; base += 64u) {
            uint4 mine = make_uint4(0, 0, 0, 0);
            if (base + lane < n) mine = lq[base + lane];
            uint32_t incl = mine.y;   // inclusive prefix sum of the widths over the lanes
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, (int)d); if (lane >= d) incl += up; }
            const uint32_t total = (uint32_t)__shfl((int)incl, 63);
            lqCtl[1 + lane] = incl - mine.y;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t g0 = 0; g0 < total; g0 += 64u) {
                const uint32_t g = g0 + lane;
                if (g < total) {
                    uint32_t i = 0;   // the last entry whose first row is <= g (entries without rows share their successor's start)
#pragma unroll
                    for (uint32_t step = 32u; step >= 1u; step >>= 1) if (lqCtl[1 + (i | step)] <= g) i |= step;
                    const uint4 e = lq[base + i];
                    const uint32_t r = g - lqCtl[1 + i];
                    const row_t row = (row_t)(((uint64_t)(e.w >> 24) << 32 | e.x) + r);
                    static_cast<Derived*>(this)->row_located((uint64_t)(e.w & 0xFFFFFFu) << 32 | e.z, r, row);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0u) lqCtl[0] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};

// leaf policy 2: --exclude-pseudo -- the set of fasta files that contain the k-mer (algo.hpp:351-364)
template <int WPP> struct FileSetEnv : LeafQueueEnv<WPP, FileSetEnv<WPP>> {
    using EnvBase<WPP>::A;
    typedef typename EnvBase<WPP>::row_t row_t;
    typedef typename EnvBase<WPP>::Root Root;
    __device__ __forceinline__ FileSetEnv(const SearchArgs& a, uint4* s, uint32_t k) : LeafQueueEnv<WPP, FileSetEnv<WPP>>(a, s, k) {}
    // with the per-row file ids the set needs no position at all: one byte per row (rows of a leaf are consecutive: coalesced)
    __device__ __forceinline__ void row_located(uint64_t pos, uint32_t r, row_t row)
    {
        if (A.rowFile) {   // wave-uniform
#ifdef GM_COUNTERS
            this->locRows++;
#endif
            add_file(pos, A.rowFile[row]);
        } else row_action(pos, r, locate_position(A.cumGlobal, A.nSeqGlobal, this->locate(row)));
    }
    __device__ __forceinline__ void add_file(uint64_t pos, uint32_t f)
    {
        uint32_t* word = &A.fileBits[(size_t)pos * A.wordsPerKmer + (f >> 5)];
        if (!((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (f & 31u)) & 1u)) atomicOr(word, 1u << (f & 31u));   // mostly set already
    }
    // target = the k-mer's slice position
    __device__ __forceinline__ void row_action(uint64_t pos, uint32_t, uint2 sp)
    {
        const uint32_t f = A.seqFile[sp.x];
        uint32_t* word = &A.fileBits[(size_t)pos * A.wordsPerKmer + (f >> 5)];
        if (!((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (f & 31u)) & 1u)) atomicOr(word, 1u << (f & 31u));   // mostly set already
    }
    __device__ __forceinline__ void leaf(const Root& rt, uint32_t kmer, row_t flo, row_t w)
    {
        const uint64_t pos = this->slice_pos(rt, kmer);
        while (w > 0) {
            const uint32_t piece = w < (row_t)LQ_PIECE ? (uint32_t)w : LQ_PIECE;
            if (!this->enqueue(flo, piece, pos))   // queue full: this lane walks the piece itself
                for (uint32_t r = 0; r < piece; ++r) row_located(pos, r, flo + r);
            flo += piece; w -= piece;
        }
    }
    __device__ __forceinline__ void leaf_flush(const Root&, uint32_t) {}
    __device__ __forceinline__ void leaf_at(const Root& rt, uint32_t kmer, row_t textPos)
    {
        const uint32_t f = A.seqFile[locate_position(A.cumGlobal, A.nSeqGlobal, textPos).x];
        atomicOr(&A.fileBits[(size_t)this->slice_pos(rt, kmer) * A.wordsPerKmer + (f >> 5)], 1u << (f & 31u));
    }
};

// leaf policy 3: csv pass 1 -- occurrences per (k-mer, strand)
template <int WPP> struct OccCountEnv : EnvBase<WPP> {
    using EnvBase<WPP>::A;
    typedef typename EnvBase<WPP>::row_t row_t;
    typedef typename EnvBase<WPP>::Root Root;
    __device__ __forceinline__ OccCountEnv(const SearchArgs& a, uint4* s, uint32_t k) : EnvBase<WPP>(a, s, k) {}
    __device__ __forceinline__ void leaf(const Root& rt, uint32_t kmer, row_t, row_t w)
    {
        atomicAdd(&A.cnt2[(size_t)rt.strand * A.windowLen + (this->slice_pos(rt, kmer) - A.posBase)], (uint32_t)w);   // windows hold < 2^31 occurrences
    }
    __device__ __forceinline__ void leaf_flush(const Root&, uint32_t) {}
    __device__ __forceinline__ void leaf_at(const Root& rt, uint32_t kmer, row_t) { leaf(rt, kmer, 0u, 1u); }
};

// leaf policy 4: csv pass 2 -- getOccurrences(iterator) of every leaf (algo.hpp:328-345), unsorted
template <int WPP> struct OccEmitEnv : LeafQueueEnv<WPP, OccEmitEnv<WPP>> {
    using EnvBase<WPP>::A;
    typedef typename EnvBase<WPP>::row_t row_t;
    typedef typename EnvBase<WPP>::Root Root;
    __device__ __forceinline__ OccEmitEnv(const SearchArgs& a, uint4* s, uint32_t k) : LeafQueueEnv<WPP, OccEmitEnv<WPP>>(a, s, k) {}
    // target = the leaf's first slot of the emit array
    __device__ __forceinline__ void row_action(uint64_t base, uint32_t r, uint2 sp) { A.emit[base + r] = (uint64_t)sp.x << 32 | sp.y; }
    __device__ __forceinline__ void leaf(const Root& rt, uint32_t kmer, row_t flo, row_t w)
    {
        const size_t slot = (size_t)rt.strand * A.windowLen + (this->slice_pos(rt, kmer) - A.posBase);
        uint64_t base = A.offs[slot] + atomicAdd(&A.cnt2[slot], (uint32_t)w);
        while (w > 0) {
            const uint32_t piece = w < (row_t)LQ_PIECE ? (uint32_t)w : LQ_PIECE;
            if (!this->enqueue(flo, piece, base))
                for (uint32_t r = 0; r < piece; ++r) row_action(base, r, locate_position(A.cumGlobal, A.nSeqGlobal, this->locate(flo + r)));
            flo += piece; w -= piece; base += piece;
        }
    }
    __device__ __forceinline__ void leaf_flush(const Root&, uint32_t) {}
    __device__ __forceinline__ void leaf_at(const Root& rt, uint32_t kmer, row_t textPos)
    {
        const size_t slot = (size_t)rt.strand * A.windowLen + (this->slice_pos(rt, kmer) - A.posBase);
        const uint2 sp = locate_position(A.cumGlobal, A.nSeqGlobal, textPos);
        A.emit[A.offs[slot] + atomicAdd(&A.cnt2[slot], 1u)] = (uint64_t)sp.x << 32 | sp.y;
    }
};

// leaf policy 5: the correction pass of an N-less frequency call.  The needles are the text windows that hold N (of the WHOLE
// index); an occurrence found at text position p means: the k-mer AT p matches that window, i.e. p's own frequency misses this
// hit (Hamming distance is symmetric, and N never equals anything on either side: algo.hpp:111-112, find2:250).  So every
// located occurrence inside the slice adds one to its own position -- if the main pass computes that position at all.
// Leaves go through the wavefront's leaf queue (a needle inside a poly-A run matches millions of rows).  All rows of a leaf hold
// the SAME k-mer, hence the same frequency: when the main pass alone has already brought one of them to MAX, the result of every
// one of them is MAX whatever is added, and the leaf is dropped.
template <int WPP> struct ScatterEnv : LeafQueueEnv<WPP, ScatterEnv<WPP>> {
    using EnvBase<WPP>::A;
    typedef typename EnvBase<WPP>::row_t row_t;
    typedef typename EnvBase<WPP>::Root Root;
    __device__ __forceinline__ ScatterEnv(const SearchArgs& a, uint4* s, uint32_t k) : LeafQueueEnv<WPP, ScatterEnv<WPP>>(a, s, k) {}
    // slice position of (seqNo, seqPos) if this call computes it, else ~0
    __device__ __forceinline__ uint64_t own_position(uint2 sp) const
    {
        const uint64_t g = A.cumGlobal[sp.x] + sp.y;
        if (g < A.sliceBegin || g - A.sliceBegin >= A.sliceLen) return ~0ull;
        const uint64_t q = g - A.sliceBegin;
        if (q < A.ownBegin || q >= A.ownEnd) return ~0ull;
        if (A.ownChunkLen && ((q - A.ownBegin) / A.ownChunkLen) % A.chunkStride != A.chunkIndex) return ~0ull;
        if (A.selBlocks) {
            uint32_t lo = 0, hi = A.nSelBlocks;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; const uint2 e = A.selBlocks[mid]; if (((uint64_t)(e.y >> 8) << 32 | e.x) <= q) lo = mid; else hi = mid; }
            const uint2 e = A.selBlocks[lo];
            const uint64_t b = (uint64_t)(e.y >> 8) << 32 | e.x;
            if (q < b || q >= b + (e.y & 0xFFu)) return ~0ull;
        }
        return q;
    }
    __device__ __forceinline__ void row_action(uint64_t, uint32_t, uint2 sp)
    {
        const uint64_t q = own_position(sp);
        if (q == ~0ull) return;
        uint32_t* p = &A.acc[q];
        if (A.noWrap) { __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }   // (CountEnv::add_acc)
        if (atomicAdd(p, 1u) == 0xFFFFFFFFu) atomicOr(p, 0x80000000u);
    }
    // INVARIANT the two shortcuts below rest on: the main pass computes a position's count as a pure function of the k-mer AT that
    // position (no duplicate shortcut, no per-position early exit other than "the count has reached MAX"), and acc only ever grows.  So
    // all rows of a leaf (same k-mer) end at the same value, acc[q] >= MAX of one owned row means MAX for all of them, and a count that
    // the difference plane still holds back (CountEnv::leaf_range) only makes the test conservative.  With no_saturate the caller
    // passes maxVal = 2^32 - 1 and neither shortcut fires.  (test_gpu_correction_pass_near_the_8_bit_maximum_under_chunks_and_selections)
    __device__ __forceinline__ void leaf(const Root&, uint32_t, row_t flo, row_t w)
    {
        if (w >= (row_t)A.maxVal) {
            // The k-mer X of these rows occurs w times.  If X holds no N, the main pass has counted at least its w exact occurrences
            // for every one of the rows: they are at MAX whether this call owns them or not -- nothing to add.  (An X with N is
            // counted by this pass alone and has to be walked.)
            const uint2 sp = locate_position(A.cumGlobal, A.nSeqGlobal, this->locate(flo));
            const uint8_t* x = A.text + (A.cumGlobal[sp.x] + sp.y);   // A.text: the whole text in this pass
            bool hasN = false;
            for (uint32_t i = 0; i < this->K; i += 8u) {
                uint64_t m = 0x8080808080808080ull & ~bytes_nonzero(EnvBase<WPP>::load8_up(x + i) ^ 0x0404040404040404ull);
                if (this->K - i < 8u) m &= (1ull << (8u * (this->K - i))) - 1ull;
                hasN |= m != 0ull;
            }
            if (!hasN) return;
        }
        if (w >= 16u) {   // wide: is the k-mer of these rows at MAX already?  (the first few rows that this call computes decide)
            for (uint32_t r = 0; r < 8u && r < w; ++r) {
                const uint64_t q = own_position(locate_position(A.cumGlobal, A.nSeqGlobal, this->locate(flo + r)));
                if (q == ~0ull) continue;
                if (__hip_atomic_load(&A.acc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= A.maxVal) return;
                break;
            }
        }
        while (w > 0) {
            const uint32_t piece = w < (row_t)LQ_PIECE ? (uint32_t)w : LQ_PIECE;
            if (!this->enqueue(flo, piece, 0ull))
                for (uint32_t r = 0; r < piece; ++r) row_action(0ull, r, locate_position(A.cumGlobal, A.nSeqGlobal, this->locate(flo + r)));
            flo += piece; w -= piece;
        }
    }
    __device__ __forceinline__ void leaf_flush(const Root&, uint32_t) {}
    __device__ __forceinline__ void leaf_at(const Root&, uint32_t, row_t textPos) { row_action(0ull, 0u, locate_position(A.cumGlobal, A.nSeqGlobal, textPos)); }
};

// the per-call records of the jump patterns as a block keeps them in LDS (search_body, expand_kernel): the layouts ride in the 4th words of
// the searches' group ends, the masks sit behind the jump records
struct LdsTab {
    const uint4* jl;
    __device__ __forceinline__ uint32_t layout(uint32_t il) const { return jl[12u + il].w >> 8; }
    __device__ __forceinline__ unsigned long long mask(uint32_t id) const
    {
        const uint2 mk = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint32_t*>(jl + 8) + 2u * id);
        return (unsigned long long)mk.y << 32 | mk.x;
    }
};
// jl[0..8): jump records of the searches, jl[8..12): masks of the groups, jl[12..20): group ends + layouts, jl[20..28): OSS records of the regular block shape
__device__ __forceinline__ void load_jump_records(uint4* jl, const SearchArgs& A)
{
    if (threadIdx.x < 8u) jl[threadIdx.x] = A.jumpJ ? A.jinfo[threadIdx.x] : make_uint4(0, 0, 0, 0);
    if (threadIdx.x < GROUP_MAX_MASKS) reinterpret_cast<unsigned long long*>(jl + 8)[threadIdx.x] = A.gmask[threadIdx.x];
    // entry L additionally carries layout L in the upper bits of its 4th word (bit offset of the group's characters: 5 bits, kind-0 plane: 8,
    // first kind-1 plane: 8) -- a table of their own would have cost the block 96 bytes of LDS it does not have
    if (threadIdx.x < 8u) {
        uint4 v = A.jumpJ ? A.jinfo2[threadIdx.x] : make_uint4(0, 0, 0, 0);
        v.w = (v.w & 0xFFu) | (A.layShift[threadIdx.x] | A.layPlane0[threadIdx.x] << 5 | A.layPlane1[threadIdx.x] << 13) << 8;
        jl[12u + threadIdx.x] = v;
    }
}

#ifdef GM_WAVES
#define GM_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(GM_WAVES, GM_WAVES)))
#else
#define GM_WAVES_ATTR
#endif
template <int WPP, class EnvT, bool COOP>
__device__ __forceinline__ void search_body(const SearchArgs& A)
{
    typedef typename BlockGeom<WPP>::row_t row_t;
    typedef NodeT<row_t> Node;
    typedef RootT<row_t> Root;
    typedef NodeIO<row_t> IO;
    constexpr uint32_t NU = IO::NU;   // 16-byte units per stored node / queue entry
    const uint32_t lane = threadIdx.x & 63u;
    const size_t gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // spilled stack levels: wavefront-interleaved (level i of the 64 lanes is one contiguous KiB), like the LDS levels
    EnvT env(A, A.stack + (gl & ~(size_t)63) * A.spillDepth * NU + lane, A.K);
    Node nd; nd.flo = nd.rlo = nd.w = 0; nd.meta = 0;
    Root rt; rt.win = 0; rt.n = 1; rt.strand = 0; rt.search = 0; rt.rec = OssRecord{0, 0, 0, 0};
    // per-wavefront queue of narrow nodes awaiting verification: filled by ballot rank, drained 64 at a time so that a
    // verification round keeps every lane busy with the same kind of loop
    extern __shared__ uint4 smem[];
    const uint32_t wv = threadIdx.x >> 6;
    // LDS per block of 4 wavefronts: [4 x vqCap x NU] queue | [4 x ldsDepth x NU x 64] stack tops | [4 x winChunks x 64] windows | users/pairing
    uint4* vq = smem + wv * A.vqCap * NU;
    env.lstk = smem + 4u * A.vqCap * NU + wv * (A.ldsDepth * NU * 64u) + lane;
    uint4* const wbase = smem + 4u * A.vqCap * NU + 4u * A.ldsDepth * NU * 64u + wv * (A.winChunks * 64u);   // this wavefront's windows
    env.lwin = reinterpret_cast<const uint8_t*>(wbase + lane);
    // work sharing inside the wavefront: a lane may work on the root of another lane (stolen stack entries); it then reads
    // that lane's needle window (wlane) and the owner may not stage a new window while users[owner] != 0
    uint32_t* const users = reinterpret_cast<uint32_t*>(smem + 4u * A.vqCap * NU + 4u * 64u * (A.ldsDepth * NU + A.winChunks)) + wv * 80u;   // [64] users (words) | [64] pairing (bytes): 320 B per wavefront
    uint8_t* const pairing = reinterpret_cast<uint8_t*>(users + 64);
    // the searches' jump records (Env::JUMPS), 8 x 16 bytes (+ 64 bytes of group masks + 8 x 16 bytes of group ends + 8 x 16 bytes of OSS records) per block behind the work-sharing bookkeeping
    uint4* const jl = smem + 4u * A.vqCap * NU + 4u * 64u * (A.ldsDepth * NU + A.winChunks) + (4u * 80u * 4u) / 16u;
    if (threadIdx.x < 8u) jl[20u + threadIdx.x] = A.table[(size_t)(A.stepSize - 1u) * 8u + threadIdx.x];   // OSS records of the regular block shape (stage 2)
    if constexpr (!EnvT::JUMPS) __syncthreads();
    if constexpr (EnvT::JUMPS) {
        load_jump_records(jl, A);   // ... the masks of its groups of patterns (gm_oss.h), and where its groups end + the layouts
        __syncthreads();
    }
    // jump patterns: the table entry in flight lives in LDS, one 16-byte slot per lane (global_load_lds: no destination registers, hence
    // no wait behind the load to move them, and 4 VGPRs less) -- 4 KB per block behind the jump records
    uint4* const ebufW = jl + 28 + wv * 64u;
    if constexpr (EnvT::LEAFQ) {   // leaf queue behind everything else: [4 x lqCap entries] [4 x 80 control words]
        uint4* const lqBase = jl + 28;
        env.lq = lqBase + wv * A.lqCap;
        env.lqCtl = reinterpret_cast<uint32_t*>(lqBase + 4u * A.lqCap) + wv * 80u;
        if (lane == 0u) env.lqCtl[0] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    uint4* const lstkW = smem + 4u * A.vqCap * NU + wv * (A.ldsDepth * NU * 64u);
    uint4* const stkW = A.stack + (gl & ~(size_t)63) * A.spillDepth * NU;
    uint32_t wlane = lane;
    if (A.steal) { users[lane] = 0u; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#ifdef GM_COUNTERS
    uint32_t nSteals = 0;
#endif
    uint32_t qsize = 0;                             // wave-uniform
#ifdef GM_COUNTERS
    uint32_t wvIter = 0, wvActive = 0, wvRounds = 0;
#endif
    bool have = false, exhausted = false;
    uint32_t w1run = 0;                             // consecutive steps this lane has taken on a single-row node
    // root fetch pipeline of this lane: 0 idle, 1 window/record loads in flight, 2 q-mer table lookup in flight
    uint32_t fs = 0, fa0 = 0, fql = 0, fnch = 0;
    // The root a lane fetches IS its next root: a lane draws only when it has no node, no stack and no fetch in flight, nobody takes work
    // from such a lane (work sharing robs lanes that hold a node) and it cannot become a thief while its fetch is in flight -- so the
    // fetch writes the root context in place (8 VGPRs less than a staged copy).
    Root frtOwn; frtOwn.win = 0; frtOwn.n = 1; frtOwn.strand = 0; frtOwn.search = 0; frtOwn.rec = OssRecord{0, 0, 0, 0};   // (LEGACY_LOOP: a staged copy)
    Root& frt = *(EnvT::LEGACY_LOOP ? &frtOwn : &rt);
    row_t ftFlo = 0, ftRlo = 0, ftW = 0;
    const uint4* fsrc = A.text4;
    // jump patterns of the lane's current root (Env::JUMPS): 2-bit packed J-mer of the needle, cursor | end << 16 into A.patterns,
    // the next pattern's descriptor (prefetched), meta of the node at depth J (with the errors of the pattern whose table entry
    // is in flight; bit 30 marks the first pattern of a root: the root context is installed with it)
    uint32_t jb = 0, jpp = 0, jd = 0, jm = 0;   // jpp: next item | end << 16; jd: the current item (JF_ITEM); jm: errs field = errors of the pattern in flight
    // groups of patterns (gm_oss.h): rotations of the current group that exist and are still to be looked up, the group's own rotations,
    // the bitmap word in flight (JF_WORD).  fs of a lane with pattern work = 2 | JF_* flags.
    unsigned long long galive = 0ull, pw = 0ull;
    uint32_t gcur = 0;
    // neighbour filter: jn = the needle's characters next to the J-mer inside the infix, packed like the table's 4th word (bit 15: the
    // filter applies to this root), ftNb = that word of the entry in flight
    uint32_t jn = 0, ftNb = 0;
    unsigned long long poolCur = 0, poolEnd = 0, poolBase = 0, poolBlock = 0;   // wave-uniform
    uint32_t poolRem = 0;
    bool globalDone = false;                        // wave-uniform

#ifdef GM_COUNTERS
    unsigned long long tFetch = 0, tVerify = 0, tStep = 0, tPop = 0, tShare = 0, tSt32 = 0, tSt1 = 0, tMark = __builtin_amdgcn_s_memtime();
#define GM_LAP2(acc) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tMark; tFetch += now_ - tMark; tMark = now_; } while (0)
#define GM_LAP(acc) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tMark; tMark = now_; } while (0)
#else
#define GM_LAP(acc) do { } while (0)
#define GM_LAP2(acc) do { } while (0)
#endif
    // (opaque scalar copies: the compiler otherwise turns "select one of two argument words" into a vector load from the argument segment,
    //  which is then waited for behind the pattern reads issued in front of the root draw)
    uint32_t jap0 = A.jumpAPacked[0], jap1 = A.jumpAPacked[1];
    asm volatile("" : "+s"(jap0), "+s"(jap1));
    // the walker of the split search (Env::NODES): what phase A left for this slice (wave-uniform), and the header of the packet in flight
    uint32_t xt0 = 0, xt1 = 0, xwork = 0, xfail = 0, xstamp = 0;
    uint4 ph0 = make_uint4(0, 0, 0, 0), ph1 = make_uint4(0, 0, 0, 0);
    uint32_t satA = 0, satB = 0; bool satPend = false;   // the saturation test of a freshly drawn packet: two counters in flight
    uint32_t xstripe = (blockIdx.x * 4u + (threadIdx.x >> 6)) % XSTRIPES, xtried = 0;   // wave-uniform: the stripe this wavefront draws from, stripes it has found empty
    if constexpr (EnvT::NODES) {
        const ExpandCtl* xc = A.xctl;
        xt0 = xc->t[0]; xt1 = xc->t[1]; xwork = xt0 + xt1 + xc->t[2]; xfail = xc->failFrom; xstamp = xc->stamp;
    }
    // bound of a hung loop (wave-uniform, scalar): consecutive iterations without a node or a verification round (tests: all iterations).
    // The fetch state machine below (parts A / B) is the code that once spun forever on the device (round 4): a wavefront that spins
    // gives up, raises the sticky error flag and the host reports GM_ERR_INTERNAL instead of waiting for ever.
    uint32_t itGuard = 0;
    for (;;) {
        if constexpr (EnvT::NODES) {
            if (satPend) {   // the packet drawn two iterations ago has taken one step; its block's counters have arrived
                satPend = false;
                if (satA >= A.maxVal && satB >= A.maxVal && env.saturated(rt, 0u, rt.n - 1u)) { have = false; env.sp = 0u; env.sbase = 0u; }   // (its stack held this packet's nodes only: a lane draws with an empty stack)
            }
        }
#ifndef GM_POP_LOOP
        if (!have && env.sp > 0) {   // one pop per iteration: a node dropped as saturated costs the lane one idle turn
#else
#pragma unroll 1
        for (int tries = 0; tries < 4 && !have && env.sp > 0; ++tries) {
#endif
            nd = env.pop(); w1run = 0;
            env.note_wave(0);
            have = true;
            if (nd.w >= A.satMinW) {   // pending work for k-mers that already reached MAX is dropped
                env.note_wave(1);
                uint32_t smin, smax;
                covered_kmers(nd.meta, rt.n, A.K, smin, smax);
                have = !env.saturated(rt, smin, smax);
            }
        }
        GM_LAP2(tPop);
        // ---- work sharing inside the wavefront: idle lanes take the bottom of the stack of lanes that have pending nodes ----
        if (A.steal) {
            if (wlane != lane && !have && env.sp == 0u) {   // the borrowed root is finished: give its window back
                atomicSub(&users[wlane], 1u);
                wlane = lane; env.lwin = reinterpret_cast<const uint8_t*>(wbase + lane);
            }
            const bool idle = !have && env.sp == 0u && fs == 0u;
            const bool rich = have && env.sp >= 1u && env.sbase < STEAL_LEVELS;
            const unsigned long long im = __ballot(idle), vm = __ballot(rich);
            if ((uint32_t)__popcll(im) >= A.steal && vm != 0ull) {
                env.note_wave(2);
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
                // the victim's stack height, root and window (every lane takes part: the victims' registers are the source)
                const uint32_t vsb = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)env.sbase);   // the BOTTOM entry: the oldest, i.e. the largest pending subtree
                uint32_t vwin = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)(uint32_t)rt.win), vwinHi = 0u;
                if (sizeof(row_t) == 8) vwinHi = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)(uint32_t)((uint64_t)rt.win >> 32));
                const uint32_t vnss = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)(rt.n | rt.strand << 9 | rt.search << 10));
                const uint32_t vrx = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)rt.rec.x), vry = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)rt.rec.y);
                const uint32_t vrz = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)rt.rec.z), vrw = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)rt.rec.w);
                const uint32_t vwo = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)(env.woff | wlane << 8));
                const uint32_t vrh = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)env.root_hits());
                if (thief) {
                    const uint32_t level = vsb;
                    nd = level < A.ldsDepth ? IO::load(lstkW + (size_t)level * NU * 64u + src, 64u) : IO::load(stkW + (size_t)(level - A.ldsDepth) * NU * 64u + src, 64u);
                    have = true; w1run = 0;
                    rt.win = (row_t)((uint64_t)vwinHi << 32 | vwin); rt.n = vnss & 0x1FFu; rt.strand = (vnss >> 9) & 1u; rt.search = vnss >> 10;
                    rt.rec.x = vrx; rt.rec.y = vry; rt.rec.z = vrz; rt.rec.w = vrw;
                    env.set_root_hits(vrh);   // a root that saturates its k-mers keeps doing so in the thief's hands
                    env.woff = vwo & 0xFFu; wlane = vwo >> 8;
                    env.lwin = reinterpret_cast<const uint8_t*>(wbase + wlane);
                    if (wlane != lane) atomicAdd(&users[wlane], 1u);   // (stealing back work of one's own root needs no reference)
#ifdef GM_COUNTERS
                    nSteals++;
#endif
                }
                if (robbed) { env.sp -= 1u; env.sbase = env.sp ? env.sbase + 1u : 0u; }
            }
        }
        GM_LAP2(tShare);
        // ---- root fetch, pipelined over iterations so that the wavefront never waits for it ----
        // stage 3: the q-mer table entry has arrived -> the root becomes the lane's node (or turns out empty)
        // Pattern turns in batches (knob pat_batch): parts A and B are ~190 VALU instructions that the whole wavefront walks through for the
        // one or two lanes that need them; with a batch they run when enough idle lanes wait for their turn, or when nobody holds a node.
        // The state is level-triggered (an arrived entry stays in its slot, an item in jd): a turn that comes later changes nothing else.
        bool patTurn = true;
        if constexpr (EnvT::JUMPS) {
            if (A.patBatch > 1u) {
                const unsigned long long pm = __ballot((fs & 3u) == 2u && !have && env.sp == 0u);
                patTurn = (uint32_t)__popcll(pm) >= A.patBatch || __ballot(have) == 0ull;
            }
        }
        if constexpr (EnvT::JUMPS) {
            // (A) the table entry of the pattern in flight has arrived, and the lane has finished the subtree of the one before
            if (patTurn && (fs & (3u | JF_ENTRY)) == (2u | JF_ENTRY) && !have && env.sp == 0u) {
                env.note_wave(3);
                fs &= ~JF_ENTRY;
                row_t eFlo, eRlo, eW; uint32_t eNb;
                if constexpr (sizeof(row_t) == 4) {   // the entry went from HBM straight into this lane's LDS slot (part B): awaited explicitly
                    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
                    const uint4 ft = ebufW[lane];
                    eFlo = ft.x; eRlo = ft.y; eW = ft.z; eNb = ft.w;
                } else { eFlo = ftFlo; eRlo = ftRlo; eW = ftW; eNb = ftNb; }
                // every k-mer of the block at MAX: nothing a further pattern finds can change the result
                const bool sat = env.root_hits() >= A.maxVal && env.saturated(rt, 0u, rt.n - 1u);
                bool take = eW != 0u && !sat;
                if (take && eW == 1u && (jn & eNb & 0x8000u) != 0u) {
                    // The substituted J-mer occurs once.  Whatever this node could still find lies at that one place and contains the whole
                    // infix, so the infix characters next to the J-mer must agree with the text there up to the errors the pattern has
                    // left; a text N or a sequence end within them ends it too (N-less pass).  No memory request is spent on the rest.
                    const uint32_t h = jl[rt.search].w;     // bits 2i / 16 + 2i: neighbour i counts; bits 12..14 / 28..30: how many do
                    const uint32_t x = eNb ^ jn;
                    const uint32_t differ = (x | x >> 1) & h & 0x05550555u;
                    take = ((eNb >> 12) & 7u) >= ((h >> 12) & 7u) && ((eNb >> 28) & 7u) >= ((h >> 28) & 7u) && meta_errs(jm) + (uint32_t)__popc(differ) <= A.E;
#ifdef GM_COUNTERS
                    env.jumpDrops += take ? 0u : 1u;
#endif
                }
                bool rowsOnly = false;
                if (take && eW == 2u && (jn & 0x8000u) != 0u && A.nbFilter == 1u) {
                    // The substituted J-mer occurs TWICE: the same test with 3 + 3 neighbours for either row.  Neither passes: no node.  One
                    // passes: the node is that row alone, and a lone row of which only the forward position is known is never stepped --
                    // it goes to the verification queue (rows-only node, rlo = all ones).
                    const uint32_t h = jl[rt.search].w;
                    const uint32_t needR = min((h >> 12) & 7u, NB_SYMS2), needL = min((h >> 28) & 7u, NB_SYMS2);
                    const uint32_t mR = h & 0x15u, mL = (h >> 16) & 0x15u, budget = A.E - meta_errs(jm);
                    bool pass[2];
#pragma unroll
                    for (uint32_t r = 0; r < 2u; ++r) {
                        const uint32_t x = (eNb >> (16u * r)) & 0xFFFFu;
                        const uint32_t dr = (x ^ jn) & 0x3Fu, dl = ((x >> 6) ^ (jn >> 16)) & 0x3Fu;
                        const uint32_t mism = (uint32_t)__popc((dr | dr >> 1) & mR) + (uint32_t)__popc((dl | dl >> 1) & mL);
                        pass[r] = ((x >> 12) & 3u) >= needR && (x >> 14) >= needL && mism <= budget;
                    }
                    take = pass[0] || pass[1];
                    if (take && pass[0] != pass[1] && A.verifyT != 0u) { rowsOnly = true; eFlo += pass[1] ? 1u : 0u; }
#ifdef GM_COUNTERS
                    env.jumpDrops2 += take ? (rowsOnly ? 1u : 0u) : 2u;
#endif
                }
                if (take) {
                    nd.flo = eFlo; nd.rlo = eRlo; nd.w = eW; nd.meta = jm;
                    if (rowsOnly) { nd.rlo = ~(row_t)0; nd.w = 1u; }
                    have = true; w1run = 0;
                }
                if (sat) { galive = 0ull; jpp = (jpp >> 16) * 0x10001u; fs = 2u; }   // the remaining items are skipped (part B ends the root)
            }
        } else if (fs == 2u) {
            env.note_wave(3);
            fs = 0u;
            bool take = ftW != 0u;
            if (ftW == 1u && (ftNb >> 31) != 0u && A.nbFilter) {
                // The q-mer occurs once, and the rest of the search's first block must follow it without an error (u[0] = 0 in every
                // scheme, gm_oss.h; a needle N, a text N and a sequence end all end an exact block): compare up to 6 of those characters
                // with the text next to that occurrence, which the table entry carries -- a chance hit of the reverse strand ends here
                // instead of after a rank step or a record read.  (The forward strand's own location passes, of course.)
                uint32_t R = oss_bl(frt.rec, 0u) - fql;
                R = R < NB_SYMS ? R : NB_SYMS;
                take = ((ftNb >> 12) & 7u) >= R;
#pragma unroll 1
                for (uint32_t i = 0; i < R && take; ++i) take = env.text_char(frt, fa0 + fql + i) == ((ftNb >> (2u * i)) & 3u);
            }
            if (take) {
                rt = frt; env.on_root();
                nd.flo = ftFlo; nd.rlo = ftRlo; nd.w = ftW; nd.meta = meta_pack(fa0, fa0 + fql, 0, 0, M_OSS);
                have = true; w1run = 0;
            }
        }
        // the walker of the split search: the packet drawn in the last iteration has arrived -> the lane holds its node
        if constexpr (EnvT::NODES) {
            if (fs == 1u) {
                env.note_wave(4);
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): header registers and the window's LDS slots (nothing else orders the LDS DMA)
                fs = 0u;
                // a slot phase A did not fill in this slice carries an older stamp; the packets of work that did not fit are redone by the next slice
                if (ph1.w == xstamp && ph1.z < xfail) {
                    nd.flo = ph0.x; nd.rlo = ph0.y; nd.w = ph0.z; nd.meta = ph0.w;
                    rt.win = ph1.x; rt.n = ph1.y & 0xFFu; rt.strand = (ph1.y >> 8) & 1u; rt.search = (ph1.y >> 9) & 7u;
                    uint4 q;
                    if (rt.n == A.stepSize) q = jl[20u + rt.search]; else q = A.table[(size_t)(rt.n - 1u) * 8u + rt.search];
                    rt.rec.x = q.x; rt.rec.y = q.y; rt.rec.z = q.z; rt.rec.w = q.w;
                    env.woff = 0u; env.on_root();
                    have = true; w1run = 0;
                    // Every k-mer of the block at MAX already (the lists come in the order of the patterns' substitutions: by the time a pattern
                    // with errors is drawn, what the exact pattern of its root has found is in the accumulators)?  Then nothing below can change
                    // the result.  The block's first and last counter are requested here and looked at in the NEXT iteration (the step below waits
                    // for its rank blocks anyway): a blocking test stalled the wavefront for a memory round trip per drawn packet -- K=30 e=1 2.1x slower.
                    if (nd.w >= A.satDrawW) {
                        satA = __hip_atomic_load(&A.acc[rt.win], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        satB = __hip_atomic_load(&A.acc[rt.win + rt.n - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        satPend = true;
                    }
                }
            }
        }
        // stage 2: window chunks and record have arrived -> stage the window in LDS, look the first q characters up
        if (!EnvT::NODES && fs == 1u) {
            env.note_wave(4);
            if (A.win2) {   // does the window touch a chunk of the 2-bit text that holds an N?  (bit 7 of the window offset: text_char)
                const uint64_t ch = (A.textBegin + (uint64_t)frt.win) >> 6;
                const uint32_t nch2 = (env.woff + A.K + frt.n - 1u + 63u) >> 6;
                if ((ftNb >> ((uint32_t)ch & 7u)) & ((1u << nch2) - 1u)) env.woff |= 128u;
            }
            if constexpr (!EnvT::LEGACY_LOOP) {   // the OSS record of the root's search: from LDS for the regular block shape, from the table for the odd ones (ends of the text / of an interval)
                uint4 q;
                if (frt.n == A.stepSize) q = jl[20u + frt.search]; else q = A.table[(size_t)(frt.n - 1u) * 8u + frt.search];
                frt.rec.x = q.x; frt.rec.y = q.y; frt.rec.z = q.z; frt.rec.w = q.w;
            }
            if (fql == 0u) { rt = frt; env.on_root(); nd = root_node(rt, (row_t)A.nRows); have = true; fs = 0u; w1run = 0; }
            else {
                // 16 symbols starting at the lowest text position of the q-mer, 4 bits each: from the window this lane staged in
                // LDS in stage 1 (the asynchronous global -> LDS loads are awaited explicitly: nothing else orders them)
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
                uint32_t bad, lo2;
                if (A.win2) {
                    // 2 bits per symbol: the q-mer's 32 bits straight from the window; an N (stored as A) shows in the 4-bit text only -- rare roots, the slow path of text_char
                    const uint32_t p0 = frt.strand ? (A.K + frt.n - 1u) - fa0 - fql : fa0, ni = (env.woff & 63u) + p0;
                    const uint8_t* wc = reinterpret_cast<const uint8_t*>(wbase + lane) + (ni >> 6) * 1024u;   // chunk stride: 64 lanes x 16 bytes
                    const uint4 c0 = *reinterpret_cast<const uint4*>(wc);
                    const uint32_t c1 = *reinterpret_cast<const uint32_t*>(wc + 1024u);
                    const uint32_t wi = (ni & 63u) >> 4, sh = (ni & 15u) * 2u;
                    const uint32_t vl = wi == 0u ? c0.x : wi == 1u ? c0.y : wi == 2u ? c0.z : c0.w, vh = wi == 0u ? c0.y : wi == 1u ? c0.z : wi == 2u ? c0.w : c1;
                    const uint32_t v = sh ? (vl >> sh) | (vh << (32u - sh)) : vl;
                    lo2 = fql >= 16u ? v : (v & ((1u << (2u * fql)) - 1u));
                    bad = 0u;
                    if (env.woff & 128u) {
                        const uint64_t g0 = A.textBegin + (uint64_t)frt.win + p0;
                        for (uint32_t i = 0; i < fql; ++i) bad |= (reinterpret_cast<const uint8_t*>(A.text4)[(g0 + i) >> 1] >> ((((uint32_t)g0 + i) & 1u) * 4u)) & 4u;
                    }
                } else {
                const uint32_t ni = env.woff + (frt.strand ? (A.K + frt.n - 1u) - fa0 - fql : fa0);
                const uint8_t* wc = reinterpret_cast<const uint8_t*>(wbase + lane) + (ni >> 5) * 1024u;   // chunk stride: 64 lanes x 16 bytes
                const uint4 c0 = *reinterpret_cast<const uint4*>(wc);
                const uint2 c1 = *reinterpret_cast<const uint2*>(wc + 1024u);
                const uint32_t bo = (ni & 31u) * 4u, b2 = bo & 63u;
                const unsigned long long w0 = (unsigned long long)c0.y << 32 | c0.x, w1 = (unsigned long long)c0.w << 32 | c0.z, w2 = (unsigned long long)c1.y << 32 | c1.x;
                const unsigned long long vl = bo < 64u ? w0 : w1, vh = bo < 64u ? w1 : w2;
                const unsigned long long v = b2 ? (vl >> b2) | (vh << (64u - b2)) : vl;
                // the table index without a loop over the symbols: nibbles -> 2-bit symbols (symbol i at bits 2i); a code above 3
                // (N) anywhere in the q-mer makes the root empty
                const unsigned long long qmask = fql >= 16u ? ~0ull : ((1ull << (4u * fql)) - 1ull);
                bad = (v & qmask & 0xCCCCCCCCCCCCCCCCull) != 0ull ? 1u : 0u;
                unsigned long long t = v & qmask & 0x3333333333333333ull;
                t = (t | (t >> 2)) & 0x0F0F0F0F0F0F0F0Full;
                t = (t | (t >> 4)) & 0x00FF00FF00FF00FFull;
                t = (t | (t >> 8)) & 0x0000FFFF0000FFFFull;
                t = (t | (t >> 16)) & 0x00000000FFFFFFFFull;
                lo2 = (uint32_t)t;                                   // sum of c_i << 2i
                }
                const uint32_t m2 = fql >= 16u ? 0xFFFFFFFFu : ((1u << (2u * fql)) - 1u);
                // reverse strand: needle(a0 + q-1-i) = 3 - c_i, i.e. the complement of every 2-bit group, same order
                // forward strand: symbol i is needle(a0 + i), most significant first: reverse the order of the groups
                uint32_t r = __builtin_bitreverse32(lo2);
                r = ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
                const uint32_t idx = frt.strand ? (~lo2 & m2) : (fql ? r >> (32u - 2u * fql) : 0u);
                if constexpr (EnvT::JUMPS) {
                    // N inside the J-mer (it may lie in a block that allows errors): this root walks the tree from its root
                    if (bad) { rt = frt; env.on_root(); nd = root_node(rt, (row_t)A.nRows); have = true; fs = 0u; w1run = 0; }
                    else {
                        jb = idx;
                        const uint4 fji = jl[frt.search];   // {first item | items << 16, meta at depth J relative to n - 1, first item, neighbour-filter mask}
                        jm = meta_pack((fji.y & 0x1FFu) + frt.n - 1u, ((fji.y >> 9) & 0x1FFu) + frt.n - 1u, fji.y >> 18, 0u, M_OSS);
                        jn = 0u;
                        if (fji.w >> 31) {   // the needle's neighbours of the J-mer, once per root
                            uint32_t notLetter = 0u;
#pragma unroll 1
                            for (uint32_t i = 0; i < ((fji.w >> 12) & 7u); ++i) { const uint32_t c = env.text_char(frt, fa0 + A.jumpJ + i); notLetter |= c >> 2; jn |= (c & 3u) << (2u * i); }
#pragma unroll 1
                            for (uint32_t i = 0; i < ((fji.w >> 28) & 7u); ++i) { const uint32_t c = env.text_char(frt, fa0 - 1u - i); notLetter |= c >> 2; jn |= (c & 3u) << (16u + 2u * i); }
                            jn = notLetter ? 0u : (jn | 0x8000u);   // a needle N mismatches everything: leave such roots to the ordinary path
                        }
                        // the first item travels with the search's record; part (B) below issues its read in this very iteration
                        jd = fji.z; galive = 0ull;
                        jpp = ((fji.x & 0xFFFFu) + 1u) | ((fji.x & 0xFFFFu) + (fji.x >> 16)) << 16;
                        env.on_root();
                        {
                            const uint4 lim = jl[12u + frt.search];
                            fs = 2u | jump_item_flags(fji.x & 0xFFFFu, lim.x, lim.y, lim.z);
                            if (lim.w & 0xFFu) {   // groups of kind 1 ask whether the J-mer occurs followed by the needle's next two letters
                                const uint32_t e0 = env.text_char(frt, fa0 + A.jumpJ), e1 = env.text_char(frt, fa0 + A.jumpJ + 1u);
                                if ((e0 | e1) < SYM_N) fs |= JF_EXTOK | (e0 << 2 | e1) << JF_EXT_SHIFT;
                            }
                        }
                    }
                } else {
                if (bad) fs = 0u;   // a pattern N never matches in an exact block (find2:330): this root finds nothing
                else { IO::load_qentry(((A.qselMask >> frt.search) & 1u) ? A.qtabB : A.qtabA, idx, ftFlo, ftRlo, ftW, ftNb); fs = 2u; }
                }
            }
        }
        // ---- self hits, BEFORE the draw below (and before part B issues its loads: the window reads here wait for every load in flight): a lane whose root ends here (most roots of a large e = 0 call: the table leaves them one
        // row) takes its next root in this very iteration (3.09 Gbp K=30 e=0: 51.6 against 54.1 ms) ----
        if constexpr (EnvT::SELF_HIT) {
            // Forward strand, no error spent, ONE row left: it is the window's own location (a string always matches itself), so every
            // k-mer the node still covers gains exactly one occurrence -- without reading the suffix array, a record or another rank
            // block.  Nothing else can be found below the node: the text there IS the needle, no mismatching child exists.  In the OSS
            // phase the self hit belongs to the search whose remaining lower bounds are all zero (find2:389-392): the others drop the
            // node.  (k-mers that cross a sequence end are zeroed by resetLimits whatever is added here; a window with an N anywhere
            // takes the ordinary path, which knows which k-mers the N spoils.)
            if (A.selfHit && have && nd.w == 1u && rt.strand == 0u && (EnvT::EXACT_ONLY || (meta_errs(nd.meta) == 0u && nd.rlo != ~(row_t)0))) {   // (not the left-over rows of a wider node; e = 0 has neither errors nor such nodes)
                const bool w2 = !EnvT::NODES && A.win2;
                const uint32_t W = A.K + rt.n - 1u, nch = w2 ? 0u : (env.woff + W + 31u) >> 5;
                uint32_t anyN = w2 ? (env.woff & 128u) : 0u;   // (2-bit windows: the chunk flags of stage 2)
                for (uint32_t c = 0; c < nch; ++c) {
                    const uint4 v = *reinterpret_cast<const uint4*>(env.lwin + c * 1024u);
                    anyN |= (v.x | v.y | v.z | v.w) & 0x44444444u;   // (nibbles of the neighbouring text in the first / last chunk count too: harmless)
                }
                if (anyN == 0u) {
                    uint32_t smin, smax;
#ifdef GM_COUNTERS
                    env.selfHits++;
#endif
                    if (self_hit_kmers(nd.meta, rt, A.K, smin, smax)) {   // gm_engine.h
                        if constexpr (EnvT::RANGE_ADD) env.leaf_range(rt, smin, smax);
                        else for (uint32_t k = smin; k <= smax; ++k) env.leaf_at(rt, k, (row_t)0);
                    }
                    have = false;
                }
            }
        }
        GM_LAP2(tSt32);
        if constexpr (EnvT::LEGACY_LOOP) {
            bool windowFree = !A.steal || __hip_atomic_load(&users[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == 0u;
#include "gm_stage1.inc"
        }
        // ---- defer narrow nodes: one queue entry per SA row ----
        if (A.verifyT) {
            const uint32_t md0 = meta_mode(nd.meta);
            // (extension-phase nodes have the whole infix behind them: one row there costs a record read and two short scans, while the
            //  walk still has ~n log n steps to go -- they may be wider than the nodes verified inside the infix)
            bool narrow = have && nd.w <= ((EnvT::EXACT_ONLY || md0 == M_OSS) ? A.verifyT : A.verifyTExt);
            if (narrow && (EnvT::EXACT_ONLY || nd.rlo != ~(row_t)0)) {   // is the subtree below worth one SA read + one text comparison per row?
                const uint32_t m = nd.meta, a = meta_a(m), bx = meta_bx(m), t = meta_t(m), md = md0;
                const uint32_t covered = md == M_OSS ? rt.n : md == M_EXT_R ? a + A.K - t + 1u : md == M_EXT_L ? t + A.K - bx + 1u : a + A.K - bx + 1u;
                const uint32_t est = (A.K - (bx - a)) + covered - 1u;   // lower bound of the steps still needed
                narrow = nd.w * A.verifyCost <= est;
                // A lone row that may not mismatch any more is, more often than not, a chance hit that the next one or
                // two characters kill with ONE rank line each (lo and hi share a block); verification costs an SA read
                // plus a text read.  Step it a little first, verify only the survivors.
                if (!EnvT::EXACT_ONLY && nd.w == 1u && meta_errs(m) == A.E && w1run < A.probation) narrow = false;
                // e = 0: the table leaves one infix character; taking it first costs one rank line and spares the
                // reverse-strand chance hits their SA + text reads (5.31 vs 5.46 ms, profiles/r01e_infix_sweeps.txt)
                if ((EnvT::EXACT_ONLY || A.E == 0u) && md == M_OSS) narrow = false;
            }
            // At most VERIFY_ROWS rows of a node are queued per iteration (SearchArgs::verifyRows; the queue holds 64 + 64 * verifyRows entries); the rows left
            // over go back onto the lane's stack as a rows-only node (rlo = all ones: never stepped, queued the moment it is popped).
            // (the queue holds 64 + 96 entries with two rows per lane: a second row that would not fit -- more than 96 - qsize lanes with a
            //  second row to queue -- waits on the stacks like the rows beyond the second; 2 KB of LDS per block are a stack level at K = 30)
            uint32_t rowsDone = 0;   // wave-uniform
#pragma unroll 1
            for (uint32_t r = 0; r < (EnvT::EXACT_ONLY ? 1u : A.verifyRows); ++r) {   // (e = 0 verifies single rows only: the plain stores rely on it)
                const bool e = narrow && r < nd.w;
                const unsigned long long m = __ballot(e);
                if (m == 0ull) break;
                if (qsize + (uint32_t)__popcll(m) > A.vqCap) break;
                rowsDone = r + 1u;
                env.note_wave(6);
                if (e) {
                    const uint32_t slot = qsize + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    IO::store_item(vq + (size_t)slot * NU, nd.flo + r, nd.meta, rt.win, rt.n | rt.strand << 8 | rt.search << 9);
                }
                qsize += (uint32_t)__popcll(m);
            }
            if (narrow) {
                have = false;
                if (nd.w > rowsDone) { nd.flo += rowsDone; nd.w -= rowsDone; nd.rlo = ~(row_t)0; env.push(nd); }
            }
            // a partial round only when the wavefront has nothing else left to do
            const bool finishing = (__ballot(have || fs != 0u || env.sp != 0u) == 0ull) && (__ballot(!exhausted) == 0ull);
            while (qsize >= 64u || (finishing && qsize > 0u)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef GM_COUNTERS
                wvRounds += 1;
#endif
                const uint32_t take = qsize < 64u ? qsize : 64u;
                if (lane < take) {
                    row_t irow, iwin; uint32_t imeta, inss;
                    IO::load_item(vq + (size_t)(qsize - 1u - lane) * NU, irow, imeta, iwin, inss);
                    Root vr; vr.win = iwin; vr.n = inss & 0xFFu; vr.strand = (inss >> 8) & 1u; vr.search = inss >> 9;
                    // the OSS record of the item's search: from LDS for the regular block shape (read AFTER the masks of a fast item are
                    // built: four registers fewer while they are), from the table for the odd shapes
                    auto load_rec = [&]() {
                        uint4 q;
                        if (vr.n == A.stepSize) q = jl[20u + vr.search]; else q = A.table[(size_t)(vr.n - 1u) * 8u + vr.search];
                        vr.rec.x = q.x; vr.rec.y = q.y; vr.rec.z = q.z; vr.rec.w = q.w;
                    };
                    if constexpr (sizeof(row_t) == 4 && !EnvT::EXACT_ONLY) {   // (e = 0 verifies one row in a hundred k-mers: not worth its registers there)
                        if (A.fastVerify) {   // wave-uniform
                            const typename EnvT::MaskItem mi = env.template mask_item<EnvT::NLESS>(irow, imeta, vr);
                            load_rec();
                            verify_with(mi, imeta, vr, A.K, A.E, env);
                        } else { load_rec(); verify_item(irow, imeta, vr, A.K, A.E, env); }
                    } else { load_rec(); verify_item(irow, imeta, vr, A.K, A.E, env); }
                }
                qsize -= take;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        GM_LAP(tVerify);
        env.drain(false);   // locating policies: queued leaves, once enough have gathered
        // a wavefront without a single node (and nothing queued) skips the step, but still draws roots and reads patterns below
        const bool stepWave = __ballot(have) != 0ull || qsize != 0u;
        if (!stepWave && __ballot(!exhausted || fs != 0u || env.sp != 0u) == 0ull) { env.drain(true); break; }   // nothing in flight, nothing queued or stacked, nothing left to draw
        if (stepWave) itGuard &= A.guardKeep;
        if (++itGuard > A.guardCap) { if (lane == 0u) atomicOr(A.errorFlag, 2u); break; }
        if constexpr (EnvT::LEGACY_LOOP) if (!stepWave) continue;   // (the draw has run already)

#ifdef GM_COUNTERS
        if (stepWave) { wvIter += 1; wvActive += (uint32_t)__popcll(__ballot(have)); }
#endif
        if (have && meta_mode(nd.meta) == M_SPLIT) {
            env.note_wave(7);
            Node left; split_node(nd, left, A.K);
            bool leftDone = false, rightDone = false;
            if (nd.w >= A.satMinW) {   // (both halves have the parent's width)
                uint32_t smin, smax;
                covered_kmers(left.meta, rt.n, A.K, smin, smax);
                leftDone = env.saturated(rt, smin, smax);
                covered_kmers(nd.meta, rt.n, A.K, smin, smax);
                rightDone = env.saturated(rt, smin, smax);
            }
            if (rightDone) { if (leftDone) have = false; else nd = left; }
            else if (!leftDone) env.push(left);
        }
        GM_LAP(tStep);
        // ---- every load of the iteration is issued from here on, back to back: the windows of new roots (LDS DMA), the pattern reads
        // (an LDS DMA too), then the rank blocks of the step.  Any LDS access behind an LDS DMA waits for ALL loads in flight (the compiler
        // cannot tell the addresses apart), so nothing else may stand between them: one memory round trip per iteration, not three ----
        // (nobody else reads this lane's window: checked here, in front of the loads -- see above)
        bool windowFree = !A.steal || __hip_atomic_load(&users[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == 0u;
        if constexpr (EnvT::JUMPS) {
            // (B) the next table read of the lane's root, as soon as the previous entry has been consumed.  Items (gm_oss.h) are plain
            // patterns -- substituted J-mer -> ONE 16-byte read of the table of all J-mers -- or GROUPS of 64 patterns that differ in three
            // adjacent characters only: one 8-byte word of a bitmap tells which of them can match at all, and only those are looked up.  A
            // group's word is requested as soon as its item is known, i.e. while the lane still works through the group before it.
            // ONE load site each for the next item, the word and the table entry: with a second site the compiler loads into temporaries
            // and waits for the load right behind it to move the value over.  The LDS reads come first, the loads last (see above).
            if (patTurn && (fs & 3u) == 2u) {
                env.note_wave(16);
                const uint4 lim = jl[12u + rt.search];   // where the groups of each layout end among the search's items
                const JumpStep D = jump_decide(fs, jd, gcur, galive, pw, jb, LdsTab{jl});   // gm_oss.h: the decisions; the loads stay here, ONE site each
                const bool want = D.want, asked = D.asked, go = D.go;
                const uint32_t widx = D.widx, wsel = D.wsel, rw = D.rw;
                if (want) {   // the next item of the lane's root (not looked at before the next iteration: nothing waits for that load)
                    const uint32_t jp = jpp & 0xFFFFu;
                    if (jp < (jpp >> 16)) {
                        jd = A.patterns[jp]; jpp += 1u;
                        fs |= jump_item_flags(jp, lim.x, lim.y, lim.z);
                    }
                }
                if (asked) {
                    pw = A.jbits[(size_t)wsel * A.jbitsWords + widx];
                    fs |= JF_WORD;
#ifdef GM_COUNTERS
                    env.jumpWords++;
#endif
                }
                if (go) {
                    if constexpr (sizeof(row_t) == 4) __builtin_amdgcn_global_load_lds(A.jtab + rot_add(jb, rw), ebufW, 16, 0, 0);
                    else IO::load_qentry(A.jtab, rot_add(jb, rw), ftFlo, ftRlo, ftW, ftNb);
                    jm = (jm & ~(7u << META_ERRS_SHIFT)) | rot_errors(rw) << META_ERRS_SHIFT;
                    fs |= JF_ENTRY;
#ifdef GM_COUNTERS
                    env.jumps++;
#endif
                }
                if (jump_done(fs, galive)) fs = 0u;   // nothing in flight, no item, nothing alive: the root's patterns are done
            }
        }
        GM_LAP2(tSt32);
        if constexpr (EnvT::NODES) {
#include "gm_stage1n.inc"
        } else if constexpr (!EnvT::LEGACY_LOOP) {
#include "gm_stage1.inc"
        }
        GM_LAP2(tSt1);
        if constexpr (!COOP) if (have) {
            const bool lone = nd.w == 1u;
            lane_step(nd, have, rt, A.K, A.E, env);
            w1run = (lone && have && nd.w == 1u) ? w1run + 1u : 0u;
        }
        if constexpr (COOP) if (stepWave) {   // the rank blocks are read by groups of lanes: every lane walks through the reads
            Plan pl; pl.right = pl.exact = pl.minErr = pl.charsLeft = pl.pos = 0;
            row_t plo = 0, phi = 0;
            if (have) {
                pl = make_plan(nd.meta, rt.rec, A.E);
                env.note_step(meta_mode(nd.meta), nd.w);
                plo = pl.right ? nd.rlo : nd.flo; phi = plo + nd.w;
            }
            row_t rl[NLET], rh[NLET];
            env.rank2_coop(pl.right, plo, phi, rl, rh);
            if (have) {
                const bool lone = nd.w == 1u;
                lane_children(nd, have, rt, A.K, A.E, env, pl, rl, rh);
                w1run = (lone && have && nd.w == 1u) ? w1run + 1u : 0u;
            }
        }
        GM_LAP(tStep);
    }
#ifdef GM_COUNTERS
    atomicAdd(&A.counters[0], (unsigned long long)env.steps);
    atomicAdd(&A.counters[1], (unsigned long long)env.lines);
    atomicAdd(&A.counters[2], (unsigned long long)env.stOss);
    atomicAdd(&A.counters[3], (unsigned long long)env.stExt);
    atomicAdd(&A.counters[4], (unsigned long long)env.stExtW1);
    atomicAdd(&A.counters[5], (unsigned long long)env.stExtW4);
    atomicAdd(&A.counters[6], (unsigned long long)env.stOssW1);
    atomicAdd(&A.counters[7], (unsigned long long)env.pushes);
    atomicAdd(&A.counters[8], (unsigned long long)env.vItems);
    atomicAdd(&A.counters[9], (unsigned long long)env.vItemsOss);
    atomicAdd(&A.counters[10], (unsigned long long)env.vChunks);
    atomicAdd(&A.counters[21], (unsigned long long)nSteals);
    atomicAdd(&A.counters[38], (unsigned long long)env.jumps);   // detail[36]: table reads of jump patterns
    atomicAdd(&A.counters[41], (unsigned long long)env.jumpDrops);   // detail[39]: one-row entries ended by the neighbour filter
    atomicAdd(&A.counters[42], (unsigned long long)env.locRows);     // detail[40]: rows located (locating policies, correction pass)
    atomicAdd(&A.counters[43], (unsigned long long)env.lfSteps);     // detail[41]: LF steps of sampled suffix-array walks
    atomicMax(&A.counters[44], (unsigned long long)env.maxSp);       // detail[42]: deepest lane stack of the call
    atomicAdd(&A.counters[45], (unsigned long long)env.selfHits);    // detail[43]: self hits (nodes settled without a lookup)
    atomicAdd(&A.counters[46], (unsigned long long)env.runs);        // detail[44]: verified runs of k-mers
    atomicAdd(&A.counters[47], (unsigned long long)env.jumpWords);   // detail[45]: bitmap words read for groups of jump patterns
    atomicAdd(&A.counters[48], (unsigned long long)env.jumpDrops2);  // detail[46]: rows of two-row table entries ended by the neighbour filter
#pragma unroll
    for (int i = 0; i < 16; ++i) if (env.whit[i]) atomicAdd(&A.counters[22 + i], (unsigned long long)env.whit[i]);
    if (lane == 0) {
        atomicAdd(&A.counters[11], (unsigned long long)wvIter);
        atomicAdd(&A.counters[12], (unsigned long long)wvActive);
        atomicAdd(&A.counters[13], (unsigned long long)wvRounds);
        atomicAdd(&A.counters[14], tFetch);
        atomicAdd(&A.counters[15], tVerify);
        atomicAdd(&A.counters[16], tStep);
        atomicAdd(&A.counters[17], tPop); atomicAdd(&A.counters[18], tShare); atomicAdd(&A.counters[19], tSt32); atomicAdd(&A.counters[20], tSt1);
    }
#endif
}

template <int WPP, class EnvT, bool COOP>
__global__ __launch_bounds__(256) GM_WAVES_ATTR void search_kernel(const SearchArgs A) { search_body<WPP, EnvT, COOP>(A); }
// the same kernel compiled for 4 waves per SIMD (at most 128 VGPRs): the cooperative 64-byte variant needs a few registers more
// than that by itself and loses a wave of occupancy otherwise
#ifndef GM_COUNTERS
template <int WPP, class EnvT, bool COOP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void search_kernel_w4(const SearchArgs A) { search_body<WPP, EnvT, COOP>(A); }
#else   // the instrumented twin keeps ~20 counters in registers: capped at 128 VGPRs it spills and its cycle attribution is skewed
template <int WPP, class EnvT, bool COOP>
__global__ __launch_bounds__(256) void search_kernel_w4(const SearchArgs A) { search_body<WPP, EnvT, COOP>(A); }
#endif

// ---- phase A of the split search (gm_expand.h) ------------------------------------------------------------------------------------------
// slice_begin: which chunks of the call this slice takes.  The number adapts to what the last slice produced (packets per chunk, X and Y
// separately), aiming at three quarters of the buffers: a slice that overflows loses nothing (its failed chunks are redone), but it has
// walked a shorter list than it could have.
__global__ void expand_slice_begin_kernel(ExpandProgress* prog, ExpandCtl* ctl, uint32_t usableX, uint32_t usableY, uint32_t firstChunks, uint32_t maxChunks)
{
    if (threadIdx.x != 0u || blockIdx.x != 0u) return;
    const unsigned long long begin = prog->committed, total = prog->totalChunks;
    unsigned long long r = firstChunks;
    if (prog->lastChunks) {
        const unsigned long long rx = (unsigned long long)usableX * prog->lastChunks / (prog->lastX ? prog->lastX : 1u);
        const unsigned long long ry = (unsigned long long)usableY * prog->lastChunks / (prog->lastY ? prog->lastY : 1u);
        r = (rx < ry ? rx : ry) * 3ull / 4ull;
    }
    if (r < 1ull) r = 1ull;
    if (r > maxChunks) r = maxChunks;
    ctl->chunkBegin = begin; ctl->chunkEnd = begin + r < total ? begin + r : total;
    ctl->nextChunk = begin; ctl->tailsX = 0ull; ctl->tailY = 0u; ctl->failFrom = 0xFFFFFFFFu; ctl->reserved0 = 0ull;
    for (uint32_t j = 0; j < XSTRIPES; ++j) ctl->walk[16u * j] = 0ull;
    ctl->valid[0] = ctl->valid[1] = ctl->valid[2] = 0u; ctl->t[0] = ctl->t[1] = ctl->t[2] = 0u;
    ctl->stamp = ++prog->stampCounter;
}
// slice_commit: phase A of the slice has ended -> what the walker draws, where the next slice starts
__global__ void expand_slice_commit_kernel(ExpandProgress* prog, ExpandCtl* ctl, uint32_t* errorFlag)
{
    if (threadIdx.x != 0u || blockIdx.x != 0u) return;
    const unsigned long long begin = ctl->chunkBegin, end = ctl->chunkEnd;
    const unsigned long long done = (unsigned long long)ctl->failFrom < end ? (unsigned long long)ctl->failFrom : end;
    ctl->t[0] = ctl->valid[0]; ctl->t[1] = ctl->valid[1]; ctl->t[2] = ctl->valid[2];
    if (done <= begin && end > begin) atomicOr(errorFlag, 4u);   // not even one chunk fits the buffers: the host sized them (never expected)
    prog->committed = done > begin ? done : begin;
    prog->slices += 1u;
    if (done > begin) { prog->lastChunks = (uint32_t)(done - begin); prog->lastX = ctl->valid[0] + ctl->valid[1]; prog->lastY = ctl->valid[2]; }
}

// One lane per (k-mer block, strand, search, item).  Blocks come in chunks of A.expandBlocks from a counter; a wavefront walks the work
// items of its chunk 64 at a time: root context from the 4-bit text (expand_root), the item's bitmap word, then -- all lanes in step -- one
// table entry per lane and turn until the items have run out of surviving rotations; what passes the neighbour filters is appended to the
// class lists.  No stacks, no LDS windows, no loop-carried memory state: the kernel runs at whatever occupancy its registers allow.
__global__ __launch_bounds__(256) void expand_kernel(const SearchArgs A)
{
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2), aligned(8)));
    struct DevMem {
        const unsigned long long* t;
        __device__ __forceinline__ void pair(uint64_t i, uint64_t& lo, uint64_t& hi) const { const u64x2 v = *reinterpret_cast<const u64x2*>(t + i); lo = v.x; hi = v.y; }
    } mem{reinterpret_cast<const unsigned long long*>(A.text4)};
    __shared__ uint4 jl[28];
    __shared__ uint4 cx0s[256], cx1s[256];   // per lane of a batch: {J-mer index, neighbours, meta at depth J, filter mask}, {window origin, root word, the item's own rotations, layout shift | flags}
    __shared__ uint2 cx2s[256];              // ... the item's surviving rotations
    __shared__ uint32_t pres[256];           // ... rotations of the lanes before it
    // root contexts of a group of blocks (A.xshare): a root's J-mer, neighbours and window are the same for every one of its ~9 items -- one lane per ROOT
    // computes them for the 10 blocks (K=30 e=2: 60 roots) whose ~560 items the next nine turns of the wavefront work through
    __shared__ uint4 rc0s[256];              // {J-mer index, neighbours, the two letters behind, window origin}
    __shared__ uint32_t rc1s[256];           // n | bad << 8 | block at MAX << 9
    __shared__ uint4 wwins[4 * 16 * 3];      // the packet window (from nibble 0) of every block of the group, up to three chunks
    load_jump_records(jl, A);
    if (threadIdx.x < 8u) jl[20u + threadIdx.x] = A.table[(size_t)(A.stepSize - 1u) * 8u + threadIdx.x];
    __syncthreads();
    const LdsTab tab{jl};
    const uint32_t lane = threadIdx.x & 63u;
    uint4* const cx0 = cx0s + (threadIdx.x & ~63u); uint4* const cx1 = cx1s + (threadIdx.x & ~63u);
    uint2* const cx2 = cx2s + (threadIdx.x & ~63u); uint32_t* const pre = pres + (threadIdx.x & ~63u);
    uint4* const rc0 = rc0s + (threadIdx.x & ~63u); uint32_t* const rc1 = rc1s + (threadIdx.x & ~63u); uint4* const wwin = wwins + (threadIdx.x >> 6) * 48u;
    const uint32_t RPB = A.rootsPerBlockA;
    const uint32_t NB = (A.xshare && RPB >= 1u && RPB <= 64u && A.pktChunks <= 3u) ? (64u / RPB < 16u ? 64u / RPB : 16u) : 0u;   // blocks per group (0: every item computes its root's context itself)
    ExpandCtl* const ctl = A.xctl;
    const uint32_t U = PKT_HEADER_UNITS + A.pktChunks;
    const uint32_t G = A.expandBlocks, IPB = A.itemsPerBlock;
    const unsigned long long chunkEnd = ctl->chunkEnd;
    const uint32_t stamp = ctl->stamp;
    uint32_t rb[3] = {0u, 0u, 0u}, ru[3] = {XREGION, XREGION, XREGION};   // wave-uniform: the wavefront's open region per class, packets used in it
    bool dead = false;                                                      // wave-uniform: a reservation did not fit
#ifdef GM_COUNTERS
    uint32_t cJumps = 0, cWords = 0, cDrops = 0, cDrops2 = 0, cPackets = 0;
#endif
    for (;;) {
        unsigned long long cid = 0ull;
        if (lane == 0u) cid = atomicAdd(&ctl->nextChunk, 1ull);
        cid = __shfl(cid, 0);
        if (cid >= chunkEnd || cid >= (unsigned long long)__hip_atomic_load(&ctl->failFrom, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const unsigned long long blk0 = cid * G;
        const uint32_t nblk = A.numBlocksCall - blk0 < G ? (uint32_t)(A.numBlocksCall - blk0) : G;
        // a root's context: the block's window origin and length, "at MAX already" (second pass), J-mer / neighbours / letters behind (expand_root)
        auto root_context = [&](uint32_t blockInChunk, uint32_t search, uint32_t strand, uint32_t& win, uint32_t& n, bool& blockDone) -> XRoot {
            XRoot xr; xr.jb = xr.jn = xr.ext = 0u; xr.bad = 1u;
            unsigned long long gb = blk0 + blockInChunk;        // ordinal of the block among this call's blocks (gm_stage1.inc)
            if (A.chunkBlocks) {
                const uint32_t qq = (uint32_t)gb / A.chunkBlocks;
                gb = (unsigned long long)(qq * A.chunkStride + A.chunkIndex) * A.chunkBlocks + ((uint32_t)gb - qq * A.chunkBlocks);
            }
            gb += A.blockBegin;
            if (A.blockList) { const uint2 e = A.blockList[gb]; win = e.x; n = e.y & 0xFFu; }   // (32-bit rows: slice positions fit 32 bits)
            else { win = (uint32_t)gb * A.stepSize; const uint32_t left = (uint32_t)A.numKmers - win; n = left < A.stepSize ? left : A.stepSize; }
            // (pass 2 of two: a block whose k-mers the first pass has all brought to MAX needs nothing more -- min(total, MAX), src/algo.hpp:36,48,191)
            blockDone = false;
            if (A.xmode == 2u) {
                blockDone = true;
                for (uint32_t i = 0; i < n && blockDone; ++i) blockDone = A.acc[win + i] >= A.maxVal;
            }
            if (n == A.stepSize && !blockDone) {
                const uint32_t a0 = n - 1u + (((search < 4u ? A.jumpAPacked[0] : A.jumpAPacked[1]) >> (8u * (search & 3u))) & 0xFFu);
                xr = expand_root(mem, A.textBegin + win, A.K + n - 1u, strand, a0, A.jumpJ, jl[search].w, (jl[12u + search].w & 0xFFu) != 0u);
            }
            return xr;
        };
#pragma unroll 1
        for (uint32_t g0 = 0; g0 < nblk && !dead; g0 += (NB ? NB : nblk)) {
        const uint32_t nbg = NB ? (nblk - g0 < NB ? nblk - g0 : NB) : nblk;
        if (NB) {   // ---- one lane per ROOT of the group ----
            uint4 c0 = make_uint4(0, 0, 0, 0); uint32_t c1 = 0x100u;
            if (lane < nbg * RPB) {
                const uint32_t bi = lane / RPB, rq = lane - bi * RPB;
                const uint32_t wm1 = A.wmapRoots[rq];
                uint32_t win, n; bool blockDone;
                const XRoot xr = root_context(g0 + bi, wm1 & 7u, (wm1 >> 3) & 1u, win, n, blockDone);
                c0 = make_uint4(xr.jb, xr.jn, xr.ext, win); c1 = n | (xr.bad ? 0x100u : 0u) | (blockDone ? 0x200u : 0u);
                if (rq == 0u && !blockDone)   // the block's packet window: what every packet of its roots carries
                    for (uint32_t j = 0; j < A.pktChunks; ++j) {
                        const unsigned long long w0 = nib64(mem, A.textBegin + win + 32u * j), w1 = nib64(mem, A.textBegin + win + 32u * j + 16u);
                        wwin[bi * 3u + j] = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
                    }
            }
            rc0[lane] = c0; rc1[lane] = c1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const uint32_t nwork = nbg * IPB;
#pragma unroll 1
        for (uint32_t base = 0; base < nwork && !dead; base += 64u) {
            const uint32_t l = base + lane;
            const bool on = l < nwork;
            XRoot xr; xr.jb = xr.jn = xr.ext = 0u; xr.bad = 1u;
            XItem it; it.alive = 0ull; it.gcur = it.sh = it.state = it.widx = it.wsel = 0u;
            uint32_t win = 0, nss = 0, jm0 = 0, hword = 0, jd = 0, big = 0;
            bool pendRoot = false;
            if (on) {
                const uint32_t bi = l / IPB, q = l - bi * IPB;
                const uint32_t wm = A.wmap[q], search = wm & 7u, strand = (wm >> 3) & 1u, jp = wm >> 8;
                uint32_t n; bool blockDone;
                if (NB) {
                    const uint32_t ri = bi * RPB + strand * A.nSearches + search;
                    const uint4 c0 = rc0[ri]; const uint32_t c1 = rc1[ri];
                    xr.jb = c0.x; xr.jn = c0.y; xr.ext = c0.z; win = c0.w; n = c1 & 0xFFu; xr.bad = (c1 >> 8) & 1u; blockDone = ((c1 >> 9) & 1u) != 0u;
                    big = bi;
                } else xr = root_context(g0 + bi, search, strand, win, n, blockDone);
                nss = pkt_root_word(n, strand, search);
                const uint4 fji = jl[search];   // {first item | items << 16, meta at depth J relative to n - 1, first item, neighbour-filter mask}
                hword = fji.w;
                // an odd block shape or an N inside the J-mer: the root walks the tree from its root (said by its first item's lane; with two passes: by the first)
                if (blockDone) xr.bad = 1u;
                else if (xr.bad) pendRoot = A.xmode == 1u ? true : A.xmode == 2u ? false : jp == (fji.x & 0xFFFFu);
                if (!xr.bad) {
                    jm0 = meta_pack((fji.y & 0x1FFu) + n - 1u, ((fji.y >> 9) & 0x1FFu) + n - 1u, fji.y >> 18, 0u, M_OSS);
                    jd = A.patterns[jp];
                    const uint4 lim = jl[12u + search];
                    it = expand_item(jd, jump_item_flags(jp, lim.x, lim.y, lim.z), xr, tab);
                    if (it.state == 2u) {
                        const unsigned long long pw = A.jbits[(size_t)it.wsel * A.jbitsWords + it.widx];
                        expand_word(it, pw, jd, xr, tab);
#ifdef GM_COUNTERS
                        cWords++;
#endif
                    }
                    if (A.xmode == 2u) expand_strip_exact(it);   // (the first pass has taken the pattern without a substitution)
                    if (wm & WMAP_ROOT_ONLY) it.state = 0u;      // (first pass: this search has no such pattern)
                }
            }
            // ---- the rotations of the 64 items, dealt out anew: lane j takes the j-th rotation of the batch, so that a turn of the loop reads 64
            // table entries whatever the items' sizes (an item has 0 .. 64 surviving rotations: with one item per lane a turn lasted as long as a
            // memory round trip and there were as many turns as the largest item had rotations -- 127 ms of a 283 ms pass, profiles/r06) ----
            uint32_t cnt = 0;
            if (pendRoot) cnt = 1u;
            else if (on && !xr.bad) cnt = expand_count(it);
            uint32_t incl = cnt;
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, (int)d); if (lane >= d) incl += up; }
            const uint32_t total = (uint32_t)__shfl((int)incl, 63);
            if (total == 0u) continue;
            // the lanes' contexts and the exclusive prefix sums of their counts, in LDS (this wavefront's 64 slots)
            cx0[lane] = make_uint4(xr.jb, xr.jn, jm0, hword);
            cx1[lane] = make_uint4(win, nss, it.gcur, it.sh | (pendRoot ? 0x100u : 0u) | (it.state == 1u ? 0x200u : 0u) | big << 16);
            cx2[lane] = make_uint2((uint32_t)it.alive, (uint32_t)(it.alive >> 32));
            pre[lane] = incl - cnt;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
            for (uint32_t t0 = 0; t0 < total; t0 += 64u) {
                const uint32_t g = t0 + lane;
                bool take = false;
                uint32_t cls = 0, pwin = 0, pnss = 0, pbi = 0;
                uint4 h0 = make_uint4(0, 0, 0, 0);
                if (g < total) {
                    uint32_t src = 0;   // the last lane whose first rotation is <= g (lanes without rotations share their successor's start)
#pragma unroll
                    for (uint32_t step = 32u; step >= 1u; step >>= 1) if (pre[src | step] <= g) src |= step;
                    const uint32_t k = g - pre[src];
                    const uint4 c0 = cx0[src], c1 = cx1[src];
                    pwin = c1.x; pnss = c1.y; pbi = c1.w >> 16;
                    if (c1.w & 0x100u) {   // an odd block shape or an N inside the J-mer: the root itself (root_node, gm_engine.h)
                        take = true; cls = 0u;
                        const uint32_t n = pnss & 0xFFu, search = (pnss >> 9) & 7u;
                        const uint32_t a = n - 1u + ((A.table[(size_t)(n - 1u) * 8u + search].y >> 16) & 0xFFu);
                        h0 = make_uint4(0u, 0u, (uint32_t)A.nRows, meta_pack(a, a, 0u, 0u, M_OSS));
                    } else {
                        const uint2 av = cx2[src];
                        const uint32_t rw = expand_nth(c1.z, c1.w & 31u, (c1.w & 0x200u) != 0u, (unsigned long long)av.y << 32 | av.x, k);   // the k-th surviving rotation of the item
                        const uint4 e = A.jtab[rot_add(c0.x, rw)];
                        const XNode x = expand_filter(e.x, e.y, e.z, e.w, rw, c0.z, c0.y, c0.w, A.E, A.nbFilter, A.verifyT);
#ifdef GM_COUNTERS
                        cJumps++;
                        if (!x.take && e.z == 1u) cDrops++;
                        if (e.z == 2u && (c0.y & 0x8000u) && A.nbFilter == 1u) cDrops2 += x.take ? (x.rlo == ~0u ? 1u : 0u) : 2u;
#endif
                        take = x.take != 0u;
                        h0 = make_uint4(x.flo, x.rlo, x.w, x.meta); cls = expand_class(x.errs);
                        // (self hits -- forward strand, no error spent, one row: the window's own location -- stay with the walker: a chunk of phase A
                        //  that does not fit its buffers is redone by the next slice, and an add into the difference plane cannot be taken back)
                    }
                }
                // ---- append: every class in turn, all lanes in step ----
#pragma unroll
                for (uint32_t c = 0; c < 3u; ++c) {
                    const bool mine = take && cls == c;
                    const unsigned long long m = __ballot(mine);
                    if (m == 0ull) continue;
                    const uint32_t cntc = (uint32_t)__popcll(m), room = XREGION - ru[c];
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    uint32_t nbase = 0xFFFFFFFFu;
                    if (cntc > room) {   // a new region for this class: ONE atomic per XREGION packets
                        if (lane == 0u) {
                            if (c == 2u) {
                                const uint32_t b2 = atomicAdd(&ctl->tailY, XREGION);
                                if (b2 <= A.capY - XREGION) { nbase = b2; atomicMax(&ctl->valid[2], b2 + XREGION); }
                            } else {
                                const unsigned long long old = atomicAdd(&ctl->tailsX, c == 0u ? (unsigned long long)XREGION : (unsigned long long)XREGION << 32);
                                const uint32_t b0 = (uint32_t)old, b1 = (uint32_t)(old >> 32);
                                if ((unsigned long long)b0 + b1 + XREGION <= A.capX) { nbase = c == 0u ? b0 : b1; atomicMax(&ctl->valid[c], nbase + XREGION); }
                            }
                        }
                        nbase = (uint32_t)__shfl((int)nbase, 0);
                        if (nbase == 0xFFFFFFFFu) dead = true;
                    }
                    if (mine && (rank < room || !dead)) {
                        const uint32_t slot = rank < room ? rb[c] + ru[c] + rank : nbase + (rank - room);
                        uint4* pk = (c == 2u ? A.pktY : A.pktX) + (size_t)(c == 1u ? A.capX - 1u - slot : slot) * U;
                        pk[0] = h0;
                        pk[1] = make_uint4(pwin, pnss, (uint32_t)cid, stamp);
                        if (NB) for (uint32_t j = 0; j < A.pktChunks; ++j) pk[2u + j] = wwin[pbi * 3u + j];   // (the group's windows are in LDS)
                        else for (uint32_t j = 0; j < A.pktChunks; ++j) {
                            const unsigned long long w0 = nib64(mem, A.textBegin + pwin + 32u * j), w1 = nib64(mem, A.textBegin + pwin + 32u * j + 16u);
                            pk[2u + j] = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
                        }
#ifdef GM_COUNTERS
                        cPackets++;
#endif
                    }
                    if (cntc > room) { rb[c] = nbase; ru[c] = cntc - room; } else ru[c] += cntc;
                }
                if (dead) break;
            }
            __builtin_amdgcn_wave_barrier();   // (the next batch overwrites the contexts)
        }
        __builtin_amdgcn_wave_barrier();   // (the next group overwrites the root contexts)
        }
        if (dead) { if (lane == 0u) atomicMin(&ctl->failFrom, (uint32_t)cid); break; }   // this chunk (and every later one) is redone by the next slice
    }
#ifdef GM_COUNTERS
    atomicAdd(&A.counters[38], (unsigned long long)cJumps);
    atomicAdd(&A.counters[41], (unsigned long long)cDrops);
    atomicAdd(&A.counters[47], (unsigned long long)cWords);
    atomicAdd(&A.counters[48], (unsigned long long)cDrops2);
    atomicAdd(&A.counters[49], (unsigned long long)cPackets);   // detail[47]: node packets written by phase A
#endif
}

// SA ranges of every ACGT string of length q in both indexes (right extensions from the root): the top of the search
// tree, tabulated once per index and q.
// Entries with exactly ONE row additionally carry the text next to that only occurrence (when the suffix array is resident; 32-bit rows):
//   bits 0..11  the 6 symbols to its right (2 bits each, nearest first), bits 12..14 how many of them are letters A,C,G,T inside the sequence,
//   bits 16..27 the 6 symbols to its left (nearest first), bits 28..30 their count, bits 15 and 31 = 1.
// A search that lands on such an entry compares its own neighbouring characters with them before it spends a single memory request on the
// node (search_body, jump patterns): on a genome most one-row entries are chance hits of a substituted string.
template <int WPP>
__global__ __launch_bounds__(256) void qmer_table_kernel(const uint32_t* __restrict__ blkRev, const uint64_t* __restrict__ Cin, uint64_t nRows, uint32_t q,
                                                         uint4* __restrict__ out, const uint32_t* __restrict__ sa, const uint8_t* __restrict__ textS)
{
    typedef typename BlockGeom<WPP>::row_t row_t;
    const uint64_t idx64 = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;   // q = 16: 2^32 strings (HIP: < 2^32 threads per grid row)
    if (idx64 >= (1ull << (2u * q))) return;
    const uint32_t idx = (uint32_t)idx64;
    constexpr uint32_t SPB = BlockGeom<WPP>::SPB, WPB = BlockGeom<WPP>::WPB;
    row_t flo = 0, rlo = 0, w = (row_t)nRows;
    for (uint32_t i = 0; i < q && w; ++i) {
        const uint32_t c = (idx >> (2u * (q - 1u - i))) & 3u;
        row_t rl[NLET], rh[NLET];
        const row_t lo = rlo, hi = rlo + w;
        block_rank<WPP>(blkRev + (size_t)(lo / SPB) * WPB, (uint32_t)(lo % SPB), rl);
        block_rank<WPP>(blkRev + (size_t)(hi / SPB) * WPB, (uint32_t)(hi % SPB), rh);
        row_t tot = 0, below = 0;
        for (uint32_t x = 0; x < NLET; ++x) { const row_t cx = rh[x] - rl[x]; tot += cx; if (x < c) below += cx; }
        flo += (w - tot) + below;           // sentinels sort before every letter
        rlo = (row_t)Cin[c] + rl[c];
        w = rh[c] - rl[c];
    }
    uint32_t nb = 0;
    if (sizeof(row_t) == 4 && w == 1u && sa != nullptr) {
        const uint8_t* t = textS + sa[flo];            // the occurrence; the sentinel text has 512 sentinels of padding on either side
        uint32_t nr = 0, nl = 0;
        while (nr < NB_SYMS && t[q + nr] < (uint8_t)SYM_N) { nb |= (uint32_t)t[q + nr] << (2u * nr); ++nr; }
        while (nl < NB_SYMS && t[-1 - (int)nl] < (uint8_t)SYM_N) { nb |= (uint32_t)t[-1 - (int)nl] << (16u + 2u * nl); ++nl; }
        nb |= nr << 12 | nl << 28 | 0x80008000u;
    }
    if (sizeof(row_t) == 4 && w == 2u && sa != nullptr) {
        // TWO rows: 3 + 3 symbols next to either occurrence, 16 bits per row (row flo in the low half): bits 0..5 the symbols to its right,
        // 6..11 to its left (nearest first), 12..13 / 14..15 how many of them are letters inside the sequence
        for (uint32_t r = 0; r < 2u; ++r) {
            const uint8_t* t = textS + sa[flo + r];
            uint32_t x = 0, nr = 0, nl = 0;
            while (nr < NB_SYMS2 && t[q + nr] < (uint8_t)SYM_N) { x |= (uint32_t)t[q + nr] << (2u * nr); ++nr; }
            while (nl < NB_SYMS2 && t[-1 - (int)nl] < (uint8_t)SYM_N) { x |= (uint32_t)t[-1 - (int)nl] << (6u + 2u * nl); ++nl; }
            nb |= (x | nr << 12 | nl << 14) << (16u * r);
        }
    }
    NodeIO<row_t>::store_qentry(out, idx, flo, rlo, w, nb);
}

// fasta id of the sequence every suffix-array row lies in (--exclude-pseudo: FileSetEnv::row_located)
__global__ __launch_bounds__(256) void row_file_kernel(const uint32_t* __restrict__ sa, uint64_t n, const uint64_t* __restrict__ cum, uint32_t nSeq,
                                                       const uint32_t* __restrict__ seqFile, uint8_t* __restrict__ out)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x)
        out[r] = (uint8_t)seqFile[locate_position(cum, nSeq, sa[r]).x];
}

// existence bitmap of the J-mers (groups of jump patterns, gm_oss.h): bit idx & 63 of word idx >> 6 = table entry idx holds a row.
// One wavefront per word: coalesced 16-byte reads of the table, one ballot.  (32-bit rows: 16-byte entries)
__global__ __launch_bounds__(256) void jbits_kernel(const uint4* __restrict__ tab, uint64_t nEntries, unsigned long long* __restrict__ bits)
{
    const uint64_t i = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    bool alive = false;
    if (i < nEntries) { const uint4 t = tab[i]; alive = (t.z | (t.w & 0x00FF0000u)) != 0u; }   // (width bits 32..39 of a 64-bit-row entry: byte 2 of .w)
    const unsigned long long m = __ballot(alive);
    if ((threadIdx.x & 63u) == 0u && i < nEntries) bits[i >> 6] = m;
}

// bitmaps of kind 1 (gm_oss.h): for every run of J + 2 letters of the sentinel text (no N, no sequence end inside) the bit of its first J
// letters in the array of its last two.  low / mid: the two index layouts (either may be null).  One thread per 64 text positions.
__global__ __launch_bounds__(256) void jbits1_kernel(const uint8_t* __restrict__ textS, uint64_t n, uint32_t J, unsigned long long* __restrict__ low,
                                                     unsigned long long* __restrict__ mid, uint64_t words)
{
    const uint64_t p0 = (((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x) * 64ull;
    if (p0 >= n) return;
    const uint64_t keep = (1ull << (2u * (J + 2u))) - 1ull;
    uint64_t code = 0; uint32_t run = 0;
    for (uint64_t q = p0; q < p0 + 64u + J + 1u && q < n; ++q) {
        const uint32_t c = textS[q];
        if (c < SYM_N) { code = ((code << 2) | c) & keep; ++run; } else run = 0;
        if (run >= J + 2u && q - (J + 1u) >= p0) {
            const uint32_t idx = (uint32_t)(code >> 4), ext = (uint32_t)code & 15u;
            if (low) atomicOr(&low[(size_t)ext * words + (idx >> 6)], 1ull << (idx & 63u));
            if (mid) { const uint32_t sw = jump_swap_mid(idx); atomicOr(&mid[(size_t)ext * words + (sw >> 6)], 1ull << (sw & 63u)); }
        }
    }
}

// store planes -> c[]: 16 bytes per lane where the three arrays are aligned alike (the planes are; `out` is the caller's)
template <typename TValue, typename TPlane>
__global__ __launch_bounds__(256) void finalize2_kernel(const TPlane* __restrict__ accF, const TPlane* __restrict__ accR, TValue* __restrict__ out, uint64_t n, uint32_t maxVal, ChunkSel sel)
{
    static_assert(sizeof(TValue) == sizeof(TPlane), "planes are as wide as the result");
    constexpr uint32_t EPV = 16u / sizeof(TValue);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nr = own_ranges(sel, n);
    const bool full = maxVal == (sizeof(TValue) == 1 ? 255u : 65535u);
    for (uint64_t q = blockIdx.y; q < nr; q += gridDim.y) {
        uint64_t b, e;
        own_range(sel, n, q, b, e);
        const uintptr_t ao = reinterpret_cast<uintptr_t>(out + b), af = reinterpret_cast<uintptr_t>(accF + b), ar = reinterpret_cast<uintptr_t>(accR + b);
        uint64_t bb = b, be = b;   // [bb, be): whole 16-byte vectors
        if (full && ((ao ^ af) & 15u) == 0u && ((ao ^ ar) & 15u) == 0u) {
            bb = b + ((16u - (uint32_t)(ao & 15u)) & 15u) / sizeof(TValue);
            if (bb > e) bb = e;
            be = bb + (e - bb) / EPV * EPV;
        }
        for (uint64_t j = b + tid; j < bb; j += nth) { const uint32_t v = (uint32_t)accF[j] + accR[j]; out[j] = (TValue)(v < maxVal ? v : maxVal); }
        for (uint64_t j = be + tid; j < e; j += nth) { const uint32_t v = (uint32_t)accF[j] + accR[j]; out[j] = (TValue)(v < maxVal ? v : maxVal); }
        const uint4* F = reinterpret_cast<const uint4*>(accF + bb); const uint4* R = reinterpret_cast<const uint4*>(accR + bb);
        uint4* O = reinterpret_cast<uint4*>(out + bb);
        for (uint64_t v = tid; v < (be - bb) / EPV; v += nth) {
            const uint4 x = F[v], y = R[v];
            uint4 r;
            if (sizeof(TValue) == 1) { r.x = sat_add_u8x4(x.x, y.x); r.y = sat_add_u8x4(x.y, y.y); r.z = sat_add_u8x4(x.z, y.z); r.w = sat_add_u8x4(x.w, y.w); }
            else { r.x = sat_add_u16x2(x.x, y.x); r.y = sat_add_u16x2(x.y, y.y); r.z = sat_add_u16x2(x.z, y.z); r.w = sat_add_u16x2(x.w, y.w); }
            O[v] = r;
        }
    }
}

template <typename TValue>
__global__ __launch_bounds__(256) void finalize_kernel(const uint32_t* __restrict__ acc, TValue* __restrict__ out, uint64_t n, uint32_t maxVal, ChunkSel sel)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nr = own_ranges(sel, n);
    for (uint64_t q = blockIdx.y; q < nr; q += gridDim.y) {
        uint64_t b, e;
        own_range(sel, n, q, b, e);
        // whole groups of four positions where acc is 16-byte and out 4- (8-) byte aligned
        uint64_t bb = b + ((16u - (uint32_t)(reinterpret_cast<uintptr_t>(acc + b) & 15u)) & 15u) / 4u, be;
        if (bb > e) bb = e;
        if ((reinterpret_cast<uintptr_t>(out + bb) & (sizeof(TValue) * 4 - 1)) != 0) { bb = b; be = b; } else be = bb + (e - bb) / 4 * 4;
        for (uint64_t j = b + tid; j < bb; j += nth) { const uint32_t v = acc[j]; out[j] = (TValue)(v < maxVal ? v : maxVal); }
        for (uint64_t j = be + tid; j < e; j += nth) { const uint32_t v = acc[j]; out[j] = (TValue)(v < maxVal ? v : maxVal); }
        for (uint64_t g = tid; g < (be - bb) / 4; g += nth) {
            const uint4 v = *reinterpret_cast<const uint4*>(acc + bb + g * 4);
            TValue r[4] = {(TValue)(v.x < maxVal ? v.x : maxVal), (TValue)(v.y < maxVal ? v.y : maxVal),
                           (TValue)(v.z < maxVal ? v.z : maxVal), (TValue)(v.w < maxVal ? v.w : maxVal)};
            if (sizeof(TValue) == 1) *reinterpret_cast<uint32_t*>(out + bb + g * 4) = *reinterpret_cast<uint32_t*>(r);
            else *reinterpret_cast<uint2*>(out + bb + g * 4) = *reinterpret_cast<uint2*>(r);
        }
    }
}

// acc + difference plane of the verified runs -> c[] (CountEnv::leaf_range).  The runs of a block never leave it and blocks start at
// multiples of stepSize from the range's first position (regular partition; a shard's chunks are whole blocks), so the running sum
// restarts at every block: count(j) = acc[j] + sum of diff over [block start of j, j].  A workgroup stages a tile of whole blocks of the
// plane in LDS; every position sums its block's entries up to itself from there (stepSize / 2 LDS reads on average).
template <typename TValue>
__global__ __launch_bounds__(256) void finalize_diff_kernel(const uint32_t* __restrict__ acc, const uint32_t* __restrict__ diff, TValue* __restrict__ out, uint64_t n, uint32_t maxVal,
                                                            ChunkSel sel, uint32_t stepSize)
{
    constexpr uint32_t TILE = 2048;
    __shared__ uint32_t sd[TILE];
    const uint32_t tile = TILE / stepSize * stepSize;   // whole blocks (stepSize <= MAX_K)
    const uint64_t nr = own_ranges(sel, n);
    for (uint64_t q = blockIdx.y; q < nr; q += gridDim.y) {
        uint64_t b, e;
        own_range(sel, n, q, b, e);
        for (uint64_t t0 = b + (uint64_t)blockIdx.x * tile; t0 < e; t0 += (uint64_t)gridDim.x * tile) {
            const uint32_t len = e - t0 < tile ? (uint32_t)(e - t0) : tile;
            for (uint32_t k = threadIdx.x; k < len; k += 256u) sd[k] = diff[t0 + k];
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < len; k += 256u) {
                const uint32_t off = k % stepSize;
                uint32_t s = 0;
                for (uint32_t i = 0; i <= off; ++i) s += sd[k - i];
                const uint64_t tot = (uint64_t)acc[t0 + k] + s;   // (a sticky top bit of acc stays above MAX)
                out[t0 + k] = (TValue)(tot < maxVal ? tot : maxVal);
            }
            __syncthreads();
        }
    }
}

// --exclude-pseudo: hits[j] = distinct_sequences.size(), a narrowing store without saturation (algo.hpp:360)
template <typename TValue>
__global__ __launch_bounds__(256) void finalize_fileset_kernel(const uint32_t* __restrict__ bits, uint32_t wordsPerKmer, TValue* __restrict__ out, uint64_t n, ChunkSel sel)
{
    const uint64_t nr = own_ranges(sel, n);
    for (uint64_t q = blockIdx.y; q < nr; q += gridDim.y) {
        uint64_t b, e;
        own_range(sel, n, q, b, e);
        for (uint64_t j = b + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < e; j += (uint64_t)gridDim.x * blockDim.x) {
            uint32_t c = 0;
            for (uint32_t w = 0; w < wordsPerKmer; ++w) c += (uint32_t)__popc(bits[j * wordsPerKmer + w]);
            out[j] = (TValue)c;
        }
    }
}

// ---- run-length form of c[] (saveWig / saveBedGraph scans, src/output.hpp:74-187) --------------------------------
template <typename TValue>
__global__ __launch_bounds__(256) void run_heads_kernel(const TValue* __restrict__ c, uint64_t n, uint8_t* __restrict__ head)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) head[i] = (i == 0 || c[i] != c[i - 1]) ? 1 : 0;
}
__global__ void seq_heads_kernel(const uint64_t* __restrict__ cumLocal, uint32_t nSeq, uint64_t n, uint8_t* __restrict__ head)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nSeq && cumLocal[s] < n) head[cumLocal[s]] = 1;   // runs never cross a sequence boundary
}
template <typename TValue>
__global__ __launch_bounds__(256) void run_values_kernel(const TValue* __restrict__ c, const uint32_t* __restrict__ starts, uint64_t nRuns, uint16_t* __restrict__ val)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nRuns; r += (uint64_t)gridDim.x * blockDim.x) val[r] = (uint16_t)c[starts[r]];
}

// ---- sampling of the suffix array (-S) ---------------------------------------------------------------------------------------
// mark[row / 32] bit row % 32 = SA[row] lies at an in-sequence offset that is a multiple of `s` (never on a sentinel)
template <typename R>
__global__ __launch_bounds__(256) void sa_mark_kernel(const R* __restrict__ sa, const uint64_t* __restrict__ cum, uint32_t nSeq, uint64_t n, uint32_t s,
                                                      uint2* __restrict__ mark)
{
    const uint64_t words = (n + 31) / 32;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t bits = 0;
        for (uint32_t t = 0; t < 32; ++t) {
            const uint64_t row = w * 32 + t;
            if (row >= n) break;
            const uint2 sp = locate_position(cum, nSeq, sa[row]);
            const uint64_t len = cum[sp.x + 1] - cum[sp.x];
            if (sp.y < len && sp.y % s == 0u) bits |= 1u << t;
        }
        mark[w] = make_uint2(bits, (uint32_t)__popc(bits));   // .y: count, turned into "samples before this word" by a scan
    }
}
__global__ __launch_bounds__(256) void sa_mark_counts_kernel(const uint2* __restrict__ mark, uint64_t words, uint32_t* __restrict__ cnt)
{
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) cnt[w] = mark[w].y;
}
__global__ __launch_bounds__(256) void sa_mark_offsets_kernel(uint2* __restrict__ mark, uint64_t words, const uint32_t* __restrict__ before)
{
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) mark[w].y = before[w];
}
template <typename R>
__global__ __launch_bounds__(256) void sa_compact_kernel(const R* __restrict__ sa, const uint2* __restrict__ mark, uint64_t n, R* __restrict__ samples)
{
    for (uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 m = mark[row >> 5];
        const uint32_t bit = 1u << (row & 31u);
        if (m.x & bit) samples[m.y + (uint32_t)__popc(m.x & (bit - 1u))] = sa[row];
    }
}

// zero the calling shard's chunks of a workspace (elements of `eb` bytes, positions [0, n) of the range)
__global__ __launch_bounds__(256) void clear_chunks_kernel(uint8_t* __restrict__ base, uint32_t eb, uint64_t n, ChunkSel sel)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nr = own_ranges(sel, n);
    for (uint64_t q = blockIdx.y; q < nr; q += gridDim.y) {
        uint64_t b, e;
        own_range(sel, n, q, b, e);
        uint8_t* p = base + b * eb;                    // bytes [0, nb): head up to a 16-byte boundary, whole vectors, tail
        const uint64_t nb = (e - b) * eb;
        uint64_t h = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u;
        if (h > nb) h = nb;
        const uint64_t nv = (nb - h) / 16;
        for (uint64_t j = tid; j < h; j += nth) p[j] = 0;
        for (uint64_t j = h + nv * 16 + tid; j < nb; j += nth) p[j] = 0;
        uint4* v = reinterpret_cast<uint4*>(p + h);
        for (uint64_t j = tid; j < nv; j += nth) v[j] = make_uint4(0, 0, 0, 0);
    }
}

// resetLimits (algo.hpp:10-22): zero the last K-1 positions of every sequence of the slice.
template <typename TValue>
__global__ void reset_limits_kernel(TValue* out, const uint64_t* cumLocal, uint32_t nSeq, uint32_t K)
{
    const uint32_t s = blockIdx.x + 1u;
    if (s > nSeq) return;
    const uint64_t len = cumLocal[s] - cumLocal[s - 1];
    const uint64_t lim = (uint64_t)K < len + 1 ? (uint64_t)K : len + 1;
    for (uint64_t j = 1 + threadIdx.x; j < lim; j += blockDim.x) out[cumLocal[s] - j] = (TValue)0;
}

}  // namespace gm
