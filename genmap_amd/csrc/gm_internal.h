// gm_internal.h -- private declarations shared by the translation units of libgenmap_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdarg>
#include <vector>
#include <map>
#include <thread>
#include "../../include/genmap_amd.h"
#include "gm_common.h"
#include "gm_oss.h"

namespace gm {

void set_error(const char* fmt, ...);

// gm_build.hip
int build_sa_bwt(const uint8_t* d_codes, const uint64_t* d_cum, uint32_t nSeq, uint64_t textLen, int rev,
                 uint32_t* d_sa_out, uint8_t* d_bwt, int* roundsOut);
// the same for 64-bit rows (two stable sorting passes per doubling round: the (rank, rank) key no longer fits 64 bits)
int build_sa_bwt_wide(const uint8_t* d_codes, const uint64_t* d_cum, uint32_t nSeq, uint64_t textLen, int rev,
                      uint64_t* d_sa_out, uint8_t* d_bwt, int* roundsOut);

}  // namespace gm

namespace gm {
// knobs of the search kernel's scheduling; set with gm_index_set_tuning (tests, sweeps), never read from the environment.
// -1 = "library default for this call" (depends on K, E and the index size, see prepare_search)
struct Tuning {
    int verifyT = -1, ldsStack = -1, blocksPerCU = 4, qtable = -1, satMinW = 256, fetchBatch = -1, probation = -1, verifyCost = 3;
    int noStore = 0, noSaturate = 0, skipDup = -1, coop = -1, useCtx = 1, steal = -1, partBias = 0, childTables = -1, ossWeights = -1, jump = -1, selfHit = 1, jumpFilter = 1, rangeAdd = 1, verifyTExt = -1, jumpGroups = -1;
    int noWrap = -1;                   // 0: adds into the accumulators always return and check for a 2^32 wrap-around (gm_api.hip: acc_cannot_wrap)
    int jumpLayouts = -1;              // 0: groups of jump patterns only in the two layouts of round 4
    int ldsPad = 0;                    // measurement: extra bytes of LDS per block
    int patBatch = -1;                 // jump patterns: idle lanes that must wait for their pattern turn before parts A / B of the loop run (-1: 1 = every iteration)
    int fastVerify = -1;               // -1: whenever the call allows it (gm_api.hip: prepare_search), 0: never
    int iterCap = -1, stallCap = -1;   // bounds of a hung search loop (gm_kernels.h: SearchArgs::iterCap / stallCap); -1: never / 2^22 idle iterations
    // the split search (gm_expand.h): phase A enumerates the jump patterns into node packets, the walker draws packets
    int expand = -1;                   // -1: where it pays (gm_api.hip: prepare_search), 0: never (the one-loop kernel of rounds 3-5), 1: whenever the call has jump patterns
    int expandMB = -1;                 // MiB of packet buffers (-1: a share of the free device memory); tests force small ones: many slices, overflowing chunks
    int expandChunk = -1;              // k-mer blocks per chunk of phase A (-1: about a thousand work items)
    int expandOverlap = -1;            // 1: phase A of slice i + 1 runs beside the walker of slice i (three walker blocks and one block of phase A per CU); 0 / -1: one after the other
    int expandTwoPass = -1;            // 1 / -1: the patterns without a substitution of every root first, then the rest for blocks not at MAX yet; 0: one pass
    int expandShare = -1;              // 1 / -1: phase A computes a root's context once (a lane per root of a group of blocks, LDS), 0: every item computes it
    int expandOcc = -1;                // blocks of phase A per CU (-1: what the occupancy query says, at most 8)
    int win2 = -1;                     // needle windows at 2 bits per symbol: 1 always, 0 never, -1 where they give LDS stack levels back (long windows)
    int satDrawW = -1;                 // the walker drops a drawn node at least this wide when its block's k-mers are all at MAX (-1: 1 = every packet)
};
}  // namespace gm

struct gm_index {
    int device = 0;
    uint32_t wpp = 3;                 // words per plane of the rank blocks (1, 3, 9; 2 = the wide geometry, 64-bit rows)
    bool wide = false;                // rows / ranges / text positions are 64-bit (d_sa, C, q-mer tables, nodes)
    uint64_t nRows = 0, textLen = 0;
    uint32_t nSeq = 0, sampling = 0, alphabet = 4;
    uint32_t* d_blk[2] = {nullptr, nullptr};
    uint64_t blkBytes = 0;            // per direction
    uint64_t C[gm::NLET + 1] = {0, 0, 0, 0, 0, 0};
    uint8_t* d_text = nullptr;        // sentinel-free codes, one byte each (= d_textAlloc + 16: readers may touch a few bytes around)
    uint8_t* d_textAlloc = nullptr; uint8_t* d_textSAlloc = nullptr;
    uint4* d_text4 = nullptr;         // the text at 4 bits per symbol: needle windows are staged into LDS from it
    uint4* d_text2 = nullptr;         // ... at 2 bits per symbol (an N stored as A), made by the first call that stages windows from it (ensure_text2)
    uint16_t* d_nflag = nullptr;      // entry b: bit j = chunk 8b + j of d_text2 (64 symbols) holds an N
    std::vector<uint64_t> cum;        // nSeq + 1
    uint64_t* d_cum = nullptr;
    void* d_sa = nullptr;             // forward suffix array, uint32_t or (wide) uint64_t per row (kept when sampling == 1): locate = one HBM read
    uint2* d_saMark = nullptr;        // sampling > 1: per 32 rows {mark bits, samples before this word}
    void* d_saSamples = nullptr;      //               SA values of the marked rows, in row order (uint32_t, or uint64_t with 64-bit rows)
    uint64_t nSamples = 0;
    uint8_t* d_textS = nullptr;       // sentinel text (verification of narrow nodes), present with d_sa
    uint4* d_ctx = nullptr;           // verification records {SA[row], 56 symbols around it}, 32 B per row, when HBM allows (gm_kernels.h: CTX_*)
    std::map<uint32_t, uint4*> qtables;   // q -> device table of 4^q entries (built on first use)
    std::map<uint32_t, unsigned long long*> jbits;   // q -> existence bitmap of the q-mers, 4^q bits (groups of jump patterns, gm_oss.h)
    std::map<uint32_t, uint32_t> jbitsExtra;         // q -> kind-0 planes of the layouts beyond LOW that follow the planes of jbitsLevel
    std::map<uint32_t, int> jbitsLevel;              // q -> 0: "the q-mer occurs" only, 1: + "... followed by two given letters" (16 x 4^q bits), 2: + the same in the MID layout
    uint4* d_jinfo2 = nullptr;
    uint64_t sig = 0; bool sigValid = false;   // signature of the call whose tables are on the device
    uint32_t lastQ = 0;                   // longest q-mer table of the last call | jump length << 8 (statistics)
    uint32_t qtableCap = 0;               // != 0: longest prefix that fitted the device so far
    uint64_t qtableBytes = 0;
    uint64_t* d_C = nullptr;
    uint32_t* d_seqFile = nullptr; uint64_t seqFileCap = 0;
    // gm_locate: work buffers kept between calls
    uint32_t* d_locCnt = nullptr; uint64_t* d_locOffs = nullptr; uint64_t* d_locEmit = nullptr; uint64_t* d_locSorted = nullptr; uint8_t* d_locTmp = nullptr; uint32_t* d_locSeg = nullptr;
    uint64_t locCntCap = 0, locOffsCap = 0, locEmitCap = 0, locSortedCap = 0, locTmpCap = 0, locSegCap = 0;
    uint8_t* d_rowFile = nullptr; uint64_t rowFileSig = 0; bool rowFileValid = false;   // fasta id per suffix-array row for the file assignment with this signature
    uint32_t* d_bits = nullptr; uint64_t bitsCap = 0;
    int numCU = 0;
    // ---- workspace of gm_map*, grown on demand, reused across calls ----
    uint32_t* d_acc = nullptr; uint64_t accCap = 0;
    uint4* d_stack = nullptr; uint64_t stackCap = 0;       // in uint4 units
    void* d_small = nullptr;                               // counter(8) | error(4) | pad | counters(16)
    uint4* d_table = nullptr; uint64_t tableCap = 0;
    gm::OssRecordL* d_tableL = nullptr; uint64_t tableLCap = 0;   // K > MAX_K (gm_longk.h)
    uint2* d_blocks = nullptr; uint64_t blocksCap = 0;
    uint64_t* d_cumLocal = nullptr; uint64_t cumLocalCap = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool evValid = false;
    // search-kernel time of the most recent calls (gm_map_kernel_times): begin/end events of the kernel, recorded on the call's stream
    static constexpr uint32_t EV_RING = 64;
    hipEvent_t evRing[EV_RING][2] = {};
    uint64_t evCount = 0;
    // every call ends by recording evDone on its stream; the next call makes ITS stream wait for it, so calls on
    // different streams are serialised on the index's shared workspaces instead of racing on them
    hipEvent_t evDone = nullptr;
    bool doneValid = false;
    // gm_map_shard: result buffer kept between calls, compute and copy streams, one event per launch of a share
    void* d_shardOut = nullptr; uint64_t shardOutCap = 0;
    hipStream_t stCompute = nullptr, stCopy = nullptr;
    // correction pass of N-less calls: ONE launch per call (not per piece of a share), on a stream of its own beside the main search
    hipStream_t stCorr = nullptr;
    hipEvent_t evCorrGo = nullptr, evCorrDone = nullptr, evCorrStart = nullptr;
    bool corrPending = false;       // evCorrDone of the running call has not been waited for by the call's stream yet
    hipEvent_t evShard[8] = {};
    // results that go to ordinary (pageable) host memory are staged: DMA into this page-locked ring, then host threads copy
    uint8_t* h_stage = nullptr;
    hipEvent_t evStage[4] = {};
    gm::Tuning tune;
    gm_map_stats stats{};
    // ---- jump patterns and the correction pass of N-less frequency calls (gm_oss.h, gm_engine.h: Env::NLESS) ----
    // ---- the split search (gm_expand.h): node packets, control blocks, progress readback ----
    uint4* d_pkt = nullptr; uint64_t pktCap = 0;   // node packets (X part | Y part), in uint4 units
    void* d_xctl = nullptr; void* d_xprog = nullptr; void* h_xprog = nullptr;              // ExpandCtl, ExpandProgress (gm_kernels.h); 4 page-locked copies
    hipEvent_t evX[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t stXA = nullptr, stXB[2] = {nullptr, nullptr};   // phase A in slice order | the walkers of even / odd slices
    hipEvent_t evXA[2] = {nullptr, nullptr}, evXB[2] = {nullptr, nullptr}, evXGo = nullptr;
    uint32_t* d_wmap = nullptr; uint64_t wmapCap = 0;
    uint32_t lastSlices = 0;           // slices of the last call that took the split search (statistics)
    uint32_t pktUnits = 0;             // 16-byte units per packet of what the buffers hold
    uint32_t* d_patterns = nullptr; uint64_t patternsCap = 0;
    uint4* d_jinfo = nullptr; uint64_t jinfoCap = 0;
    bool nRunsValid = false;
    std::vector<std::pair<uint64_t, uint64_t>> nRuns;     // maximal runs of N of the whole text, sorted
    uint2* d_cblocks = nullptr; uint64_t cblocksCap = 0;   // block list of the text windows that hold N, for (corrK, corrE, corrInfix)
    uint64_t nCBlocks = 0; uint32_t corrK = 0, corrE = 0, corrInfix = 0; bool corrValid = false;
    bool corrTimed = false;    // ev[1], ev[2] bracket the correction pass of the last call
    // the share a caller is delivering piece by piece through gm_map_device (GM_MAP_FLAG_PIECE): what the next piece must continue
    struct PieceState { bool active; uint64_t wholeBegin, wholeEnd, next, textBegin, textLen; uint32_t K, E, chunkIndex, chunkStride; };
    PieceState piece{false, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t pieceIndex = 0;   // gm_map_shard delivers a call in several launches: launch number inside the call (statistics accumulate)
    uint32_t statPieces = 0;   // launches the statistics of the last call cover
    int buildRounds[2] = {0, 0};
};
