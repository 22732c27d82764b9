/*
 * genmap_amd.h -- C ABI of the MI355X-native (k,e)-mappability engine (libgenmap_amd.so).
 *
 * The reference (cpockrandt/genmap) has no plugin/FFI interface; its de-facto boundary for the hot path
 * is the function
 *     computeMappability<errors>(index, text, c, params, directory, chromLengths, chromCumLengths,
 *                                locations, mappingSeqIdFile, intervals, completeSameKmers, ...)
 *     /root/reference/src/algo.hpp:405-410      callers: src/mappability.hpp:177-185, tests/tests.cpp:192-193
 * plus the index it is handed (built by src/indexing.hpp:73-148, opened by src/genmap_helper.hpp:72-98).
 * Every entry point below names the reference interface it replaces.  Plain pointers and sizes only;
 * no C++/torch types.  All functions return 0 (GM_OK) or a negative gm_status and never exit().
 * The library needs a HIP device: without one every compute entry point returns GM_ERR_NO_DEVICE
 * (there is no CPU fallback).
 *
 * Symbol codes in every buffer: A=0 C=1 G=2 T=3 N=4 (non-ACGTU characters are N, src/indexing.hpp:13-20);
 * BWT buffers additionally use 5 for the per-sequence sentinel.
 */
#ifndef GENMAP_AMD_H
#define GENMAP_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum gm_status {
    GM_OK = 0,
    GM_ERR_NO_DEVICE = -1,     /* no usable HIP device / HIP runtime error at start-up */
    GM_ERR_BAD_ERRORS = -2,    /* E > 4: "E > 4 not yet supported." src/mappability.hpp:187 */
    GM_ERR_BAD_VALUE_BITS = -3,/* value_bits must be 8 (-fs) or 16 (-fl, mappability) src/mappability.hpp:388-394 */
    GM_ERR_NEED_LOCATE = -4,   /* csv / --exclude-pseudo requested on an index without SA samples */
    GM_ERR_BAD_OVERLAP = -5,   /* -xo larger than min(K-1, K-E-2), src/mappability.hpp:528-540; or infix < #blocks */
    GM_ERR_BAD_K = -6,         /* K < 1 or K > 32768 (K <= 255: the persistent kernel with 16-byte nodes; longer k-mers: the plain tree walk, gm_longk.h) */
    GM_ERR_TOO_LONG = -7,      /* beyond the build's limits: 2^40 rows; a gm_locate window of 2^31 occurrences */
    GM_ERR_BAD_ARG = -8,
    GM_ERR_HIP = -9,           /* a HIP call failed; gm_last_error() has the text */
    GM_ERR_IO = -10,
    GM_ERR_OOM = -11,
    GM_ERR_INTERNAL = -12      /* a device-side invariant was violated (e.g. lane stack bound) */
} gm_status;

const char *gm_status_string(int status);
const char *gm_last_error(void);          /* thread-local text of the last failing call */
int gm_device_count(void);                /* number of HIP devices visible, 0 if none */

/* ------------------------------------------------------------------------------------------------
 * Index  (replaces Index<StringSet, BidirectionalIndex<FMIndex>>: src/common.hpp:38-52;
 *         construction src/indexing.hpp:73-148 + src/seqan_libdivsufsort.h:36-240;
 *         open() src/genmap_helper.hpp:72-98)
 * ---------------------------------------------------------------------------------------------- */
typedef struct gm_index gm_index;

typedef struct gm_index_info {
    uint64_t n_rows;          /* text symbols + one sentinel per sequence */
    uint64_t text_len;        /* sum of sequence lengths */
    uint32_t n_seq;
    uint32_t sampling;        /* SA sampling rate (0 = no samples, counts only) */
    uint32_t alphabet_size;   /* 4 or 5 (index.info "alphabet_size", src/indexing.hpp:102) */
    uint32_t block_bytes;     /* rank block size of this index: 32, 64 or 128 */
    uint64_t device_bytes;    /* HBM held by the index */
    int32_t  device;
    uint32_t row_bits;        /* 32, or 64 for indexes of 2^32 - 1 rows or more (and with GM_BLOCK_WIDE_ROWS) */
    uint32_t verify_records;  /* 1: the index holds one 32-byte record {SA[row], 56 text symbols around it} per row (built with
                                 sampling 1 when HBM allows): a narrow search node is verified with ONE read */
} gm_index_info;

/* Build both FM indexes ON THE GPU from host sequences (concatenated codes, no sentinels).
 * block_bytes: 32, 64 or 128 (0 = library default).  sampling: keep SA[i] where the in-sequence offset
 * is a multiple of `sampling` (src/seqan_libdivsufsort.h:129-143); 0 = none. */
/* OR this into block_bytes to force 64-bit rows on a small index (tests).  Indexes of 2^32 - 1 rows or more get them by
 * themselves (the reference's 64-bit BWT variants, src/indexing.hpp:158-169, src/mappability.hpp:373-385): 64-byte rank blocks
 * of 64 symbols with 64-bit counts, 64-bit suffix array, 32-byte search nodes; sa_fwd of gm_index_import / gm_index_export_sa is
 * then an array of uint64_t.  csv / --exclude-pseudo additionally need every single sequence to be shorter than 2^32. */
#define GM_BLOCK_WIDE_ROWS 0x10000u
int gm_index_build(const uint8_t *codes, const uint64_t *seq_len, uint32_t n_seq,
                   uint32_t sampling, uint32_t block_bytes, int device, gm_index **out);

/* Adopt BWTs that already exist on the host (an index file read from disk, or another builder):
 * uploads them and packs the rank blocks on the GPU.  sa_fwd may be NULL (no locate); otherwise it holds one entry per row,
 * sa_entry_bytes wide: 4 (uint32_t) for indexes of fewer than 2^32 - 1 rows, 8 (uint64_t) for wider ones and for indexes
 * forced wide with GM_BLOCK_WIDE_ROWS.  A width that does not match the index is GM_ERR_BAD_ARG (it used to be a silent cast). */
int gm_index_import(const uint8_t *bwt_fwd, const uint8_t *bwt_rev, const void *sa_fwd, uint32_t sa_entry_bytes,
                    const uint8_t *codes, const uint64_t *seq_len, uint32_t n_seq,
                    uint32_t sampling, uint32_t block_bytes, int device, gm_index **out);

/* The sampled form of the suffix array (sampling rate s in 2..64, the reference's -S, src/indexing.hpp:311-315 and
 * src/seqan_libdivsufsort.h:129-143): SA[row] is kept where its in-sequence offset is a multiple of s (sequence starts always
 * are); mark_words holds one bit per row (bit r % 32 of word r / 32), samples the kept values in row order.  locate walks the
 * LF mapping until it meets a marked row (at most s - 1 rank-block reads).  With sampling > 1 narrow search nodes are not
 * verified against the text (that needs one read per row, not a walk).  Works for 32- and 64-bit rows alike (the reference's
 * default -S 10 on every index it accepts); samples are sa_entry_bytes wide (4, or 8 with 64-bit rows), fewer than 2^32 of them.
 * gm_index_build / gm_index_import take sampling > 1 directly (the full array is built or uploaded, sampled on the device and
 * released); these two entries move the sampled form itself, e.g. to and from an index directory. */
int gm_index_import_sampled(const uint8_t *bwt_fwd, const uint8_t *bwt_rev, const uint32_t *mark_words, const void *samples, uint32_t sa_entry_bytes,
                            uint64_t n_samples, const uint8_t *codes, const uint64_t *seq_len, uint32_t n_seq,
                            uint32_t sampling, uint32_t block_bytes, int device, gm_index **out);
/* samples == NULL: only *n_samples is set.  mark_words: ceil(n_rows / 32) words. */
int gm_index_export_sa_sampled(const gm_index *idx, uint32_t *mark_words, void *samples, uint32_t sa_entry_bytes, uint64_t *n_samples);

/* Copy the BWTs (codes 0..5, n_rows bytes each) back to the host: index files, parity tests. */
int gm_index_export_bwt(const gm_index *idx, uint8_t *bwt_fwd, uint8_t *bwt_rev);

/* Copy the forward suffix array (n_rows x uint32, sentinel-text positions; uint64 for wide indexes) back to the host; needs sampling 1. */
int gm_index_export_sa(const gm_index *idx, void *sa_fwd, uint32_t sa_entry_bytes);   /* entries as wide as gm_index_info.row_bits says */

int gm_index_get_info(const gm_index *idx, gm_index_info *info);
void gm_index_free(gm_index *idx);

/* ------------------------------------------------------------------------------------------------
 * computeMappability  (src/algo.hpp:405-483, for ONE fasta file's slice of the concatenated text:
 * the loop body of src/mappability.hpp:289-365)
 * ---------------------------------------------------------------------------------------------- */
typedef struct gm_map_params {
    uint32_t K;               /* -K  SearchParams.length */
    uint32_t E;               /* -E  0..4 */
    int32_t  overlap;         /* hidden -xo (src/mappability.hpp:463-465).  < 0 = not given: the library then uses its own
                                 MI355X-tuned block shape, gm_tuned_infix_length(); results do not depend on it.  The
                                 reference's default is available as infix = gm_default_infix_length(K, E, -1). */
    int32_t  infix;           /* > 0: set the common-infix length (SearchParams.overlap) directly, as
                                 tests/tests.cpp:179-181 does; overrides `overlap` */
    int32_t  revcompl;        /* 0 with -nc */
    int32_t  value_bits;      /* 8 or 16 */
    int32_t  exclude_pseudo;  /* -ep: count distinct fasta files (needs SA samples) */
    int32_t  flags;           /* GM_MAP_FLAG_* */
    uint64_t kmer_begin;      /* shard: compute only k-mer start positions in [kmer_begin, kmer_end) of   */
    uint64_t kmer_end;        /*        the slice (whole k-mer blocks).  Both 0 without GM_MAP_FLAG_RANGE =   */
                              /*        everything; with the flag the range is taken literally (an empty    */
                              /*        range computes nothing).  gm_map leaves the other positions zero,     */
                              /*        gm_map_device does not touch them (apart from the boundary reset),  */
                              /*        so shards can share one device buffer.                              */
    /* interleaved chunks (multi-GPU load balance; the reference deals >= 50 dynamic chunks per worker because "repeats are
     * slower than unique regions", src/algo.hpp:422-434): with chunk_blocks > 0 and chunk_stride > 1 the k-mer blocks of the
     * range are cut into chunks of chunk_blocks whole blocks (a block = K - infix + 1 consecutive k-mers) and this call
     * computes the chunks whose number is congruent to chunk_index modulo chunk_stride.  gm_map_device touches only the
     * positions of those chunks.  Not supported by gm_locate. */
    uint32_t chunk_blocks, chunk_index, chunk_stride, reserved1;
    /* GM_MAP_FLAG_PIECE: this call is one launch of a larger share [whole_begin, whole_end) of k-mer positions (same chunk arguments) that the
     * caller delivers in several gm_map_device calls on ONE stream, first piece first -- so that the transfer of one piece overlaps the search
     * of the next (what gm_map_shard does inside one call).  The piece that begins at whole_begin clears the accumulators of the whole share and
     * runs the call's ONE correction pass (the text windows with N, searched beside the main kernel); the later pieces do neither.  Round 4
     * ran that pass behind every launch: a fixed cost per rank that did not shrink with the number of GPUs. */
    uint64_t whole_begin, whole_end;
} gm_map_params;
#define GM_MAP_FLAG_RANGE 1   /* [kmer_begin, kmer_end) is a shard even when it is empty or (0,0) */
#define GM_MAP_FLAG_PIECE 2   /* [kmer_begin, kmer_end) is one launch of the share [whole_begin, whole_end) (implies GM_MAP_FLAG_RANGE) */

/* Concurrency: an index owns ONE set of device workspaces (work counter, accumulators, lane stacks, per-call tables).
 * Calls on the same index may be issued from one host thread at a time, on any streams: every call makes its stream
 * wait for the end-of-call event of the previous call on that index, so they execute one after the other on the device.
 * gm_map_device returns without synchronising; device-side invariant violations are reported by the next
 * synchronising call (gm_map, gm_map_runs, gm_locate, gm_last_map_stats, gm_index_sync). */

/* text_begin/text_len: the slice in sentinel-free global coordinates (src/mappability.hpp:312);
 * first_seq/n_seq: its sequences (for resetLimits, src/algo.hpp:10-22);
 * intervals: n_intervals half-open pairs in slice coordinates or NULL (-S, src/mappability.hpp:334-357);
 * seq_file_id: fasta id per GLOBAL sequence (only read with exclude_pseudo, src/mappability.hpp:230-248);
 * out: text_len values of value_bits width, caller-owned HOST memory (gm_map) or DEVICE memory on the
 * index's device (gm_map_device; stream = hipStream_t or NULL).  Values are bit-identical to the
 * reference's c[] after resetLimits and the selection reset of src/mappability.hpp:83-99. */
int gm_map(gm_index *idx, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq,
           const gm_map_params *params, const uint64_t *intervals, uint64_t n_intervals,
           const uint32_t *seq_file_id, void *out_host);

int gm_map_device(gm_index *idx, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq,
                  const gm_map_params *params, const uint64_t *intervals, uint64_t n_intervals,
                  const uint32_t *seq_file_id, void *out_device, void *stream);

/* The per-FASTA-file loop of the reference (src/mappability.hpp:289-365: one computeMappability per file of the index) as ONE call:
 * n_files files, file i made of the sequences [file_first_seq[i], file_first_seq[i] + file_n_seq[i]), consecutive and in ascending order;
 * out_host[i]: the values of file i (its text length, value_bits wide).  The files' k-mers are searched in one launch of the persistent
 * kernel -- a value depends on the k-mer at its position and on the whole index, not on the file it is computed with, and the positions that
 * cross a sequence (hence a file) boundary are zeroed by resetLimits either way -- instead of filling and draining the device once per
 * file (config C5: five launches of 6.4 ms for 4 Mbp each).  No selection (intervals) and no shard arguments; gm_map per file is the
 * general form. */
int gm_map_files(gm_index *idx, uint32_t n_files, const uint32_t *file_first_seq, const uint32_t *file_n_seq,
                 const gm_map_params *params, const uint32_t *seq_file_id, void *const *out_host);

/* One device's share of a computeMappability call whose result is assembled in HOST memory by several devices (the
 * `genmap map -D 0,1,..` path: one index replica and one host thread per GPU).  Computes the interleaved chunks selected by
 * params->chunk_blocks / chunk_index / chunk_stride (all of the k-mer range when chunk_stride <= 1) and copies exactly
 * those chunks' positions into out_host -- the other bytes are not touched, so the devices fill one vector concurrently
 * and nothing is merged on the CPU.  The share is processed in a few launches; the copy of one launch's chunks
 * (device -> host over PCIe, DMA engines) runs while the next launch computes.  out_host should be pinned
 * (gm_host_pin) for the copies to be asynchronous; pageable memory works but is staged by the runtime. */
int gm_map_shard(gm_index *idx, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq,
                 const gm_map_params *params, const uint64_t *intervals, uint64_t n_intervals,
                 const uint32_t *seq_file_id, void *out_host);
int gm_host_pin(void *host, uint64_t bytes);      /* hipHostRegister / hipHostUnregister */
int gm_host_unpin(void *host);

/* One process per GPU (torch.distributed launchers): the root's result vector is shared with the other ranks through a HIP IPC
 * handle and every rank pushes its finished chunks straight into it -- device-to-device copies over xGMI by the DMA engines,
 * on their own stream, while the rank's search kernel works on its next chunks (the persistent kernel holds every CU, a
 * collective's kernel could not run beside it).  gm_device_alloc returns a base allocation (IPC handles name allocations, not
 * pointers inside a framework's caching allocator).  handle = 64 bytes (hipIpcMemHandle_t). */
int gm_device_alloc(int device, uint64_t bytes, void **dptr);
int gm_device_free(int device, void *dptr);
int gm_ipc_export(int device, void *dptr, uint8_t handle[64]);
int gm_ipc_open(int device, const uint8_t handle[64], void **dptr);
int gm_ipc_close(int device, void *dptr);
/* copy n_rows pieces of piece_bytes each, pitch_bytes apart (the chunks of one shard), from src to the same offsets of dst:
 * dst + first_byte + i * pitch_bytes <- src + first_byte + i * pitch_bytes.  One asynchronous copy per piece on `stream`
 * (1-D copies go to the DMA engines; a 2-D copy would be a blit kernel that waits for free CUs).  The last piece may be
 * shortened with last_piece_bytes (0 = full). */
int gm_push_pieces(int device, void *dst, const void *src, uint64_t first_byte, uint64_t pitch_bytes, uint64_t piece_bytes,
                   uint64_t n_rows, uint64_t last_piece_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Locations for csv output  (the `locations` map filled by src/algo.hpp:311-387 and consumed by
 * saveCsv, src/output.hpp:189-288).  For every slice position j in [pos_begin, pos_begin + n_positions):
 * the sorted occurrences of the k-mer at j on the + strand, plus[plus_off[j']..plus_off[j'+1]), and of its
 * reverse complement, minus[...], j' = j - pos_begin.  An occurrence is (seqNo << 32 | seqPos) with the
 * GLOBAL sequence number of the index.  Positions that were not computed (outside the selection / shard)
 * have empty lists.  Needs an index with sampling 1.  The window is params->kmer_begin/kmer_end rounded
 * to whole k-mer blocks (both 0 = the whole slice); windows holding >= 2^31 occurrences are refused with
 * GM_ERR_TOO_LONG -- split the range.  Host arrays are owned by the library: gm_locations_free().
 * ---------------------------------------------------------------------------------------------- */
typedef struct gm_locations {
    uint64_t pos_begin, n_positions;
    uint64_t *plus_off, *minus_off;   /* n_positions + 1 each */
    uint64_t *plus, *minus;
} gm_locations;

int gm_locate(gm_index *idx, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq,
              const gm_map_params *params, const uint64_t *intervals, uint64_t n_intervals, gm_locations **out);
void gm_locations_free(gm_locations *loc);

/* ------------------------------------------------------------------------------------------------
 * Run-length form of the result for wig / bedgraph / bed output (the scans of saveWig / saveBedGraph,
 * src/output.hpp:74-187, done on the GPU): maximal runs of equal NON-ZERO value that do not cross a
 * sequence boundary, in text order.  Same arguments as gm_map; only the runs cross PCIe.
 * ---------------------------------------------------------------------------------------------- */
typedef struct gm_runs {
    uint64_t n_runs;
    uint64_t *start;      /* slice position of the run's first value */
    uint64_t *length;
    uint16_t *value;      /* the frequency (8-bit results are widened) */
} gm_runs;

int gm_map_runs(gm_index *idx, uint64_t text_begin, uint64_t text_len, uint32_t first_seq, uint32_t n_seq,
                const gm_map_params *params, const uint64_t *intervals, uint64_t n_intervals,
                const uint32_t *seq_file_id, gm_runs **out);
void gm_runs_free(gm_runs *runs);

/* counters of the most recent gm_map* call on this index (for the roofline numerator) */
typedef struct gm_map_stats {
    uint64_t kmers;           /* k-mer positions searched */
    uint64_t roots;           /* (block, strand, search) work items */
    uint64_t node_steps;      /* bidirectional extensions evaluated (0 unless the library was built with GM_COUNTERS) */
    uint64_t rank_lines;      /* distinct rank blocks those steps read (same) */
    uint64_t detail[48];      /* GM_COUNTERS only: steps in OSS phase, in extension phase, extension steps at range
                                 width 1, at width 2..4, OSS steps at width 1, nodes pushed to lane stacks,
                                 verification items, of which in OSS phase, 8-symbol comparison chunks,
                                 wavefront iterations, lanes holding a node summed over iterations, verification rounds,
                                 shader cycles (summed over wavefronts) in root fetch, verification, stepping;
                                 [15..19]: cycles in pop, work sharing, fetch stages 3+2, fetch stage 1 (parts of "root fetch"), stolen nodes;
                                 [20..35]: how often a WAVEFRONT executed a code region (any lane enabled): pop, saturation
                                 test, work-sharing exchange, fetch stage 3, 2, 1, deferral round, split, mismatch round,
                                 leaf, leaf flush, verification: OSS block, 8-symbol chunk, mismatch event, k-mer loop
                                 iteration; stack push past the LDS levels;
                                 [36] table reads of jump patterns; [37] correction pass in microseconds, [38] longest q-mer table of
                                 the call | jump length << 8 (both set by the host in every build); [39] one-row table entries ended
                                 by the neighbour filter; [40] rows located (one suffix-array or mark-word read each: --exclude-pseudo, csv,
                                 correction pass), [41] LF steps of sampled suffix-array walks, [42] deepest lane stack of the call,
                                 [43] self hits, [44] verified runs of k-mers, [45] 8-byte bitmap words read for groups of jump patterns, [46] rows of two-row
                                 table entries ended by the neighbour filter; [47] bits 0..47: node packets written by phase A of the split search
                                 (gm_expand.h), bits 48..63: slices the call's split search took (set by the host in every build; 0: the one-loop kernel ran) */
    double   search_ms;       /* HIP-event time of the search kernel alone */
    double   total_ms;        /* memset + search + finalize, HIP events on the call's stream */
} gm_map_stats;
int gm_last_map_stats(const gm_index *idx, gm_map_stats *stats);

/* search-kernel durations (ms, HIP events recorded on each call's own stream) of the last min(n, 64) gm_map* calls on this
 * index, oldest first; *n_out = how many were written.  Synchronises with those calls.  (bench.py: roofline denominator
 * over the timed steps themselves.) */
int gm_map_kernel_times(const gm_index *idx, double *ms, uint32_t n, uint32_t *n_out);

/* wait until every call issued on this index has finished; GM_ERR_INTERNAL if one of them tripped a device-side check */
int gm_index_sync(gm_index *idx);

/* scheduling knobs of the search kernel, for sweeps and tests.  Names: verify_t, lds_stack, blocks_per_cu, qtable,
 * sat_min_w, fetch_batch, probation, verify_cost, skip_dup, coop, use_ctx, jump, jump_filter, self_hit, steal (0: no work sharing inside a
 * wavefront, n > 0: an exchange when at least n lanes are idle), part_bias (e = 1: characters moved from the second OSS block
 * to the first; every split gives the same result; may be negative, default 0), oss_weights (e >= 1: nibble i = relative length of
 * OSS block i, left to right; 0 = the reference's equal split); qtable / jump = 1..16 force the length of the q-mer table / of the jump
 * patterns (16: the 69 GB table of all 16-mers, the default beyond 2^30 rows), jump_filter = 0 switches the neighbour test of one- and
 * two-row table entries off (2: one-row entries only), jump_groups = 0 / 1 forbids / forces the groups of jump patterns behind the
 * bitmaps (default: where they are expected to save table reads), range_add = 0 adds verified runs k-mer by k-mer instead of through the
 * difference plane, verify_t_ext = widest node verified in the extension phase; expand = 0 / 1 forbids / forces the split search (phase A writes
 * node packets, a walker draws them: gm_expand.h; default: frequency calls with errors at K < 64 over 2^20 roots or more), expand_mb = its packet
 * buffers in MiB, expand_two_pass / expand_share / expand_chunk / sat_draw_w = its schedule; win2 = 0 / 1 forbids / forces needle windows staged
 * from a 2-bit copy of the text (default: K >= 64 where that saves LDS).  Results never depend on these.
 * Two TEST-ONLY knobs do change the output: no_saturate = 1 counts without the min(total, MAX) clamp and stores the low bits,
 * no_store = 1 only switches e = 0 from plain stores to the atomic accumulators (same result).
 * value -1 restores the library default of any knob except part_bias; values outside a knob's range are GM_ERR_BAD_ARG.
 * Nothing is read from the environment. */
int gm_index_set_tuning(gm_index *idx, const char *name, int64_t value);

/* reference default of SearchParams.overlap (common-infix length) for (K,E,-xo): src/mappability.hpp:519-543.
 * Returns 0 if -xo is too large. */
uint32_t gm_default_infix_length(uint32_t K, uint32_t E, int32_t xo);
/* common-infix length this build schedules with when neither -xo nor params.infix is given (0 for invalid K/E) */
uint32_t gm_tuned_infix_length(uint32_t K, uint32_t E);
/* ... for the calls that locate their occurrences: gm_map* with params.exclude_pseudo, gm_locate (shorter blocks: those kernels walk the
 * search tree from its root).  A caller that cuts such a call into shares of whole blocks (kmer_begin / kmer_end) aligns them with this one. */
uint32_t gm_tuned_infix_length_locating(uint32_t K, uint32_t E);

#ifdef __cplusplus
}
#endif
#endif
