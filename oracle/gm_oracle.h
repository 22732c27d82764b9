/*
 * gm_oracle.h -- CPU ORACLE for the (k,e)-mappability hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libgenmap_amd.so) never
 * links, loads or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference algorithm
 *   /root/reference/src/algo.hpp                 (computeMappability, SingleBlock, extend, approxSearch, extendExact, resetLimits)
 *   /root/reference/src/find2_index_approx.hpp   (Optimum Search Schemes, Hamming distance paths)
 *   /root/reference/tests/tests.cpp:30-131       (trivial backtracking = the definition the reference tests against)
 * on top of an own, simple bidirectional FM index (the reference gets this from SeqAn,
 * an un-vendored submodule that is absent from /root/reference).
 *
 * PARITY PINNING: the reference binary cannot be built here (SeqAn missing).  The oracle is
 * pinned against all 18 end-to-end fixture cases of the reference (tests/golden/reference_cases,
 * copied data files of /root/reference/tests/test_cases) and against an index-free brute-force
 * Hamming counter implementing the definition of tests/tests.cpp:105-131.
 *
 * Symbol codes: A=0 C=1 G=2 T=3 N=4 ; BWT arrays additionally use 5 = sentinel.
 */
#ifndef GM_ORACLE_H
#define GM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gmo_index gmo_index;

/* Build the bidirectional FM index over all sequences (concatenated codes, no sentinels in the
 * input).  keep_sa != 0 keeps the full forward suffix array (needed for locate: csv,
 * --exclude-pseudo and the duplicate-copy shortcut of src/algo.hpp:389-396). */
gmo_index *gmo_index_build(const uint8_t *codes, const uint64_t *seq_len, uint32_t n_seq, int keep_sa);

/* Adopt BWTs computed elsewhere (e.g. downloaded from the GPU builder) -- used by bench.py's
 * cpu_baseline leg so that a large index need not be suffix-sorted on the CPU.  bwt arrays have
 * n_total = sum(seq_len)+n_seq entries with codes 0..5.  No SA => no locate. */
gmo_index *gmo_index_from_bwt(const uint8_t *bwt_fwd, const uint8_t *bwt_rev,
                              const uint8_t *codes, const uint64_t *seq_len, uint32_t n_seq);

/* Adopt a forward suffix array for an index made by gmo_index_from_bwt (sentinel-text positions, n_total entries) so that
 * locate (csv, --exclude-pseudo) works without suffix-sorting a large text on the CPU.  The caller vouches for it: the tests
 * check sa against the adopted BWT (bwt[i] == symbol before sa[i], LF-walk consistency) before using it.  Returns 0. */
int gmo_index_adopt_sa(gmo_index *, const uint32_t *sa_fwd);

void gmo_index_free(gmo_index *);

/* Accessors used by tests to compare the product's builder against the oracle's. */
uint64_t gmo_index_size(const gmo_index *);                 /* n_total */
const uint8_t *gmo_index_bwt(const gmo_index *, int rev);   /* codes 0..5 */
const uint32_t *gmo_index_sa(const gmo_index *);            /* forward SA over the sentinel text, or NULL */

typedef struct gmo_params {
    uint32_t K;
    uint32_t E;               /* 0..4 */
    int32_t  overlap;         /* value of the hidden -xo option; <0 = reference default (src/mappability.hpp:519-525) */
    int32_t  revcompl;        /* 1 unless -nc */
    int32_t  value_bits;      /* 8 (-fs) or 16 (-fl / mappability) */
    int32_t  directory;       /* index built with -FD */
    int32_t  exclude_pseudo;  /* -ep */
    int32_t  csv;             /* -d: harvest locations */
    int32_t  threads;         /* OpenMP threads */
    int32_t  use_shortcut;    /* 1 = keep the duplicate-copy shortcut + trimming of src/algo.hpp:236-242,389-396 (needs SA) */
    int32_t  infix;           /* >0: set SearchParams.overlap (= common-infix length) directly, as tests/tests.cpp:179-181 does */
} gmo_params;

/* Location lists for csv (flattened std::map of src/mappability.hpp:168-170). Entries sorted by key. */
typedef struct gmo_locations {
    uint64_t n_entries;
    uint32_t *key_seq;   /* i1 */
    uint64_t *key_pos;   /* i2 */
    uint64_t *plus_off;  /* n_entries+1 offsets into plus_* */
    uint64_t *minus_off; /* n_entries+1 offsets into minus_* */
    uint32_t *plus_seq;  uint64_t *plus_pos;
    uint32_t *minus_seq; uint64_t *minus_pos;
} gmo_locations;
void gmo_locations_free(gmo_locations *);

/* computeMappability<E>(...) of src/algo.hpp:405-483 for ONE fasta file's slice of the concatenated
 * text: [text_begin, text_begin+text_len) in sentinel-free global coordinates; first_seq = index of
 * the slice's first sequence, n_seq_local sequences.  intervals: pairs (begin,end) relative to the
 * slice (src/mappability.hpp:334-357) or NULL.  seq_file_id: fasta id per global sequence (only
 * read with exclude_pseudo).  out: text_len values of value_bits width, zeroed by the callee.
 * complete_same_kmers_out mirrors the reference's out-parameter.  Returns 0 or a negative error. */
int gmo_compute_mappability(const gmo_index *idx, uint64_t text_begin, uint64_t text_len,
                            uint32_t first_seq, uint32_t n_seq_local,
                            const gmo_params *p,
                            const uint64_t *intervals, uint64_t n_intervals,
                            const uint32_t *seq_file_id,
                            void *out, int *complete_same_kmers_out,
                            gmo_locations **locations_out);

/* computeMappabilityTrivial of tests/tests.cpp:105-131 via _trivialBacktracking (:30-87) on the
 * oracle's index: whole index, all sequences, out has sum(seq_len) entries. */
int gmo_trivial_backtracking(const gmo_index *idx, uint32_t K, uint32_t E, int revcompl, int value_bits, void *out);

/* Index-free definition: for every position of every sequence count text windows at Hamming
 * distance <= E (N in the k-mer always mismatches), plus reverse complement. O(n^2 K): tiny inputs only.
 * Counts over ALL sequences, values for the slice [text_begin, text_begin+text_len). */
int gmo_brute_force(const uint8_t *codes, const uint64_t *seq_len, uint32_t n_seq,
                    uint64_t text_begin, uint64_t text_len,
                    uint32_t K, uint32_t E, int revcompl, int value_bits, void *out);

/* reference default for params.overlap (the common-infix length), src/mappability.hpp:519-543.
 * xo < 0: not set.  Returns the infix length K - overlap, or -1 if xo is too large (PARSE_ERROR). */
int gmo_default_infix_length(uint32_t K, uint32_t E, int32_t xo);

/* Instrumentation for the roofline numerator: node visits (= bidirectional extensions evaluated)
 * and distinct 64-B rank lines of the LAST gmo_compute_mappability call (summed over threads). */
void gmo_last_counters(uint64_t *node_visits, uint64_t *rank_lines);
/* symbols per rank line of the layout being priced (default 96 = the product's 64-B block). */
void gmo_set_line_symbols(uint32_t syms);

/* 1: gmo_compute_mappability does not clear `out` (the caller hands in zeros); measurement only.  Process-wide and not
 * thread-safe: set it, time, and reset it (bench.py does so in a try / finally). */
void gmo_set_skip_clear(int on);

#ifdef __cplusplus
}
#endif

#endif
