/*
 * gm_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never called by the product).
 * See gm_oracle.h for scope and pinning.  Every function that restates reference code cites it.
 * All citations are relative to /root/reference/.
 */
#define _POSIX_C_SOURCE 200809L   /* posix_memalign */
#include "gm_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <omp.h>

enum { SYM_N = 4, SYM_SENT = 5, NLET = 5 };

/* ------------------------------------------------------------------------------------------ */
/* own bidirectional FM index (stands where SeqAn's Index<StringSet, BidirectionalIndex<FMIndex>> */
/* stands in the reference: src/common.hpp:38-52)                                               */
/* ------------------------------------------------------------------------------------------ */

/* one cache line per block: a rank query touches exactly one line (the CPU baseline of bench.py is timed on this layout) */
typedef struct { uint32_t cnt[NLET]; uint32_t pad; uint64_t pl[3]; uint64_t pad2[2]; } rankblk; /* 64 symbols, 64 bytes */

struct gmo_index {
    uint64_t n;           /* symbols incl. sentinels */
    uint32_t nseq;
    uint64_t *cum;        /* nseq+1: sentinel-free cumulative lengths */
    uint8_t *text;        /* sentinel-free concatenation */
    uint8_t *bwt[2];      /* 0 = forward text, 1 = reversed sequences */
    rankblk *rb[2];
    uint64_t C[NLET + 1]; /* C[c] = first row whose suffix starts with letter c; C[NLET] = n */
    uint32_t *sa;         /* forward SA (sentinel-text coordinates) or NULL */
};

static uint32_t g_line_syms = 96;
static uint64_t g_visits, g_lines;
void gmo_set_line_symbols(uint32_t s) { g_line_syms = s ? s : 96; }
void gmo_last_counters(uint64_t *v, uint64_t *l) { if (v) *v = g_visits; if (l) *l = g_lines; }

static void build_rank(gmo_index *ix, int d)
{
    /* two passes over chunks of blocks, both under OpenMP with the same static schedule: letter totals per chunk, then the blocks.
     * The thread that will later share the array also touches its pages first (NUMA first touch on a many-core host). */
    uint64_t n = ix->n, nb = n / 64 + 1;
    rankblk *rb = NULL;
    if (posix_memalign((void **)&rb, 64, nb * sizeof(rankblk)) != 0) { ix->rb[d] = NULL; return; }
    const uint8_t *bwt = ix->bwt[d];
    int nt = omp_get_max_threads();
    if (nt < 1) nt = 1;
    uint64_t nchunks = (uint64_t)nt * 8, per = (nb + nchunks - 1) / nchunks;
    uint64_t *tot = (uint64_t *)calloc((nchunks + 1) * NLET, sizeof(uint64_t));
#pragma omp parallel for schedule(static)
    for (uint64_t c = 0; c < nchunks; ++c) {
        uint64_t b0 = c * per, b1 = b0 + per < nb ? b0 + per : nb, cnt[NLET] = {0, 0, 0, 0, 0};
        for (uint64_t i = b0 * 64; i < b1 * 64 && i < n; ++i) if (bwt[i] < NLET) cnt[bwt[i]]++;
        for (int x = 0; x < NLET; ++x) tot[(c + 1) * NLET + x] = cnt[x];
    }
    for (uint64_t c = 1; c <= nchunks; ++c) for (int x = 0; x < NLET; ++x) tot[c * NLET + x] += tot[(c - 1) * NLET + x];
#pragma omp parallel for schedule(static)
    for (uint64_t c = 0; c < nchunks; ++c) {
        uint64_t b0 = c * per, b1 = b0 + per < nb ? b0 + per : nb;
        uint32_t run[NLET];
        for (int x = 0; x < NLET; ++x) run[x] = (uint32_t)tot[c * NLET + x];
        for (uint64_t b = b0; b < b1; ++b) {
            for (int x = 0; x < NLET; ++x) rb[b].cnt[x] = run[x];
            uint64_t p0 = 0, p1 = 0, p2 = 0;
            for (uint64_t j = 0; j < 64; ++j) {
                uint64_t i = b * 64 + j;
                uint8_t ch = (i < n) ? bwt[i] : SYM_SENT;
                if (ch & 1) p0 |= 1ULL << j;
                if (ch & 2) p1 |= 1ULL << j;
                if (ch & 4) p2 |= 1ULL << j;
                if (i < n && ch < NLET) run[ch]++;
            }
            rb[b].pad = 0; rb[b].pl[0] = p0; rb[b].pl[1] = p1; rb[b].pl[2] = p2; rb[b].pad2[0] = rb[b].pad2[1] = 0;
        }
    }
    ix->rb[d] = rb;
    if (d == 0) {
        uint64_t acc = ix->nseq; /* sentinel suffixes occupy rows [0, nseq) */
        for (int x = 0; x < NLET; ++x) { ix->C[x] = acc; acc += tot[nchunks * NLET + x]; }
        ix->C[NLET] = acc;
    }
    free(tot);
}

static inline void rank5(const rankblk *rb, uint64_t i, uint32_t out[NLET])
{
    const rankblk *b = &rb[i >> 6];
    unsigned off = (unsigned)(i & 63);
    uint64_t m = off ? (~0ULL >> (64 - off)) : 0ULL;
    uint64_t p0 = b->pl[0], p1 = b->pl[1], p2 = b->pl[2];
    uint64_t lo = ~p2 & m;
    out[0] = b->cnt[0] + (uint32_t)__builtin_popcountll(lo & ~p1 & ~p0);
    out[1] = b->cnt[1] + (uint32_t)__builtin_popcountll(lo & ~p1 & p0);
    out[2] = b->cnt[2] + (uint32_t)__builtin_popcountll(lo & p1 & ~p0);
    out[3] = b->cnt[3] + (uint32_t)__builtin_popcountll(lo & p1 & p0);
    out[4] = b->cnt[4] + (uint32_t)__builtin_popcountll(p2 & ~p0 & m);
}

/* suffix array by prefix doubling with counting sorts (Manber & Myers); sym[] has unique sentinels */
static uint32_t *suffix_array(const uint32_t *sym, uint64_t n, uint32_t alpha)
{
    uint32_t *sa = (uint32_t *)malloc(n * 4), *rk = (uint32_t *)malloc(n * 4);
    uint32_t *tmp = (uint32_t *)malloc(n * 4), *rk2 = (uint32_t *)malloc(n * 4);
    uint64_t nb = (n > alpha ? n : alpha) + 1;
    uint32_t *cnt = (uint32_t *)malloc(nb * 4);
    memset(cnt, 0, nb * 4);
    for (uint64_t i = 0; i < n; ++i) cnt[sym[i]]++;
    { uint32_t s = 0; for (uint64_t c = 0; c < alpha; ++c) { uint32_t t = cnt[c]; cnt[c] = s; s += t; } }
    for (uint64_t i = 0; i < n; ++i) sa[cnt[sym[i]]++] = (uint32_t)i;
    rk[sa[0]] = 0;
    for (uint64_t i = 1; i < n; ++i) rk[sa[i]] = rk[sa[i - 1]] + (sym[sa[i]] != sym[sa[i - 1]]);
    for (uint64_t h = 1; rk[sa[n - 1]] != n - 1; h <<= 1) {
        uint64_t k = 0;
        for (uint64_t i = n - h; i < n; ++i) tmp[k++] = (uint32_t)i; /* empty second key sorts first */
        for (uint64_t i = 0; i < n; ++i) if (sa[i] >= h) tmp[k++] = sa[i] - (uint32_t)h;
        uint64_t maxr = (uint64_t)rk[sa[n - 1]] + 1;
        memset(cnt, 0, (maxr + 1) * 4);
        for (uint64_t i = 0; i < n; ++i) cnt[rk[i]]++;
        { uint32_t s = 0; for (uint64_t c = 0; c < maxr; ++c) { uint32_t t = cnt[c]; cnt[c] = s; s += t; } }
        for (uint64_t i = 0; i < n; ++i) sa[cnt[rk[tmp[i]]]++] = tmp[i];
        rk2[sa[0]] = 0;
        for (uint64_t i = 1; i < n; ++i) {
            uint32_t a = sa[i - 1], b = sa[i];
            int same = rk[a] == rk[b] && a + h < n && b + h < n && rk[a + h] == rk[b + h];
            rk2[b] = rk2[a] + !same;
        }
        uint32_t *t = rk; rk = rk2; rk2 = t;
    }
    free(rk); free(rk2); free(tmp); free(cnt);
    return sa;
}

static void set_layout(gmo_index *ix, const uint8_t *codes, const uint64_t *seq_len, uint32_t nseq)
{
    ix->nseq = nseq;
    ix->cum = (uint64_t *)malloc((nseq + 1) * 8);
    ix->cum[0] = 0;
    for (uint32_t s = 0; s < nseq; ++s) ix->cum[s + 1] = ix->cum[s] + seq_len[s];
    ix->n = ix->cum[nseq] + nseq;
    ix->text = (uint8_t *)malloc(ix->cum[nseq] + 1);
    memcpy(ix->text, codes, ix->cum[nseq]);
}

/* text symbols: one sentinel after EVERY sequence, sentinels smallest and ordered by position
 * (cf. src/seqan_libdivsufsort.h:80-91,121-123); reverse index = every sequence reversed, same order
 * (src/indexing.hpp:130). */
static uint8_t *bwt_of(const gmo_index *ix, int rev, uint32_t **sa_out)
{
    uint64_t n = ix->n;
    uint32_t nseq = ix->nseq;
    uint32_t *sym = (uint32_t *)malloc(n * 4);
    uint8_t *tc = (uint8_t *)malloc(n);
    uint64_t p = 0;
    for (uint32_t s = 0; s < nseq; ++s) {
        uint64_t b = ix->cum[s], e = ix->cum[s + 1];
        for (uint64_t i = b; i < e; ++i) {
            uint8_t c = rev ? ix->text[e - 1 - (i - b)] : ix->text[i];
            sym[p] = nseq + c; tc[p] = c; ++p;
        }
        sym[p] = s; tc[p] = SYM_SENT; ++p;
    }
    uint32_t *sa = suffix_array(sym, n, nseq + NLET);
    uint8_t *bwt = (uint8_t *)malloc(n);
    for (uint64_t i = 0; i < n; ++i) bwt[i] = sa[i] ? tc[sa[i] - 1] : tc[n - 1];
    free(sym); free(tc);
    if (sa_out) *sa_out = sa; else free(sa);
    return bwt;
}

gmo_index *gmo_index_build(const uint8_t *codes, const uint64_t *seq_len, uint32_t nseq, int keep_sa)
{
    gmo_index *ix = (gmo_index *)calloc(1, sizeof(*ix));
    set_layout(ix, codes, seq_len, nseq);
    if (ix->n >= 0xFFFFFFFFULL) { gmo_index_free(ix); return NULL; }
    ix->bwt[0] = bwt_of(ix, 0, keep_sa ? &ix->sa : NULL);
    ix->bwt[1] = bwt_of(ix, 1, NULL);
    build_rank(ix, 0); build_rank(ix, 1);
    return ix;
}

gmo_index *gmo_index_from_bwt(const uint8_t *bf, const uint8_t *br, const uint8_t *codes,
                              const uint64_t *seq_len, uint32_t nseq)
{
    gmo_index *ix = (gmo_index *)calloc(1, sizeof(*ix));
    set_layout(ix, codes, seq_len, nseq);
    if (ix->n >= 0xFFFFFFFFULL) { gmo_index_free(ix); return NULL; }
    ix->bwt[0] = (uint8_t *)malloc(ix->n); memcpy(ix->bwt[0], bf, ix->n);
    ix->bwt[1] = (uint8_t *)malloc(ix->n); memcpy(ix->bwt[1], br, ix->n);
    build_rank(ix, 0); build_rank(ix, 1);
    return ix;
}

int gmo_index_adopt_sa(gmo_index *ix, const uint32_t *sa)
{
    if (!ix || !sa) return -1;
    free(ix->sa);
    ix->sa = (uint32_t *)malloc(ix->n * 4);
    if (!ix->sa) return -1;
    memcpy(ix->sa, sa, ix->n * 4);
    return 0;
}

void gmo_index_free(gmo_index *ix)
{
    if (!ix) return;
    free(ix->cum); free(ix->text); free(ix->bwt[0]); free(ix->bwt[1]);
    free(ix->rb[0]); free(ix->rb[1]); free(ix->sa); free(ix);
}
uint64_t gmo_index_size(const gmo_index *ix) { return ix->n; }
const uint8_t *gmo_index_bwt(const gmo_index *ix, int rev) { return ix->bwt[rev ? 1 : 0]; }
const uint32_t *gmo_index_sa(const gmo_index *ix) { return ix->sa; }

/* Bidirectional iterator: SA ranges of the pattern in the forward index (lo[0]) and of the reversed
 * pattern in the reverse index (lo[1]); both have width w.  Root = all rows. */
typedef struct { uint32_t lo[2]; uint32_t w; } biter;

typedef struct { uint64_t visits, lines; } counters;

/* all children of a node in direction d (0: extend LEFT via forward BWT = SeqAn's Fwd tag,
 * 1: extend RIGHT via reverse BWT = SeqAn's Rev tag; call sites src/algo.hpp:61,74,108,145). */
static inline void children(const gmo_index *ix, const biter *it, int d, biter out[NLET], counters *ct)
{
    uint32_t a[NLET], b[NLET];
    uint64_t lo = it->lo[d], hi = (uint64_t)it->lo[d] + it->w;
    rank5(ix->rb[d], lo, a);
    rank5(ix->rb[d], hi, b);
    if (ct) { ct->visits++; ct->lines += 1 + (lo / g_line_syms != hi / g_line_syms); }
    uint32_t tot = 0;
    for (int c = 0; c < NLET; ++c) tot += b[c] - a[c];
    uint32_t smaller = it->w - tot; /* sentinels in BWT[lo,hi) sort before every letter */
    for (int c = 0; c < NLET; ++c) {
        uint32_t cw = b[c] - a[c];
        out[c].lo[d] = (uint32_t)(ix->C[c] + a[c]);
        out[c].lo[1 - d] = it->lo[1 - d] + smaller;
        out[c].w = cw;
        smaller += cw;
    }
}

static inline int go_down_char(const gmo_index *ix, biter *it, uint8_t c, int d, counters *ct)
{
    biter ch[NLET];
    children(ix, it, d, ch, ct);
    if (ch[c].w == 0) return 0; /* SeqAn goDown(it,c,dir): no state change on failure */
    *it = ch[c];
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* Optimum Search Schemes -- src/find2_index_approx.hpp                                        */
/* ------------------------------------------------------------------------------------------ */
#define MAXB 6
#define MAXS 7
typedef struct { uint8_t nb; uint8_t pi[MAXB], l[MAXB], u[MAXB]; uint32_t bl[MAXB]; uint32_t startPos; } srch;
typedef struct { uint8_t ns; srch s[MAXS]; } scheme;

/* tables: src/find2_index_approx.hpp:67-134 */
static const scheme SCHEMES[5] = {
    {1, {{1, {1}, {0}, {0}, {0}, 0}}},
    {2, {{2, {1, 2}, {0, 0}, {0, 1}, {0}, 0},
         {2, {2, 1}, {0, 1}, {0, 1}, {0}, 0}}},
    {3, {{4, {1, 2, 3, 4}, {0, 0, 1, 1}, {0, 0, 2, 2}, {0}, 0},
         {4, {3, 2, 1, 4}, {0, 0, 0, 0}, {0, 1, 1, 2}, {0}, 0},
         {4, {4, 3, 2, 1}, {0, 0, 0, 2}, {0, 1, 2, 2}, {0}, 0}}},
    {4, {{5, {1, 2, 3, 4, 5}, {0, 0, 0, 0, 3}, {0, 1, 2, 3, 3}, {0}, 0},
         {5, {2, 3, 4, 5, 1}, {0, 0, 0, 2, 2}, {0, 1, 2, 2, 3}, {0}, 0},
         {5, {3, 4, 5, 2, 1}, {0, 0, 1, 1, 1}, {0, 1, 1, 3, 3}, {0}, 0},
         {5, {5, 4, 3, 2, 1}, {0, 0, 0, 0, 0}, {0, 0, 3, 3, 3}, {0}, 0}}},
    {7, {{6, {1, 2, 3, 4, 5, 6}, {0, 0, 0, 0, 0, 4}, {0, 2, 3, 3, 4, 4}, {0}, 0},
         {6, {3, 4, 5, 6, 2, 1}, {0, 0, 0, 1, 4, 4}, {0, 0, 1, 1, 4, 4}, {0}, 0},
         {6, {2, 3, 4, 5, 6, 1}, {0, 0, 0, 0, 0, 0}, {0, 2, 2, 3, 3, 4}, {0}, 0},
         {6, {3, 2, 4, 5, 6, 1}, {0, 1, 1, 1, 1, 1}, {0, 1, 2, 3, 3, 4}, {0}, 0},
         {6, {4, 3, 2, 5, 6, 1}, {0, 0, 2, 2, 2, 2}, {0, 0, 2, 3, 3, 4}, {0}, 0},
         {6, {4, 3, 2, 5, 6, 1}, {0, 1, 2, 2, 2, 2}, {0, 1, 2, 3, 3, 4}, {0}, 0},
         {6, {6, 5, 4, 3, 2, 1}, {0, 0, 0, 0, 3, 3}, {0, 0, 4, 4, 4, 4}, {0}, 0}}},
};

/* _optimalSearchSchemeComputeFixedBlocklengthGM + SetBlockLengthGM + InitGM: :139-176 */
static void scheme_set_lengths(scheme *sc, uint32_t needleLength)
{
    uint8_t blocks = sc->s[0].nb;
    uint32_t blocklength = needleLength / blocks;
    uint8_t rest = (uint8_t)(needleLength - blocks * blocklength);
    uint32_t bls[MAXB];
    for (uint8_t i = 0; i < blocks; ++i) bls[i] = blocklength + (i < rest);
    for (uint8_t k = 0; k < sc->ns; ++k) {
        srch *s = &sc->s[k];
        for (uint8_t i = 0; i < s->nb; ++i) s->bl[i] = bls[s->pi[i] - 1] + ((i > 0) ? s->bl[i - 1] : 0);
        s->startPos = 0;
        for (uint8_t i = 0; i < s->nb; ++i)
            if (s->pi[i] < s->pi[0]) s->startPos += s->bl[i] - s->bl[i - 1];
    }
}

/* per-block search context (the lambdas' captures of src/algo.hpp:262-298) */
typedef struct { uint32_t lo, w; } frange;
typedef struct { frange *v; uint32_t n, cap; } flist;

typedef struct {
    const gmo_index *ix;
    const uint8_t *text;     /* slice base (sentinel-free) */
    uint64_t beginPos;       /* needles = text[beginPos, beginPos + W) */
    uint32_t W;              /* K + n - 1 */
    uint32_t K, E, L;        /* L = length of the common infix ("overlap" in :246) */
    int rc;                  /* searching the reverse complement of the window (:286-287) */
    uint64_t max_val;
    uint64_t *hits;          /* n entries, holds TValue-saturated values */
    frange *itExact;         /* n entries */
    flist *itAll;            /* n lists (forward or revcompl pass) */
    int csv;
    uint64_t bb;
    counters ct;
} bctx;

static inline uint8_t needle_at(const bctx *c, uint64_t p)
{
    if (!c->rc) return c->text[c->beginPos + p];
    uint8_t x = c->text[c->beginPos + (c->W - 1 - p)];
    return x < 4 ? (uint8_t)(3 - x) : x; /* complement, N stays N */
}
static inline uint8_t infix_at(const bctx *c, uint32_t q) { return needle_at(c, (uint64_t)(c->K - c->L) + q); }

static void flist_push(flist *f, frange r)
{
    if (f->n == f->cap) { f->cap = f->cap ? f->cap * 2 : 4; f->v = (frange *)realloc(f->v, f->cap * sizeof(frange)); }
    f->v[f->n++] = r;
}

static void extend(bctx *c, biter it, int reportExact, unsigned errorsLeft, uint64_t a, uint64_t b, uint64_t ab, uint64_t bb);

/* src/algo.hpp:26-79 */
static void extendExact(bctx *c, biter it, int reportExact, uint64_t a, uint64_t b, uint64_t ab, uint64_t bb)
{
    uint32_t length = c->K;
    if (b - a + 1 == length) {
        if (reportExact && c->E == 0) { c->itExact[a - ab].lo = it.lo[0]; c->itExact[a - ab].w = it.w; }
        if (c->csv) { frange r = {it.lo[0], it.w}; flist_push(&c->itAll[a - ab], r); }
        uint64_t v = (uint64_t)it.w + c->hits[a - ab];
        c->hits[a - ab] = v < c->max_val ? v : c->max_val;
        return;
    }
    {
        biter it2 = it;
        uint64_t brm = a + length - 1;
        uint64_t b_new = b + (((brm - b) + 2 - 1) >> 1);
        if (b_new <= bb) {
            int success = 1;
            for (uint64_t i = b + 1; i <= b_new && success; ++i) {
                uint8_t ch = needle_at(c, i);
                success = (ch != SYM_N) && go_down_char(c->ix, &it2, ch, 1, &c->ct);
            }
            if (success) extendExact(c, it2, reportExact, a, b_new, ab, bb);
        }
    }
    if (a - 1 >= ab) { /* unsigned, as in the reference; a >= 1 here */
        int64_t alm = (int64_t)b + 1 - length;
        int64_t half = (((int64_t)a - alm) - 1) >> 1;
        uint64_t a_new = (uint64_t)(alm + (half > 0 ? half : 0));
        for (int64_t i = (int64_t)a - 1; i >= (int64_t)a_new; --i) {
            uint8_t ch = needle_at(c, (uint64_t)i);
            if (ch == SYM_N || !go_down_char(c->ix, &it, ch, 0, &c->ct)) return;
        }
        extendExact(c, it, reportExact, a_new, b, ab, bb);
    }
}

/* src/algo.hpp:90-126 (Rev: to the right) and :127-163 (Fwd: to the left) */
static void approxSearch(bctx *c, biter it, int reportExact, unsigned errorsLeft, uint64_t a, uint64_t b,
                         uint64_t ab, uint64_t bb, uint64_t target, int right)
{
    if (right ? (b == target) : (a == target)) { extend(c, it, reportExact, errorsLeft, a, b, ab, bb); return; }
    if (errorsLeft > 0) {
        biter ch[NLET];
        children(c->ix, &it, right, ch, &c->ct);
        uint8_t tc = needle_at(c, right ? b + 1 : a - 1);
        for (int x = 0; x < NLET; ++x) { /* goDown / goRight: non-empty children in alphabet order */
            if (ch[x].w == 0) continue;
            unsigned delta = (x != tc) || (tc == SYM_N);
            if (right) approxSearch(c, ch[x], reportExact, errorsLeft - delta, a, b + 1, ab, bb, target, 1);
            else approxSearch(c, ch[x], reportExact, errorsLeft - delta, a - 1, b, ab, bb, target, 0);
        }
    } else if (right) {
        for (uint64_t i = b + 1; i <= target; ++i) {
            uint8_t t = needle_at(c, i);
            if (t == SYM_N || !go_down_char(c->ix, &it, t, 1, &c->ct)) return;
        }
        extendExact(c, it, reportExact, a, target, ab, bb);
    } else {
        for (int64_t i = (int64_t)a - 1; i >= (int64_t)target; --i) {
            uint8_t t = needle_at(c, (uint64_t)i);
            if (t == SYM_N || !go_down_char(c->ix, &it, t, 0, &c->ct)) return;
        }
        extendExact(c, it, reportExact, target, b, ab, bb);
    }
}

/* src/algo.hpp:165-218 */
static void extend(bctx *c, biter it, int reportExact, unsigned errorsLeft, uint64_t a, uint64_t b, uint64_t ab, uint64_t bb)
{
    uint32_t length = c->K;
    if (errorsLeft == 0) { extendExact(c, it, reportExact, a, b, ab, bb); return; }
    if (b - a + 1 == length) {
        if (reportExact && c->E == errorsLeft) { c->itExact[a - ab].lo = it.lo[0]; c->itExact[a - ab].w = it.w; }
        if (c->csv) { frange r = {it.lo[0], it.w}; flist_push(&c->itAll[a - ab], r); }
        uint64_t v = (uint64_t)it.w + c->hits[a - ab];
        c->hits[a - ab] = v < c->max_val ? v : c->max_val;
        return;
    }
    uint64_t brm = a + length - 1;
    uint64_t b_new = b + (((brm - b) + 2 - 1) >> 1);
    if (b_new <= bb) approxSearch(c, it, reportExact, errorsLeft, a, b, ab, bb, b_new, 1);
    if (a - 1 >= ab) {
        int64_t alm = (int64_t)b + 1 - length;
        int64_t half = (((int64_t)a - alm) - 1) >> 1;
        uint64_t a_new = (uint64_t)(alm + (half > 0 ? half : 0));
        approxSearch(c, it, reportExact, errorsLeft, a, b, ab, bb, a_new, 0);
    }
}

/* delegate / delegateRevCompl: src/algo.hpp:262-298 */
static void delegate(bctx *c, biter it, unsigned errors_spent)
{
    int reportExact = (!c->rc) && errors_spent == 0;
    extend(c, it, reportExact, c->E - errors_spent, c->K - c->L, c->K - 1, 0, c->bb);
}

static void oss(bctx *c, biter it, uint32_t nl, uint32_t nr, uint8_t errors, const srch *s, uint8_t bi, int right);

/* _optimalSearchSchemeChildrenGM (Hamming): src/find2_index_approx.hpp:223-301 */
static void oss_children(bctx *c, biter it, uint32_t nl, uint32_t nr, uint8_t errors, const srch *s, uint8_t bi,
                         uint8_t minErrorsLeftInBlock, int right)
{
    biter ch[NLET];
    children(c->ix, &it, right, ch, &c->ct);
    /* at the needle's end the reference reads one element past the infix (UB); every child is then
     * pruned by :254-258 (charsLeft == 0, minErrorsLeftInBlock > 0), so any value works. */
    uint32_t q = right ? nr - 1 : nl - 1;
    uint8_t needleChar = (q < c->L) ? infix_at(c, q) : 0;
    uint32_t charsLeft = s->bl[bi] - (nr - nl - 1);
    for (int x = 0; x < NLET; ++x) {
        if (ch[x].w == 0) continue;
        unsigned delta = (x != needleChar) || (needleChar == SYM_N);
        if (minErrorsLeftInBlock > 0 && charsLeft + delta < minErrorsLeftInBlock + 1u) continue;
        int32_t nl2 = (int32_t)nl - !right;
        uint32_t nr2 = nr + right;
        if (nr - nl == s->bl[bi]) {
            uint8_t bi2 = (uint8_t)((bi + 1 < s->nb - 1) ? bi + 1 : s->nb - 1);
            int right2 = s->pi[bi2] > s->pi[bi2 - 1];
            oss(c, ch[x], (uint32_t)nl2, nr2, (uint8_t)(errors + delta), s, bi2, right2);
        } else {
            oss(c, ch[x], (uint32_t)nl2, nr2, (uint8_t)(errors + delta), s, bi, right);
        }
    }
}

/* _optimalSearchSchemeExactGM: src/find2_index_approx.hpp:303-369 */
static void oss_exact(bctx *c, biter it, uint32_t nl, uint32_t nr, uint8_t errors, const srch *s, uint8_t bi, int right)
{
    int right2 = (bi < s->nb - 1) && s->pi[bi + 1] > s->pi[bi];
    uint8_t bi2 = (uint8_t)((bi + 1 < s->nb - 1) ? bi + 1 : s->nb - 1);
    if (right) {
        uint32_t infixPosLeft = nr - 1;
        uint32_t infixPosRight = nl + s->bl[bi] - 1;
        while (infixPosLeft <= infixPosRight) {
            uint8_t t = infix_at(c, infixPosLeft);
            if (t == SYM_N || !go_down_char(c->ix, &it, t, 1, &c->ct)) return;
            ++infixPosLeft;
        }
        oss(c, it, nl, infixPosRight + 2, errors, s, bi2, right2);
    } else {
        int32_t infixPosLeft = (int32_t)nr - (int32_t)s->bl[bi] - 1;
        int32_t infixPosRight = (int32_t)nl - 1;
        while (infixPosLeft <= infixPosRight) {
            uint8_t t = infix_at(c, (uint32_t)infixPosRight);
            if (t == SYM_N || !go_down_char(c->ix, &it, t, 0, &c->ct)) return;
            --infixPosRight;
        }
        oss(c, it, (uint32_t)infixPosLeft, nr, errors, s, bi2, right2);
    }
}

/* _optimalSearchSchemeGM: src/find2_index_approx.hpp:371-428 (HammingDistance) */
static void oss(bctx *c, biter it, uint32_t nl, uint32_t nr, uint8_t errors, const srch *s, uint8_t bi, int right)
{
    uint8_t maxErrorsLeftInBlock = (uint8_t)(s->u[bi] - errors);
    uint8_t minErrorsLeftInBlock = (s->l[bi] > errors) ? (uint8_t)(s->l[bi] - errors) : 0;
    if (minErrorsLeftInBlock == 0 && nl == 0 && nr == c->L + 1)
        delegate(c, it, errors);
    else if (maxErrorsLeftInBlock == 0 && nr - nl - 1 != s->bl[bi])
        oss_exact(c, it, nl, nr, errors, s, bi, right);
    else
        oss_children(c, it, nl, nr, errors, s, bi, minErrorsLeftInBlock, right);
}

static void oss_all(bctx *c, const scheme *sc)
{
    biter root; root.lo[0] = 0; root.lo[1] = 0; root.w = (uint32_t)c->ix->n;
    for (uint8_t k = 0; k < sc->ns; ++k) /* :449-457, :441 */
        oss(c, root, sc->s[k].startPos, sc->s[k].startPos + 1, 0, &sc->s[k], 0, 1);
}

/* ------------------------------------------------------------------------------------------ */
/* locate + location harvesting                                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t seq; uint64_t pos; } loc;
static int loc_cmp(const void *a, const void *b)
{
    const loc *x = (const loc *)a, *y = (const loc *)b;
    if (x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}
static loc locate_row(const gmo_index *ix, uint32_t row)
{
    /* sentinel-text position -> (seqNo, seqPos): sequence s starts at cum[s] + s */
    uint64_t p = ix->sa[row];
    uint32_t lo = 0, hi = ix->nseq;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) / 2; if (ix->cum[mid] + mid <= p) lo = mid; else hi = mid; }
    loc r; r.seq = lo; r.pos = p - (ix->cum[lo] + lo);
    return r;
}

typedef struct { loc key; uint32_t nplus, nminus; loc *plus, *minus; uint64_t order; } lentry;
typedef struct { lentry *v; uint64_t n, cap; omp_lock_t lock; } lstore;

static loc *loc_dup(const loc *v, uint32_t n) { loc *r = (loc *)malloc((n ? n : 1) * sizeof(loc)); memcpy(r, v, n * sizeof(loc)); return r; }
static void lstore_add(lstore *st, loc key, const loc *plus, uint32_t np, const loc *minus, uint32_t nm)
{
    omp_set_lock(&st->lock);
    if (st->n == st->cap) { st->cap = st->cap ? st->cap * 2 : 1024; st->v = (lentry *)realloc(st->v, st->cap * sizeof(lentry)); }
    lentry *e = &st->v[st->n];
    e->key = key; e->nplus = np; e->nminus = nm; e->plus = loc_dup(plus, np); e->minus = loc_dup(minus, nm); e->order = st->n;
    st->n++;
    omp_unset_lock(&st->lock);
}
static int lentry_cmp(const void *a, const void *b)
{
    const lentry *x = (const lentry *)a, *y = (const lentry *)b;
    int c = loc_cmp(&x->key, &y->key);
    if (c) return c;
    return x->order < y->order ? -1 : (x->order > y->order);
}

void gmo_locations_free(gmo_locations *L)
{
    if (!L) return;
    free(L->key_seq); free(L->key_pos); free(L->plus_off); free(L->minus_off);
    free(L->plus_seq); free(L->plus_pos); free(L->minus_seq); free(L->minus_pos); free(L);
}

/* ------------------------------------------------------------------------------------------ */
/* computeMappabilitySingleBlock / computeMappability / resetLimits -- src/algo.hpp             */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const gmo_index *ix;
    const gmo_params *p;
    const uint8_t *text; uint64_t textLength;
    uint32_t first_seq, nseq_local;
    const uint64_t *chromCum;   /* local cumulative lengths, nseq_local+1 */
    const uint32_t *seq_file_id;
    void *c; uint64_t max_val;
    lstore *locs;
    uint32_t infix;             /* params.overlap after src/mappability.hpp:543 */
} mctx;

static inline uint64_t c_get(const mctx *m, uint64_t i) { return m->p->value_bits == 8 ? ((uint8_t *)m->c)[i] : ((uint16_t *)m->c)[i]; }
static inline void c_set(const mctx *m, uint64_t i, uint64_t v) { if (m->p->value_bits == 8) ((uint8_t *)m->c)[i] = (uint8_t)v; else ((uint16_t *)m->c)[i] = (uint16_t)v; }

static uint32_t harvest(const gmo_index *ix, const flist *f, loc **out)
{
    uint64_t size = 0;
    for (uint32_t k = 0; k < f->n; ++k) size += f->v[k].w;
    loc *v = (loc *)malloc((size ? size : 1) * sizeof(loc));
    uint64_t q = 0;
    for (uint32_t k = 0; k < f->n; ++k)
        for (uint32_t r = 0; r < f->v[k].w; ++r) v[q++] = locate_row(ix, f->v[k].lo + r);
    qsort(v, q, sizeof(loc), loc_cmp);
    *out = v;
    return (uint32_t)q;
}

/* src/algo.hpp:221-403 */
static void single_block(mctx *m, uint64_t i, uint64_t j, int have_intervals, int completeSameKmers, counters *ct)
{
    const gmo_params *p = m->p;
    uint32_t K = p->K;
    uint64_t maxPos = i + K - m->infix;
    if (m->textLength - K < maxPos) maxPos = m->textLength - K;
    maxPos += 1;
    if (maxPos > j) maxPos = j;
    uint64_t beginPos = i;
    while (beginPos < maxPos && c_get(m, beginPos) != 0) ++beginPos;
    uint64_t endPos = maxPos;
    while (i > 0 && endPos - 1 >= i && c_get(m, endPos - 1) != 0) --endPos;
    if (!(beginPos < endPos)) return;

    uint32_t n = (uint32_t)(endPos - beginPos);
    uint32_t overlap = K - n + 1;
    scheme sc = SCHEMES[p->E];
    scheme_set_lengths(&sc, overlap);

    bctx c; memset(&c, 0, sizeof(c));
    c.ix = m->ix; c.text = m->text; c.beginPos = beginPos; c.W = K + n - 1; c.K = K; c.E = p->E; c.L = overlap;
    c.max_val = m->max_val; c.csv = p->csv || p->exclude_pseudo;
    c.hits = (uint64_t *)calloc(n, sizeof(uint64_t));
    c.itExact = (frange *)calloc(n, sizeof(frange));
    flist *itAll = (flist *)calloc(n, sizeof(flist)), *itAllrc = (flist *)calloc(n, sizeof(flist));
    uint64_t bb = (uint64_t)K - 1 + K - overlap;
    if (m->textLength - 1 < bb) bb = m->textLength - 1;
    c.bb = bb;

    if (p->revcompl) {
        c.rc = 1; c.itAll = itAllrc;
        oss_all(&c, &sc);
        for (uint32_t a = 0, b = n - 1; a < b; ++a, --b) { uint64_t t = c.hits[a]; c.hits[a] = c.hits[b]; c.hits[b] = t; } /* :304 */
    }
    c.rc = 0; c.itAll = itAll;
    oss_all(&c, &sc);
    ct->visits += c.ct.visits; ct->lines += c.ct.lines;

    for (uint64_t jj = beginPos; jj < endPos; ++jj) {
        uint32_t t = (uint32_t)(jj - beginPos);
        if (c.csv) { /* :311-387 */
            loc *plus, *minus;
            uint32_t np = harvest(m->ix, &itAll[t], &plus);
            uint32_t nm = harvest(m->ix, &itAllrc[endPos - 1 - jj], &minus);
            if (p->exclude_pseudo) { /* :351-364: number of distinct fasta files, narrowing store */
                uint32_t distinct = 0, cap = np + nm + 1;
                uint32_t *seen = (uint32_t *)malloc(cap * 4);
                for (uint32_t k = 0; k < np + nm; ++k) {
                    uint32_t f = m->seq_file_id[k < np ? plus[k].seq : minus[k - np].seq];
                    int found = 0;
                    for (uint32_t q = 0; q < distinct; ++q) if (seen[q] == f) { found = 1; break; }
                    if (!found) seen[distinct++] = f;
                }
                free(seen);
                c.hits[t] = p->value_bits == 8 ? (uint8_t)distinct : (uint16_t)distinct;
            }
            if (p->csv) {
                if (!p->directory && c.itExact[t].w > 1) {
                    for (uint32_t r = 0; r < c.itExact[t].w; ++r) {
                        loc o = locate_row(m->ix, c.itExact[t].lo + r);
                        uint64_t len = m->ix->cum[o.seq + 1] - m->ix->cum[o.seq];
                        if ((int64_t)o.pos <= (int64_t)len - (int64_t)K) lstore_add(m->locs, o, plus, np, minus, nm);
                    }
                } else if (np + nm > 0) {
                    /* myPosLocalize(entry.first, j, chromCumLengths): src/common.hpp:21-28 */
                    uint32_t lo = 0, hi = m->nseq_local;
                    while (hi - lo > 1) { uint32_t mid = (lo + hi) / 2; if (m->chromCum[mid] <= jj) lo = mid; else hi = mid; }
                    /* upper_bound - 1: last limit <= pos (duplicate limits cannot occur: empty sequences are skipped) */
                    loc key; key.seq = lo; key.pos = jj - m->chromCum[lo];
                    uint64_t len = m->chromCum[lo + 1] - m->chromCum[lo];
                    if ((int64_t)key.pos <= (int64_t)len - (int64_t)K) lstore_add(m->locs, key, plus, np, minus, nm);
                }
            }
            free(plus); free(minus);
        }
        if (p->use_shortcut && !p->directory && (!have_intervals || completeSameKmers) && c.itExact[t].w > 1) { /* :389-396 */
            for (uint32_t r = 0; r < c.itExact[t].w; ++r) {
                loc o = locate_row(m->ix, c.itExact[t].lo + r);
                c_set(m, m->ix->cum[o.seq] + o.pos, c.hits[t]); /* posGlobalize(occ, limits) */
            }
        } else {
            c_set(m, jj, c.hits[t]);
        }
    }
    for (uint32_t t = 0; t < n; ++t) { free(itAll[t].v); free(itAllrc[t].v); }
    free(itAll); free(itAllrc); free(c.hits); free(c.itExact);
}

/* src/algo.hpp:10-22 */
static void resetLimits(mctx *m)
{
    for (uint64_t i = 1; i < (uint64_t)m->nseq_local + 1; ++i) {
        uint64_t lim = m->chromCum[i] - m->chromCum[i - 1] + 1;
        if (m->p->K < lim) lim = m->p->K;
        for (uint64_t j = 1; j < lim; ++j) c_set(m, m->chromCum[i] - j, 0);
    }
}

int gmo_default_infix_length(uint32_t K, uint32_t E, int32_t xo)
{
    /* src/mappability.hpp:519-543 */
    unsigned overlap;
    if (xo >= 0) overlap = (unsigned)xo;
    else if (E == 0) overlap = (unsigned)(K * 0.7);
    else {
        unsigned mm = K > 30u ? K : 30u; if (mm > 100u) mm = 100u;
        overlap = (unsigned)((K * mm) * pow((double)0.7f, (double)E) / 100.0);
    }
    unsigned m1 = K - 1, m2 = K - E - 2; /* unsigned wrap as in the reference */
    uint64_t maxPossibleOverlap = m1 < m2 ? m1 : m2;
    if (overlap > maxPossibleOverlap) {
        if (xo < 0) overlap = (unsigned)maxPossibleOverlap; else return -1;
    }
    return (int)(K - overlap);
}

/* (bench.py's cpu_baseline: the caller hands in a zeroed vector and clears what a call wrote, so that a timed call on a sample of a
   3 Gbp text does not spend a second clearing 3 GB on one thread) */
static int g_skip_clear = 0;
void gmo_set_skip_clear(int on) { g_skip_clear = on; }

int gmo_compute_mappability(const gmo_index *ix, uint64_t text_begin, uint64_t text_len,
                            uint32_t first_seq, uint32_t nseq_local, const gmo_params *p,
                            const uint64_t *intervals, uint64_t n_intervals, const uint32_t *seq_file_id,
                            void *out, int *csk_out, gmo_locations **locs_out)
{
    if (p->E > 4) return -2;                     /* "E > 4 not yet supported." src/mappability.hpp:187 */
    if (p->value_bits != 8 && p->value_bits != 16) return -3;
    if ((p->csv || p->exclude_pseudo || p->use_shortcut) && !ix->sa) return -4;
    int infix = p->infix > 0 ? p->infix : gmo_default_infix_length(p->K, p->E, p->overlap);
    if (infix < 0) return -5;
    if (!g_skip_clear) memset(out, 0, text_len * (p->value_bits / 8));
    if (csk_out) *csk_out = 0;
    if (locs_out) *locs_out = NULL;

    mctx m; memset(&m, 0, sizeof(m));
    m.ix = ix; m.p = p; m.text = ix->text + text_begin; m.textLength = text_len;
    m.first_seq = first_seq; m.nseq_local = nseq_local; m.seq_file_id = seq_file_id;
    uint64_t *cum = (uint64_t *)malloc((nseq_local + 1) * 8);
    for (uint32_t s = 0; s <= nseq_local; ++s) cum[s] = ix->cum[first_seq + s] - ix->cum[first_seq];
    m.chromCum = cum; m.c = out; m.max_val = p->value_bits == 8 ? 255 : 65535; m.infix = (uint32_t)infix;
    lstore st; memset(&st, 0, sizeof(st)); omp_init_lock(&st.lock); m.locs = &st;

    int threads = p->threads > 0 ? p->threads : 1;
    uint64_t visits = 0, lines = 0;
    int completeSameKmers = 0;
    if (text_len >= p->K) { /* the reference underflows numberOfKmers for textLength < K (src/algo.hpp:414) */
        uint64_t numberOfKmers = text_len - p->K + 1;
        uint64_t stepSize = p->K - (uint64_t)infix + 1;
        if (n_intervals == 0) {
            uint64_t chunk = numberOfKmers / (stepSize * threads * 50); if (chunk < 1) chunk = 1;
            uint64_t nblocks = (numberOfKmers + stepSize - 1) / stepSize;
            #pragma omp parallel for schedule(dynamic, chunk) num_threads(threads) reduction(+:visits, lines)
            for (uint64_t b = 0; b < nblocks; ++b) {
                counters ct = {0, 0};
                single_block(&m, b * stepSize, b * stepSize + stepSize, 0, 1, &ct);
                visits += ct.visits; lines += ct.lines;
            }
        } else { /* src/algo.hpp:441-476 */
            uint64_t interval_sum = 0, nd = 0, cap = 0; uint64_t *det = NULL;
            for (uint64_t k = 0; k < n_intervals; ++k) {
                uint64_t f = intervals[2 * k], s = intervals[2 * k + 1];
                interval_sum += s - f;
                for (uint64_t q = f; q < s; q += stepSize) {
                    if (nd == cap) { cap = cap ? cap * 2 : 64; det = (uint64_t *)realloc(det, cap * 16); }
                    det[2 * nd] = q; det[2 * nd + 1] = (q + stepSize < s) ? q + stepSize : s; ++nd;
                }
            }
            float fraction = (float)interval_sum / text_len;
            completeSameKmers = fraction > 0.5f;
            uint64_t chunk = nd / ((uint64_t)threads * 50); if (chunk < 1) chunk = 1;
            #pragma omp parallel for schedule(dynamic, chunk) num_threads(threads) reduction(+:visits, lines)
            for (uint64_t k = 0; k < nd; ++k) {
                counters ct = {0, 0};
                single_block(&m, det[2 * k], det[2 * k + 1], 1, completeSameKmers, &ct);
                visits += ct.visits; lines += ct.lines;
            }
            free(det);
        }
    }
    resetLimits(&m);
    g_visits = visits; g_lines = lines;
    if (csk_out) *csk_out = completeSameKmers;

    if (locs_out && p->csv) {
        qsort(st.v, st.n, sizeof(lentry), lentry_cmp);
        gmo_locations *L = (gmo_locations *)calloc(1, sizeof(*L));
        uint64_t ne = 0, tp = 0, tm = 0;
        for (uint64_t k = 0; k < st.n; ++k)
            if (k == 0 || loc_cmp(&st.v[k].key, &st.v[k - 1].key) != 0) { ne++; tp += st.v[k].nplus; tm += st.v[k].nminus; }
        L->n_entries = ne;
        L->key_seq = (uint32_t *)malloc((ne + 1) * 4); L->key_pos = (uint64_t *)malloc((ne + 1) * 8);
        L->plus_off = (uint64_t *)malloc((ne + 1) * 8); L->minus_off = (uint64_t *)malloc((ne + 1) * 8);
        L->plus_seq = (uint32_t *)malloc((tp + 1) * 4); L->plus_pos = (uint64_t *)malloc((tp + 1) * 8);
        L->minus_seq = (uint32_t *)malloc((tm + 1) * 4); L->minus_pos = (uint64_t *)malloc((tm + 1) * 8);
        uint64_t e = 0, op = 0, om = 0;
        for (uint64_t k = 0; k < st.n; ++k) {
            if (!(k == 0 || loc_cmp(&st.v[k].key, &st.v[k - 1].key) != 0)) continue; /* std::map::emplace keeps the first */
            L->key_seq[e] = st.v[k].key.seq; L->key_pos[e] = st.v[k].key.pos;
            L->plus_off[e] = op; L->minus_off[e] = om;
            for (uint32_t q = 0; q < st.v[k].nplus; ++q) { L->plus_seq[op] = st.v[k].plus[q].seq; L->plus_pos[op] = st.v[k].plus[q].pos; ++op; }
            for (uint32_t q = 0; q < st.v[k].nminus; ++q) { L->minus_seq[om] = st.v[k].minus[q].seq; L->minus_pos[om] = st.v[k].minus[q].pos; ++om; }
            ++e;
        }
        L->plus_off[ne] = op; L->minus_off[ne] = om;
        *locs_out = L;
    }
    for (uint64_t k = 0; k < st.n; ++k) { free(st.v[k].plus); free(st.v[k].minus); }
    free(st.v); omp_destroy_lock(&st.lock); free(cum);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* tests/tests.cpp:30-131                                                                       */
/* ------------------------------------------------------------------------------------------ */
static void trivial_bt(const gmo_index *ix, biter it, const uint8_t *needle, uint32_t pos, uint32_t K,
                       unsigned errors, unsigned threshold, uint64_t *frequency)
{
    if (errors == threshold) { /* :39-49 */
        while (pos < K) {
            if (needle[pos] == SYM_N || !go_down_char(ix, &it, needle[pos], 1, NULL)) break;
            ++pos;
        }
        if (pos == K) *frequency += it.w;
    } else if (errors < threshold) {
        if (pos == K) { *frequency += it.w; return; } /* :55-58 */
        biter ch[NLET];
        children(ix, &it, 1, ch, NULL);
        for (int x = 0; x < NLET; ++x) { /* :69-85 */
            if (ch[x].w == 0) continue;
            unsigned delta = (needle[pos] == SYM_N) || (x != needle[pos]);
            trivial_bt(ix, ch[x], needle, pos + 1, K, errors + delta, threshold, frequency);
        }
    }
}

int gmo_trivial_backtracking(const gmo_index *ix, uint32_t K, uint32_t E, int revcompl, int value_bits, void *out)
{
    uint64_t max_val = value_bits == 8 ? 255 : 65535;
    uint64_t total = ix->cum[ix->nseq];
    memset(out, 0, total * (value_bits / 8));
    uint8_t *rc = (uint8_t *)malloc(K + 1);
    for (uint32_t s = 0; s < ix->nseq; ++s) { /* :116-130 */
        uint64_t len = ix->cum[s + 1] - ix->cum[s];
        if (len < K) continue;
        for (uint64_t i = 0; i + K <= len; ++i) {
            const uint8_t *needle = ix->text + ix->cum[s] + i;
            biter root; root.lo[0] = 0; root.lo[1] = 0; root.w = (uint32_t)ix->n;
            uint64_t f = 0;
            trivial_bt(ix, root, needle, 0, K, 0, E, &f);
            uint64_t hits = f < max_val ? f : max_val;
            if (revcompl && hits < max_val) {
                for (uint32_t t = 0; t < K; ++t) { uint8_t x = needle[K - 1 - t]; rc[t] = x < 4 ? (uint8_t)(3 - x) : x; }
                f = 0;
                trivial_bt(ix, root, rc, 0, K, 0, E, &f);
                hits = hits + f < max_val ? hits + f : max_val;
            }
            if (value_bits == 8) ((uint8_t *)out)[ix->cum[s] + i] = (uint8_t)hits; else ((uint16_t *)out)[ix->cum[s] + i] = (uint16_t)hits;
        }
    }
    free(rc);
    return 0;
}

int gmo_brute_force(const uint8_t *codes, const uint64_t *seq_len, uint32_t nseq, uint64_t text_begin, uint64_t text_len,
                    uint32_t K, uint32_t E, int revcompl, int value_bits, void *out)
{
    uint64_t max_val = value_bits == 8 ? 255 : 65535;
    uint64_t *cum = (uint64_t *)malloc((nseq + 1) * 8);
    cum[0] = 0;
    for (uint32_t s = 0; s < nseq; ++s) cum[s + 1] = cum[s] + seq_len[s];
    memset(out, 0, text_len * (value_bits / 8));
    uint8_t *pat = (uint8_t *)malloc(2 * (K + 1));
    for (uint32_t s = 0; s < nseq; ++s) {
        for (uint64_t i = 0; i + K <= seq_len[s]; ++i) {
            uint64_t g = cum[s] + i;
            if (g < text_begin || g >= text_begin + text_len) continue;
            for (uint32_t t = 0; t < K; ++t) {
                pat[t] = codes[g + t];
                uint8_t x = codes[g + K - 1 - t]; pat[K + 1 + t] = x < 4 ? (uint8_t)(3 - x) : x;
            }
            uint64_t total = 0;
            for (int strand = 0; strand < (revcompl ? 2 : 1); ++strand) {
                const uint8_t *P = pat + strand * (K + 1);
                for (uint32_t s2 = 0; s2 < nseq; ++s2)
                    for (uint64_t i2 = 0; i2 + K <= seq_len[s2]; ++i2) {
                        const uint8_t *T = codes + cum[s2] + i2;
                        uint32_t mm = 0;
                        for (uint32_t t = 0; t < K && mm <= E; ++t) mm += (P[t] == SYM_N) || (P[t] != T[t]);
                        total += mm <= E;
                    }
            }
            if (total > max_val) total = max_val;
            if (value_bits == 8) ((uint8_t *)out)[g - text_begin] = (uint8_t)total; else ((uint16_t *)out)[g - text_begin] = (uint16_t)total;
        }
    }
    free(pat); free(cum);
    return 0;
}
