/*
 * apd_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See apd_oracle.h.
 * PARITY UNPINNED (no reference tests/fixtures exist, Rust toolchain absent).
 *
 * Every function cites the reference lines it restates.  Compile with
 * -ffp-contract=off so that a*b+c stays two roundings as in the Rust build.
 */
#include "apd_oracle.h"

#include <math.h>
#include <omp.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_INF ((float)INFINITY)

/* ------------------------------------------------------------------ numerics */

/* numerics.rs:114-120: distance += powf(x[i]-y[i], 2.0); sqrt(distance).
 * powf(v, 2.0) is exactly v*v in IEEE f32. */
float orc_euclidean(const float *x, const float *y, uint32_t dim)
{
    float distance = 0.0f;
    for (uint32_t i = 0; i < dim; i++) {
        float d = x[i] - y[i];
        float sq = d * d;
        distance = distance + sq;
    }
    return sqrtf(distance);
}

/* Rust `f32 as usize`: saturating, NaN -> 0. */
static uint64_t f32_as_usize(float v)
{
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)v;
}

/* discovery.rs:38-45: (self.warping_band_percentage * n_size as f32) as usize */
uint64_t orc_warping_band(float pct, uint64_t n_size)
{
    float prod = pct * (float)n_size;
    return f32_as_usize(prod);
}

/* numerics.rs:138-155 */
static uint64_t abs_usize(uint64_t n, uint64_t m) { return n > m ? n - m : m - n; }
static uint64_t diff_usize(uint64_t n, uint64_t m) { return n > m ? n - m : 0; }
static uint64_t max_u64(uint64_t a, uint64_t b) { return a > b ? a : b; }
static uint64_t min_u64(uint64_t a, uint64_t b) { return a < b ? a : b; }

/* alignments.rs:173-175: w = max(band, |n-m|) + 2; i in 1..=n; j in max(i-w,1)..min(i+w, m+1) */
uint64_t orc_dtw_cells(uint64_t n, uint64_t m, uint64_t band)
{
    uint64_t w = max_u64(band, abs_usize(n, m)) + 2;
    uint64_t cells = 0;
    for (uint64_t i = 1; i <= n; i++) {
        uint64_t lo = max_u64(diff_usize(i, w), 1);
        uint64_t hi = min_u64(i + w, m + 1);
        if (hi > lo) cells += hi - lo;
    }
    return cells;
}

/* alignments.rs:153-159: the select.  Ties (and NaN) fall through to MATCH. */
static inline float select_node(float match_score, float insert_score, float delete_score,
                                float distance, float ins_pen, float del_pen, float match_pen)
{
    if (delete_score < match_score && delete_score < insert_score) {
        float p = del_pen * distance;
        return delete_score + p;
    } else if (insert_score < match_score && insert_score < delete_score) {
        float p = ins_pen * distance;
        return insert_score + p;
    } else {
        float p = match_pen * distance;
        return match_score + p;
    }
}

/* alignments.rs:107-180, dense restatement.  The HashMap of the reference holds
 * (0,0)->0 plus every visited cell; a lookup of anything else yields +INF
 * (alignments.rs:139-152).  Two rolling rows with explicit INF sentinels at the
 * band edges reproduce exactly those lookups. */
float orc_dtw_pair(const float *x, uint64_t n, const float *y, uint64_t m, uint32_t dim,
                   uint64_t band, float ins_pen, float del_pen, float match_pen)
{
    if (n == 0 && m == 0) return ORC_INF;         /* alignments.rs:117-118 */
    if (n == 0 || m == 0) return NAN;             /* usize underflow at :120 -- undefined */
    uint64_t w = max_u64(band, abs_usize(n, m)) + 2;   /* :173 */
    float *buf = (float *)malloc(sizeof(float) * 2 * (m + 3));
    if (!buf) return NAN;
    float *prev = buf, *cur = buf + (m + 3);
    /* row 0: only (0,0) exists (alignments.rs:109) */
    for (uint64_t j = 0; j < m + 3; j++) { prev[j] = ORC_INF; cur[j] = ORC_INF; }
    prev[0] = 0.0f;
    /* value of cell (n-1, m-1) if present (alignments.rs:120) */
    int have = 0;
    float cell = 0.0f;
    if (n - 1 == 0 && m - 1 == 0) { have = 1; cell = 0.0f; }
    for (uint64_t i = 1; i <= n; i++) {           /* :174 */
        uint64_t lo = max_u64(diff_usize(i, w), 1);
        uint64_t hi = min_u64(i + w, m + 1);      /* exclusive, :175 */
        cur[lo - 1] = ORC_INF;                    /* (i, lo-1) never visited; (i,0) absent for i>=1 */
        const float *xi = x + (i - 1) * (uint64_t)dim;
        for (uint64_t j = lo; j < hi; j++) {
            float distance = orc_euclidean(xi, y + (j - 1) * (uint64_t)dim, dim);  /* :137 */
            float match_score = prev[j - 1];      /* :139 */
            float insert_score = prev[j];         /* :144 */
            float delete_score = cur[j - 1];      /* :149 */
            cur[j] = select_node(match_score, insert_score, delete_score, distance,
                                 ins_pen, del_pen, match_pen);
        }
        if (hi > lo) cur[hi] = ORC_INF; else cur[lo] = ORC_INF;   /* (i, hi) not visited */
        if (i == n - 1 && m - 1 >= lo && m - 1 < hi) { have = 1; cell = cur[m - 1]; }
        float *t = prev; prev = cur; cur = t;
    }
    free(buf);
    if (!have) return ORC_INF;                    /* :122 */
    return cell / (float)(n + m);                 /* :121 */
}

/* ---------------------------------------------- hash-map ("reference-like" cost) */

typedef struct { uint64_t ki, kj; float v; uint8_t used; } hm_slot;
typedef struct { hm_slot *slots; uint64_t cap, len; } hm_map;

#define ROTL(x, b) (uint64_t)(((x) << (b)) | ((x) >> (64 - (b))))
#define SIPROUND do { \
    v0 += v1; v1 = ROTL(v1, 13); v1 ^= v0; v0 = ROTL(v0, 32); \
    v2 += v3; v3 = ROTL(v3, 16); v3 ^= v2; \
    v0 += v3; v3 = ROTL(v3, 21); v3 ^= v0; \
    v2 += v1; v1 = ROTL(v1, 17); v1 ^= v2; v2 = ROTL(v2, 32); } while (0)

/* SipHash-1-3 of the 16-byte key (i,j), as Rust's DefaultHasher does for (usize,usize). */
static uint64_t sip13_pair(uint64_t a, uint64_t b)
{
    const uint64_t k0 = 0x0706050403020100ULL, k1 = 0x0f0e0d0c0b0a0908ULL;
    uint64_t v0 = k0 ^ 0x736f6d6570736575ULL, v1 = k1 ^ 0x646f72616e646f6dULL;
    uint64_t v2 = k0 ^ 0x6c7967656e657261ULL, v3 = k1 ^ 0x7465646279746573ULL;
    v3 ^= a; SIPROUND; v0 ^= a;
    v3 ^= b; SIPROUND; v0 ^= b;
    uint64_t tail = (uint64_t)16 << 56;
    v3 ^= tail; SIPROUND; v0 ^= tail;
    v2 ^= 0xff; SIPROUND; SIPROUND; SIPROUND;
    return v0 ^ v1 ^ v2 ^ v3;
}

static void hm_init(hm_map *h) { h->cap = 0; h->len = 0; h->slots = NULL; }
static void hm_free(hm_map *h) { free(h->slots); }
static void hm_put_raw(hm_slot *s, uint64_t cap, uint64_t i, uint64_t j, float v)
{
    uint64_t p = sip13_pair(i, j) & (cap - 1);
    while (s[p].used && !(s[p].ki == i && s[p].kj == j)) p = (p + 1) & (cap - 1);
    s[p].used = 1; s[p].ki = i; s[p].kj = j; s[p].v = v;
}
static void hm_insert(hm_map *h, uint64_t i, uint64_t j, float v)
{
    if (h->cap == 0 || (h->len + 1) * 8 > h->cap * 7) {       /* grow at 7/8 load like hashbrown */
        uint64_t ncap = h->cap ? h->cap * 2 : 4;
        hm_slot *ns = (hm_slot *)calloc(ncap, sizeof(hm_slot));
        for (uint64_t p = 0; p < h->cap; p++)
            if (h->slots[p].used) hm_put_raw(ns, ncap, h->slots[p].ki, h->slots[p].kj, h->slots[p].v);
        free(h->slots);
        h->slots = ns; h->cap = ncap;
    }
    hm_put_raw(h->slots, h->cap, i, j, v);
    h->len++;
}
static int hm_get(const hm_map *h, uint64_t i, uint64_t j, float *v)
{
    if (!h->cap) return 0;
    uint64_t p = sip13_pair(i, j) & (h->cap - 1);
    while (h->slots[p].used) {
        if (h->slots[p].ki == i && h->slots[p].kj == j) { *v = h->slots[p].v; return 1; }
        p = (p + 1) & (h->cap - 1);
    }
    return 0;
}

/* alignments.rs:107-180 with the sparse map kept (the literal structure). */
float orc_dtw_pair_hashmap(const float *x, uint64_t n, const float *y, uint64_t m, uint32_t dim,
                           uint64_t band, float ins_pen, float del_pen, float match_pen)
{
    if (n == 0 && m == 0) return ORC_INF;
    if (n == 0 || m == 0) return NAN;
    hm_map sparse; hm_init(&sparse);
    hm_insert(&sparse, 0, 0, 0.0f);                               /* :109 */
    uint64_t w = max_u64(band, abs_usize(n, m)) + 2;              /* :173 */
    for (uint64_t i = 1; i <= n; i++) {
        uint64_t lo = max_u64(diff_usize(i, w), 1), hi = min_u64(i + w, m + 1);
        for (uint64_t j = lo; j < hi; j++) {
            float distance = orc_euclidean(x + (i - 1) * (uint64_t)dim, y + (j - 1) * (uint64_t)dim, dim);
            float ms, is, ds;
            if (!hm_get(&sparse, i - 1, j - 1, &ms)) ms = ORC_INF;
            if (!hm_get(&sparse, i - 1, j, &is)) is = ORC_INF;
            if (!hm_get(&sparse, i, j - 1, &ds)) ds = ORC_INF;
            hm_insert(&sparse, i, j, select_node(ms, is, ds, distance, ins_pen, del_pen, match_pen));
        }
    }
    float cell, out;
    if (hm_get(&sparse, n - 1, m - 1, &cell)) out = cell / (float)(n + m);
    else out = ORC_INF;
    hm_free(&sparse);
    return out;
}

/* --------------------------------------------------------------- all pairs */

typedef struct {
    const float *frames; const uint64_t *offsets; uint32_t n_seq, dim;
    float pct, ins, del, mat; int use_hashmap; float *out;
    uint64_t start, stop;                 /* row block (align_all) or pair range (sample) */
    const uint32_t *pi, *pj; uint64_t cells;
} worker_arg;

static float one_pair(const worker_arg *a, uint32_t i, uint32_t j, uint64_t *cells)
{
    uint64_t li = a->offsets[i + 1] - a->offsets[i], lj = a->offsets[j + 1] - a->offsets[j];
    uint64_t len = max_u64(li, lj);                                   /* alignments.rs:52 */
    uint64_t band = orc_warping_band(a->pct, len);                    /* :53 */
    const float *x = a->frames + a->offsets[i] * (uint64_t)a->dim;
    const float *y = a->frames + a->offsets[j] * (uint64_t)a->dim;
    if (cells) *cells += orc_dtw_cells(li, lj, band);
    return a->use_hashmap
        ? orc_dtw_pair_hashmap(x, li, y, lj, a->dim, band, a->ins, a->del, a->mat)
        : orc_dtw_pair(x, li, y, lj, a->dim, band, a->ins, a->del, a->mat);
}

static void *align_rows(void *p)
{
    worker_arg *a = (worker_arg *)p;
    for (uint64_t i = a->start; i < a->stop; i++)                     /* :42 */
        for (uint32_t j = 0; j < a->n_seq; j++)                       /* :50 */
            if (i != j)                                               /* :51 */
                a->out[i * a->n_seq + j] = one_pair(a, (uint32_t)i, j, NULL);   /* :57 */
    return NULL;
}

int orc_align_all(const float *frames, const uint64_t *offsets, uint32_t n_seq, uint32_t dim,
                  float band_pct, float ins_pen, float del_pen, float match_pen,
                  uint32_t workers, int use_hashmap, float *out)
{
    if (!frames || !offsets || !out || workers == 0 || dim == 0) return -1;
    uint64_t n = n_seq;
    memset(out, 0, sizeof(float) * n * n);                            /* :21-23 */
    uint64_t batch_size = n / workers + 1;                            /* :33 */
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * workers);
    worker_arg *args = (worker_arg *)calloc(workers, sizeof(worker_arg));
    for (uint32_t b = 0; b < workers; b++) {                          /* :35 */
        worker_arg *a = &args[b];
        a->frames = frames; a->offsets = offsets; a->n_seq = n_seq; a->dim = dim;
        a->pct = band_pct; a->ins = ins_pen; a->del = del_pen; a->mat = match_pen;
        a->use_hashmap = use_hashmap; a->out = out;
        a->start = b * batch_size;                                    /* :36 */
        a->stop = min_u64((b + 1) * batch_size, n);                   /* :37 */
        pthread_create(&th[b], NULL, align_rows, a);
    }
    for (uint32_t b = 0; b < workers; b++) pthread_join(th[b], NULL); /* :64-66 */
    free(th); free(args);
    return 0;
}

static void *align_sample(void *p)
{
    worker_arg *a = (worker_arg *)p;
    for (uint64_t k = a->start; k < a->stop; k++)
        a->out[k] = one_pair(a, a->pi[k], a->pj[k], &a->cells);
    return NULL;
}

int orc_align_sample(const float *frames, const uint64_t *offsets, uint32_t n_seq, uint32_t dim,
                     float band_pct, float ins_pen, float del_pen, float match_pen,
                     const uint32_t *pi, const uint32_t *pj, uint64_t n_pairs,
                     uint32_t workers, int use_hashmap, float *out, uint64_t *cells)
{
    if (!frames || !offsets || !out || !pi || !pj || workers == 0 || dim == 0) return -1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * workers);
    worker_arg *args = (worker_arg *)calloc(workers, sizeof(worker_arg));
    uint64_t per = (n_pairs + workers - 1) / workers;
    for (uint32_t b = 0; b < workers; b++) {
        worker_arg *a = &args[b];
        a->frames = frames; a->offsets = offsets; a->n_seq = n_seq; a->dim = dim;
        a->pct = band_pct; a->ins = ins_pen; a->del = del_pen; a->mat = match_pen;
        a->use_hashmap = use_hashmap; a->out = out; a->pi = pi; a->pj = pj;
        a->start = min_u64(b * per, n_pairs); a->stop = min_u64((b + 1) * per, n_pairs);
        pthread_create(&th[b], NULL, align_sample, a);
    }
    uint64_t total = 0;
    for (uint32_t b = 0; b < workers; b++) { pthread_join(th[b], NULL); total += args[b].cells; }
    if (cells) *cells = total;
    free(th); free(args);
    return 0;
}

/* -------------------------------------------------------------- percentile */

static int cmp_f32(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* numerics.rs:125-133 */
int orc_percentile(const float *x, uint64_t len, float perc, float *value)
{
    float nf = (float)len * perc;                                     /* :126 */
    float *numbers = (float *)malloc(sizeof(float) * (len ? len : 1));
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < len; i++) if (x[i] == x[i]) numbers[cnt++] = x[i];   /* :127-130 */
    qsort(numbers, cnt, sizeof(float), cmp_f32);                      /* :131 */
    uint64_t idx = f32_as_usize(nf);                                  /* :132 */
    int rc = 0;
    if (idx < cnt) *value = numbers[idx]; else rc = -1;               /* index panic */
    free(numbers);
    return rc;
}

/* -------------------------------------------------------------- clustering */

typedef struct { uint64_t *parents; uint64_t n_par, cap; const float *d; uint32_t n; uint32_t n_clusters; } dendro;

static uint64_t cl_root(const dendro *g, uint64_t i)                  /* clustering.rs:115-121 */
{
    uint64_t p = i;
    while (p != g->parents[p]) p = g->parents[p];
    return p;
}

static float cl_linkage(const dendro *g, const uint64_t *assign, uint64_t i, uint64_t j)   /* :153-170 */
{
    float size_x = 0.0f, size_y = 0.0f, distance = 0.0f;
    for (uint32_t x = 0; x < g->n; x++) {
        if (assign[x] == i) {
            size_y = 0.0f;
            for (uint32_t y = 0; y < g->n; y++) {
                if (assign[y] == j) {
                    distance = distance + g->d[(uint64_t)x * g->n + y];
                    size_y = size_y + 1.0f;
                }
            }
            size_x = size_x + 1.0f;
        }
    }
    float denom = size_x * size_y;
    return distance / denom;
}

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

static uint32_t cl_clusters(const dendro *g, const uint64_t *assign, uint64_t *uniq)       /* :146-148 */
{
    memcpy(uniq, assign, sizeof(uint64_t) * g->n);
    qsort(uniq, g->n, sizeof(uint64_t), cmp_u64);
    uint32_t c = 0;
    for (uint32_t i = 0; i < g->n; i++) if (i == 0 || uniq[i] != uniq[i - 1]) uniq[c++] = uniq[i];
    return c;
}

int orc_clustering(const float *dist, uint32_t n, float perc, orc_cluster_op *ops,
                   uint32_t *n_ops, uint32_t *roots, uint32_t *n_roots, float *threshold_out)
{
    float threshold;
    if (orc_percentile(dist, (uint64_t)n * n, perc, &threshold) != 0) return -1;   /* :101 */
    if (threshold_out) *threshold_out = threshold;
    dendro g;
    g.cap = 2 * (uint64_t)n + 2; g.n_par = n; g.d = dist; g.n = n; g.n_clusters = n;
    g.parents = (uint64_t *)malloc(sizeof(uint64_t) * g.cap);
    for (uint32_t i = 0; i < n; i++) g.parents[i] = i;                /* :88-91 */
    uint64_t *assign = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint64_t *uniq = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint32_t cnt = 0;
    float distance = 0.0f;                                            /* :103 */
    while (g.n_clusters > 1 && distance < threshold) {                /* :104 */
        /* merge(), :175-209 */
        for (uint32_t i = 0; i < n; i++) assign[i] = cl_root(&g, i);  /* :176 */
        uint32_t nc = cl_clusters(&g, assign, uniq);                  /* :177, ascending instead of HashSet order */
        float min_linkage = ORC_INF;
        uint64_t p = 0, q = 0;                                        /* :179 */
        for (uint32_t a = 0; a < nc; a++)
            for (uint32_t b = 0; b < nc; b++)
                if (uniq[a] != uniq[b]) {
                    float l = cl_linkage(&g, assign, uniq[a], uniq[b]);
                    if (l < min_linkage) { min_linkage = l; p = uniq[a]; q = uniq[b]; }   /* :184-187 */
                }
        uint64_t k = g.n_par;                                         /* :134-141 */
        if (g.n_par + 1 > g.cap) { g.cap *= 2; g.parents = (uint64_t *)realloc(g.parents, sizeof(uint64_t) * g.cap); }
        g.parents[p] = k; g.parents[q] = k; g.parents[g.n_par++] = k;
        g.n_clusters -= 1;
        uint32_t op;                                                  /* :193-201 */
        if (p < n && q < n) op = ORC_S2S;
        else if (p >= n && q >= n) op = ORC_C2C;
        else if (p >= n && q < n) op = ORC_C2S;
        else op = ORC_S2C;
        ops[cnt].merge_i = (uint32_t)p; ops[cnt].merge_j = (uint32_t)q; ops[cnt].into = (uint32_t)k;
        ops[cnt].distance = min_linkage; ops[cnt].operation = op;
        cnt++;
        distance = min_linkage;                                       /* :106 */
    }
    for (uint32_t i = 0; i < n; i++) assign[i] = cl_root(&g, i);      /* :109 clusters() */
    uint32_t nc = n ? cl_clusters(&g, assign, uniq) : 0;
    for (uint32_t i = 0; i < nc; i++) roots[i] = (uint32_t)uniq[i];
    *n_roots = nc; *n_ops = cnt;
    free(assign); free(uniq); free(g.parents);
    return 0;
}

/* ---------------------------------------------------- clustering, fast restatement
 *
 * Same merge sequence and the same linkage BITS as orc_clustering, in O(n^3) worst case instead of O(n^4):
 * a linkage only changes when one of its two clusters is created, so linkage(p, q) (clustering.rs:153-170) is kept
 * in a matrix and only the row and the column of the freshly merged cluster are recomputed -- by the reference's own
 * loop order (x ascending over Cp, y ascending over Cq, ONE f32 accumulator, then / (size_x * size_y)), from sorted
 * member lists.  The arg-min of merge() (clustering.rs:178-190: ascending cluster id for i, then j, strict '<') is
 * the first minimum of the per-row first minima; a row's cached minimum is rescanned when its column merged away.
 * tests/test_oracle.py proves it equal to the literal orc_clustering on hundreds of random / tied / infinite
 * matrices; it exists so that the device UPGMA can be checked at sizes the O(n^4) loop cannot reach. */
typedef struct { uint32_t *mem; uint32_t cnt; } fc_list;

static float fc_linkage(const float *d, uint32_t n, const fc_list *cx, const fc_list *cy)   /* :153-170 */
{
    float distance = 0.0f;
    for (uint32_t a = 0; a < cx->cnt; a++) {
        const float *row = d + (uint64_t)cx->mem[a] * n;
        for (uint32_t b = 0; b < cy->cnt; b++) distance = distance + row[cy->mem[b]];
    }
    float denom = (float)cx->cnt * (float)cy->cnt;                    /* size_x * size_y, counted in f32 (exact < 2^24) */
    return distance / denom;
}

/* OpenMP threads of orc_clustering_fast.  The default (one per hardware thread of the MACHINE) oversubscribes a host whose
 * CPU share is smaller than the machine -- a GPU box grants 16 cores of a few hundred -- and two parallel regions per merge
 * then spend their time in barrier spins: a 1382-merge case that takes 1.4 s on 8 threads did not finish in 150 s.  The
 * binding sets this to the cores the process may actually use. */
static int orc_threads = 0;
void orc_set_threads(int threads) { orc_threads = threads > 0 ? threads : 0; }

int orc_clustering_fast(const float *dist, uint32_t n, float perc, orc_cluster_op *ops,
                        uint32_t *n_ops, uint32_t *roots, uint32_t *n_roots, float *threshold_out)
{
    if (orc_threads > 0) omp_set_num_threads(orc_threads);
    float threshold;
    if (orc_percentile(dist, (uint64_t)n * n, perc, &threshold) != 0) return -1;   /* :101 */
    if (threshold_out) *threshold_out = threshold;
    const uint64_t nn = (uint64_t)n * n;
    float *L = (float *)malloc(sizeof(float) * (nn ? nn : 1));        /* L[sp*n+sq]: linkage of the clusters in slots sp, sq */
    fc_list *lists = (fc_list *)calloc(n ? n : 1, sizeof(fc_list));   /* sorted members of the cluster in a slot */
    uint32_t *slot_id = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *active = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));   /* live slots in ascending cluster id */
    float *best_v = (float *)malloc(sizeof(float) * (n ? n : 1));
    uint32_t *best_c = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint8_t *stale = (uint8_t *)malloc(n ? n : 1);
    uint64_t *parents = (uint64_t *)malloc(sizeof(uint64_t) * (2 * (uint64_t)n + 2));
    for (uint32_t i = 0; i < n; i++) {
        lists[i].mem = (uint32_t *)malloc(sizeof(uint32_t)); lists[i].mem[0] = i; lists[i].cnt = 1;
        slot_id[i] = i; active[i] = i; stale[i] = 1; parents[i] = i;
    }
    for (uint64_t e = 0; e < nn; e++) { float s = 0.0f; s = s + dist[e]; L[e] = s / (1.0f * 1.0f); }
    uint32_t n_act = n, cnt = 0;
    uint64_t n_par = n;
    float distance = 0.0f;                                            /* :103 */
    while (n_act > 1 && distance < threshold) {                       /* :104 (n_clusters == n_act until a degenerate merge ends the loop) */
        #pragma omp parallel for schedule(dynamic, 16)
        for (uint32_t a = 0; a < n_act; a++) {
            const uint32_t sp = active[a];
            if (!stale[sp]) continue;
            float bv = ORC_INF; uint32_t bc = UINT32_MAX;
            for (uint32_t b = 0; b < n_act; b++) {
                const uint32_t sq = active[b];
                if (sq == sp) continue;                               /* :182 */
                const float l = L[(uint64_t)sp * n + sq];
                if (l < bv) { bv = l; bc = sq; }
            }
            best_v[sp] = bv; best_c[sp] = bc; stale[sp] = 0;
        }
        float min_linkage = ORC_INF;                                  /* :178 */
        uint32_t wp = UINT32_MAX, wq = UINT32_MAX;
        for (uint32_t a = 0; a < n_act; a++) {
            const uint32_t sp = active[a];
            if (best_c[sp] != UINT32_MAX && best_v[sp] < min_linkage) { min_linkage = best_v[sp]; wp = sp; wq = best_c[sp]; }
        }
        const uint64_t k = n_par;                                     /* :135 */
        if (wp == UINT32_MAX) {
            /* no linkage below +INF: min_merge stays (0, 0) (:179) and is merged: instance 0 is re-parented */
            parents[0] = k; parents[n_par++] = k;
            ops[cnt].merge_i = 0; ops[cnt].merge_j = 0; ops[cnt].into = (uint32_t)k;
            ops[cnt].distance = min_linkage; ops[cnt].operation = ORC_S2S;
            cnt++;
            distance = min_linkage;                                   /* INF < threshold is false: the loop ends */
            break;
        }
        const uint64_t p = slot_id[wp], q = slot_id[wq];
        parents[p] = k; parents[q] = k; parents[n_par++] = k;         /* :136-139 */
        uint32_t op;                                                  /* :193-201 */
        if (p < n && q < n) op = ORC_S2S;
        else if (p >= n && q >= n) op = ORC_C2C;
        else if (p >= n && q < n) op = ORC_C2S;
        else op = ORC_S2C;
        ops[cnt].merge_i = (uint32_t)p; ops[cnt].merge_j = (uint32_t)q; ops[cnt].into = (uint32_t)k;
        ops[cnt].distance = min_linkage; ops[cnt].operation = op;
        cnt++;
        distance = min_linkage;                                       /* :106 */
        /* the new cluster lives in slot wp: merged sorted member list */
        fc_list *lp = &lists[wp], *lq = &lists[wq];
        uint32_t *mm = (uint32_t *)malloc(sizeof(uint32_t) * (lp->cnt + lq->cnt));
        uint32_t ia = 0, ib = 0, io = 0;
        while (ia < lp->cnt && ib < lq->cnt) mm[io++] = (lp->mem[ia] < lq->mem[ib]) ? lp->mem[ia++] : lq->mem[ib++];
        while (ia < lp->cnt) mm[io++] = lp->mem[ia++];
        while (ib < lq->cnt) mm[io++] = lq->mem[ib++];
        free(lp->mem); free(lq->mem);
        lp->mem = mm; lp->cnt = io; lq->mem = NULL; lq->cnt = 0;
        slot_id[wp] = (uint32_t)k;
        uint32_t w = 0;                                               /* id order: drop p and q, append k (the largest id) */
        for (uint32_t a = 0; a < n_act; a++) if (active[a] != wp && active[a] != wq) active[w++] = active[a];
        active[w++] = wp;
        n_act = w;
        #pragma omp parallel for schedule(dynamic, 4)
        for (uint32_t a = 0; a < n_act - 1; a++) {
            const uint32_t s = active[a];
            L[(uint64_t)wp * n + s] = fc_linkage(dist, n, &lists[wp], &lists[s]);
            const float lrk = fc_linkage(dist, n, &lists[s], &lists[wp]);
            L[(uint64_t)s * n + wp] = lrk;
            if (best_c[s] == wp || best_c[s] == wq) stale[s] = 1;
            else if (lrk < best_v[s]) { best_v[s] = lrk; best_c[s] = wp; }   /* k is last in id order: strict '<' */
        }
        stale[wp] = 1;
    }
    /* clusters() (:109, :146-148): the distinct roots, ascending */
    uint64_t *rt = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) { uint64_t r = i; while (r != parents[r]) r = parents[r]; rt[i] = r; }
    qsort(rt, n, sizeof(uint64_t), cmp_u64);
    uint32_t nc = 0;
    for (uint32_t i = 0; i < n; i++) if (i == 0 || rt[i] != rt[i - 1]) roots[nc++] = (uint32_t)rt[i];
    *n_roots = nc; *n_ops = cnt;
    for (uint32_t i = 0; i < n; i++) free(lists[i].mem);
    free(rt); free(L); free(lists); free(slot_id); free(active); free(best_v); free(best_c); free(stale); free(parents);
    return 0;
}

/* clustering.rs:40-76 */
int orc_cluster_sets(const orc_cluster_op *ops, uint32_t n_ops, const uint32_t *roots,
                     uint32_t n_roots, uint32_t n, uint32_t *members, uint32_t *set_off,
                     uint32_t *n_sets)
{
    /* results: id -> list, ids < n + n_ops + 1 */
    uint64_t ids = (uint64_t)n + n_ops + 2;
    uint32_t **lists = (uint32_t **)calloc(ids, sizeof(uint32_t *));
    uint32_t *lens = (uint32_t *)calloc(ids, sizeof(uint32_t));
    for (uint32_t t = 0; t < n_ops; t++) {
        uint32_t i = ops[t].merge_i, j = ops[t].merge_j, k = ops[t].into;
        uint32_t li = (i < ids && lists[i]) ? lens[i] : 1, lj = (j < ids && lists[j]) ? lens[j] : 1;
        uint32_t *c = (uint32_t *)malloc(sizeof(uint32_t) * (li + lj));
        uint32_t pos = 0;
        if (i < ids && lists[i]) { memcpy(c, lists[i], sizeof(uint32_t) * li); pos = li; } else c[pos++] = i;
        if (j < ids && lists[j]) { memcpy(c + pos, lists[j], sizeof(uint32_t) * lj); pos += lj; } else c[pos++] = j;
        if (k < ids) { free(lists[k]); lists[k] = c; lens[k] = pos; } else free(c);
    }
    uint32_t ns = 0, mpos = 0;
    set_off[0] = 0;
    for (uint32_t r = 0; r < n_roots; r++) {
        uint32_t id = roots[r];
        if (id < ids && lists[id]) {
            for (uint32_t t = 0; t < lens[id]; t++)
                if (lists[id][t] < n) members[mpos++] = lists[id][t];     /* :65-69 */
            set_off[++ns] = mpos;
        }                                                                 /* else "Cluster not found", :71 */
    }
    *n_sets = ns;
    for (uint64_t t = 0; t < ids; t++) free(lists[t]);
    free(lists); free(lens);
    return 0;
}

/* ------------------------------------------------------------------ encoder */

/* numerics.rs:12-18 */
static float orc_mean(const float *x, uint64_t len)
{
    float mean = 0.0f;
    for (uint64_t i = 0; i < len; i++) mean = mean + x[i];
    return mean / (float)len;
}
/* numerics.rs:23-29 */
static float orc_std(const float *x, uint64_t len, float mu)
{
    float s = 0.0f;
    for (uint64_t i = 0; i < len; i++) { float d = x[i] - mu; float sq = d * d; s = s + sq; }
    return sqrtf(s / (float)len);
}

/* neural.rs:55-71 */
void orc_encode(const float *x, uint64_t t, uint32_t d_in, const float *w, const float *b,
                uint32_t latent, float *out)
{
    for (uint64_t f = 0; f < t; f++) {
        float *pred = out + f * (uint64_t)latent;
        const float *xf = x + f * (uint64_t)d_in;
        for (uint32_t j = 0; j < latent; j++) {
            float acc = 0.0f;                                         /* numerics.rs:310-316 */
            for (uint32_t k = 0; k < d_in; k++) { float p = xf[k] * w[(uint64_t)k * latent + j]; acc = acc + p; }
            acc = acc + b[j];                                         /* add_col, :247-258 */
            float e = expf(-acc);
            float s = 1.0f / (1.0f + e);                              /* sigmoid, :227-236 */
            pred[j] = s * 255.0f;                                     /* scale, :297-302 */
        }
        float mu = orc_mean(pred, latent);                            /* neural.rs:61 */
        float sigma = fmaxf(orc_std(pred, latent, mu), 1.0f);         /* :62 */
        for (uint32_t j = 0; j < latent; j++) pred[j] = (pred[j] - mu) / sigma;   /* :66, numerics.rs:71-73 */
    }
}

/* ----------------------------------------------------------------- cepstrum */

uint64_t orc_cepstrum(const int16_t *samples, uint64_t n_samples, uint32_t fft_size,
                      uint32_t fft_step, uint32_t filter_size, float *out, uint32_t *n_bins)
{
    if (fft_size < 2 || fft_step == 0 || filter_size == 0) return 0;
    uint32_t L = fft_size / filter_size;                              /* spectrogram.rs:38 */
    uint32_t half = fft_size / 2;                                     /* :64 */
    uint32_t step = L / 2;                                            /* :67 */
    if (L == 0 || step == 0) return 0;
    /* convolve output count: i in (L..half).step_by(step)  (numerics.rs:105) */
    uint32_t K = 0;
    for (uint32_t i = L; i < half; i += step) K++;
    if (K < 5) return 0;
    if (n_bins) *n_bins = K - 4;                                      /* :76 */
    uint64_t T = 0;
    for (uint64_t i = fft_size; i < n_samples; i += fft_step) T++;   /* :51 */
    if (!out) return T;

    float *hamming = (float *)malloc(sizeof(float) * fft_size);
    for (uint32_t i = 0; i < fft_size; i++) {                         /* numerics.rs:60-66 */
        float arg = (2.0f * 3.14159265358979323846f * (float)i) / (float)fft_size;
        hamming[i] = 0.54f + 0.46f * cosf(arg);
    }
    float *triag = (float *)calloc(L, sizeof(float));                 /* numerics.rs:78-86 */
    uint32_t center = (L - 1) / 2;
    for (uint32_t i = 0; i <= center; i++) { triag[i] = (float)i / (float)L; triag[L - 1 - i] = (float)i / (float)L; }

    float *win = (float *)malloc(sizeof(float) * fft_size);
    float *mag = (float *)malloc(sizeof(float) * half);
    float *conv = (float *)malloc(sizeof(float) * K);
    float *ceps = (float *)malloc(sizeof(float) * K);
    double *ctab = (double *)malloc(sizeof(double) * fft_size);
    double *stab = (double *)malloc(sizeof(double) * fft_size);
    for (uint32_t t = 0; t < fft_size; t++) {
        ctab[t] = cos(2.0 * M_PI * (double)t / (double)fft_size);
        stab[t] = sin(2.0 * M_PI * (double)t / (double)fft_size);
    }
    uint64_t f = 0;
    for (uint64_t i = fft_size; i < n_samples; i += fft_step, f++) {
        uint64_t start = i - fft_size;                                /* :52 */
        for (uint32_t s = 0; s < fft_size; s++) win[s] = (float)samples[start + s] * hamming[s];   /* :55-59 */
        for (uint32_t k = 0; k < half; k++) {                         /* forward DFT, e^{-2 pi i k s / N} */
            double re = 0.0, im = 0.0;
            for (uint32_t s = 0; s < fft_size; s++) {
                uint32_t idx = (uint32_t)(((uint64_t)k * s) % fft_size);
                re += (double)win[s] * ctab[idx];
                im -= (double)win[s] * stab[idx];
            }
            float fr = (float)re, fi = (float)im;
            float nsq = fr * fr + fi * fi;                            /* norm_sqr, :63 */
            mag[k] = sqrtf(nsq);
        }
        uint32_t c = 0;
        for (uint32_t p = L; p < half; p += step, c++) {              /* numerics.rs:102-109 */
            float dot = 0.0f;
            for (uint32_t q = 0; q < L; q++) { float pr = triag[q] * mag[p - L + q]; dot = dot + pr; }
            conv[c] = logf(dot + 1e-6f);                              /* :69 */
        }
        for (uint32_t k = 0; k < K; k++) {                            /* DCT-I, rustdct definition */
            double acc = 0.5 * (double)conv[0] + ((k & 1) ? -0.5 : 0.5) * (double)conv[K - 1];
            for (uint32_t q = 1; q + 1 < K; q++) acc += (double)conv[q] * cos(M_PI * (double)q * (double)k / (double)(K - 1));
            ceps[k] = (float)acc;
        }
        float mu = orc_mean(ceps + 4, K - 4);                         /* :74 */
        for (uint32_t k = 4; k < K; k++) out[f * (uint64_t)(K - 4) + (k - 4)] = ceps[k] - mu;   /* :75 */
    }
    free(hamming); free(triag); free(win); free(mag); free(conv); free(ceps); free(ctab); free(stab);
    return T;
}

/* ---------------------------------------------------------------------- VAT */

/* spectrogram.rs:174-187 */
void orc_variance(const float *frames, uint64_t t, uint32_t n_bins, uint32_t k, float *out)
{
    float *deltas = (float *)malloc(sizeof(float) * (t ? t : 1));
    for (uint64_t i = 0; i < t; i++) {
        const float *v = frames + i * (uint64_t)n_bins;
        float m = orc_mean(v, n_bins);
        deltas[i] = orc_std(v, n_bins, m);
    }
    for (uint64_t i = 0; i < t; i++) out[i] = (i >= k) ? orc_mean(deltas + (i - k), k) : 0.0f;
    free(deltas);
}

/* spectrogram.rs:192-216 */
int orc_interesting_ranges(const float *frames, uint64_t t, uint32_t n_bins, uint32_t moving_average, float perc,
                           uint64_t min_len, uint64_t *ranges, uint64_t capacity, uint64_t *n_ranges)
{
    float *var = (float *)malloc(sizeof(float) * (t ? t : 1));
    orc_variance(frames, t, n_bins, moving_average, var);
    float th;
    if (orc_percentile(var, t, perc, &th) != 0) { free(var); return -1; }     /* :198 */
    uint64_t start = 0, cnt = 0;
    int recording = 1;                                                           /* :202 */
    for (uint64_t i = 0; i < t; i++) {
        if (var[i] >= th && !recording) { start = i; recording = 1; }           /* :204-207 */
        if (var[i] < th && recording) {                                          /* :208-213 */
            recording = 0;
            if (i - start > min_len) { if (cnt < capacity) { ranges[2 * cnt] = start; ranges[2 * cnt + 1] = i; } cnt++; }
        }
    }
    *n_ranges = cnt;
    free(var);
    return 0;
}
