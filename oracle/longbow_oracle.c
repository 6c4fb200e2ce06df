/*
 * longbow_oracle.c -- CPU restatement of the Longbow k-NN hot path.
 * TEST INFRASTRUCTURE ONLY (see longbow_oracle.h).  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off: every f32 operation rounds once, as Go on amd64).
 */
#include "longbow_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ======================================================================
 * Per-pair metrics
 * ====================================================================== */

/* SEQ: referenceEuclidean (internal/simd/simd_test.go:13-20), before sqrt.
 * UNROLL4: L2SquaredFloat32 (internal/simd/distance_functions.go:195-227) ==
 *          the sum inside euclideanUnrolled4x (internal/simd/simd.go:365-396). */
float lbo_l2sq(const float *a, const float *b, int n, int order)
{
    if (n <= 0) return 0.0f;
    if (order == LBO_ORDER_SEQ) {
        float sum = 0.0f;
        for (int i = 0; i < n; i++) {
            float d = a[i] - b[i];
            sum += d * d;
        }
        return sum;
    }
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int i = 0;
    for (; i <= n - 4; i += 4) {
        float d0 = a[i] - b[i];
        float d1 = a[i + 1] - b[i + 1];
        float d2 = a[i + 2] - b[i + 2];
        float d3 = a[i + 3] - b[i + 3];
        s0 += d0 * d0;
        s1 += d1 * d1;
        s2 += d2 * d2;
        s3 += d3 * d3;
    }
    for (; i < n; i++) {
        float d = a[i] - b[i];
        s0 += d * d;
    }
    /* Go evaluates s0 + s1 + s2 + s3 left to right */
    float t = s0 + s1;
    t = t + s2;
    t = t + s3;
    return t;
}

/* float32(math.Sqrt(float64(sum)))  (simd.go:131-134, :395; simd_test.go:19) */
float lbo_euclidean(const float *a, const float *b, int n, int order)
{
    if (n <= 0) return 0.0f; /* distance_functions.go:21-23 */
    return (float)sqrt((double)lbo_l2sq(a, b, n, order));
}

/* cosineGeneric (simd.go:138-152) / cosineUnrolled4x (simd.go:399-450);
 * len 0 -> 1.0 (distance_functions.go:51-53). */
float lbo_cosine(const float *a, const float *b, int n, int order)
{
    if (n <= 0) return 1.0f;
    float dot, na, nb;
    if (order == LBO_ORDER_SEQ) {
        dot = 0.0f; na = 0.0f; nb = 0.0f;
        for (int i = 0; i < n; i++) {
            dot += a[i] * b[i];
            na += a[i] * a[i];
            nb += b[i] * b[i];
        }
    } else {
        float d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        float b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        int i = 0;
        for (; i <= n - 4; i += 4) {
            float x0 = a[i], x1 = a[i + 1], x2 = a[i + 2], x3 = a[i + 3];
            float y0 = b[i], y1 = b[i + 1], y2 = b[i + 2], y3 = b[i + 3];
            d0 += x0 * y0; d1 += x1 * y1; d2 += x2 * y2; d3 += x3 * y3;
            a0 += x0 * x0; a1 += x1 * x1; a2 += x2 * x2; a3 += x3 * x3;
            b0 += y0 * y0; b1 += y1 * y1; b2 += y2 * y2; b3 += y3 * y3;
        }
        for (; i < n; i++) {
            d0 += a[i] * b[i];
            a0 += a[i] * a[i];
            b0 += b[i] * b[i];
        }
        dot = d0 + d1; dot = dot + d2; dot = dot + d3;
        na = a0 + a1; na = na + a2; na = na + a3;
        nb = b0 + b1; nb = nb + b2; nb = nb + b3;
    }
    if (na == 0.0f || nb == 0.0f) return 1.0f;
    float denom = (float)sqrt((double)na * (double)nb);
    float ratio = dot / denom;
    return 1.0f - ratio;
}

/* dotGeneric (simd.go:154-163) / dotUnrolled4x (simd.go:453-479) */
float lbo_dot(const float *a, const float *b, int n, int order)
{
    if (n <= 0) return 0.0f;
    if (order == LBO_ORDER_SEQ) {
        float sum = 0.0f;
        for (int i = 0; i < n; i++) sum += a[i] * b[i];
        return sum;
    }
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = 0;
    for (; i <= n - 4; i += 4) {
        s0 += a[i] * b[i];
        s1 += a[i + 1] * b[i + 1];
        s2 += a[i + 2] * b[i + 2];
        s3 += a[i + 3] * b[i + 3];
    }
    for (; i < n; i++) s0 += a[i] * b[i];
    float t = s0 + s1;
    t = t + s2;
    t = t + s3;
    return t;
}

float lbo_distance(int metric, const float *q, const float *x, int n, int order)
{
    switch (metric) {
    case LBO_METRIC_EUCLIDEAN: return lbo_euclidean(q, x, n, order);
    case LBO_METRIC_COSINE:    return lbo_cosine(q, x, n, order);
    default:                   return -lbo_dot(q, x, n, order);
    }
}

void lbo_batch_flat(int metric, int order, const float *q, const float *flat,
                    int64_t n, int dims, float *out)
{
    for (int64_t i = 0; i < n; i++)
        out[i] = lbo_distance(metric, q, flat + i * (int64_t)dims, dims, order);
}

/* ======================================================================
 * Go container/heap, restated (src/container/heap/heap.go)
 * ====================================================================== */
typedef struct { int64_t id; float score; int src; } hitem;
typedef int (*less_fn)(const hitem *h, int i, int j);

static void h_swap(hitem *h, int i, int j) { hitem t = h[i]; h[i] = h[j]; h[j] = t; }

static void h_up(hitem *h, int j, less_fn less)
{
    for (;;) {
        int i = (j - 1) / 2; /* parent; Go truncates toward zero so j==0 -> i==0 */
        if (i == j || !less(h, j, i)) break;
        h_swap(h, i, j);
        j = i;
    }
}

static void h_down(hitem *h, int i0, int n, less_fn less)
{
    int i = i0;
    for (;;) {
        int j1 = 2 * i + 1;
        if (j1 >= n || j1 < 0) break;
        int j = j1;
        int j2 = j1 + 1;
        if (j2 < n && less(h, j2, j1)) j = j2;
        if (!less(h, j, i)) break;
        h_swap(h, i, j);
        i = j;
    }
}

static void h_push(hitem *h, int *len, hitem x, less_fn less)
{
    h[*len] = x;
    (*len)++;
    h_up(h, *len - 1, less);
}

static hitem h_pop(hitem *h, int *len, less_fn less)
{
    int n = *len - 1;
    h_swap(h, 0, n);
    h_down(h, 0, n, less);
    (*len)--;
    return h[n];
}

/* bfSearchHeap.Less: max-heap (adaptive_index.go:334) */
static int less_max(const hitem *h, int i, int j) { return h[i].score > h[j].score; }
/* ResultHeap.Less: min-heap (result_merger.go:16) */
static int less_min(const hitem *h, int i, int j) { return h[i].score < h[j].score; }

int lbo_bruteforce_goheap(int metric, int order, const float *q, const float *flat,
                          int64_t n, int dims, int k, int64_t *out_ids, float *out_dist)
{
    if (n <= 0 || k <= 0) return 0; /* adaptive_index.go:172-174 */
    hitem *h = (hitem *)malloc(sizeof(hitem) * (size_t)(k + 1));
    int len = 0;
    for (int64_t i = 0; i < n; i++) {
        float dist = lbo_distance(metric, q, flat + i * (int64_t)dims, dims, order);
        if (len < k) {
            hitem it = { i, dist, 0 };
            h_push(h, &len, it, less_max);
        } else if (dist < h[0].score) {
            (void)h_pop(h, &len, less_max);
            hitem it = { i, dist, 0 };
            h_push(h, &len, it, less_max);
        }
    }
    int cnt = len;
    for (int i = cnt - 1; i >= 0; i--) { /* adaptive_index.go:215-222 */
        hitem it = h_pop(h, &len, less_max);
        out_ids[i] = it.id;
        out_dist[i] = it.score;
    }
    free(h);
    return cnt;
}

/* ======================================================================
 * Canonical top-k: ascending (distance, index)
 * ====================================================================== */
typedef struct { float d; int64_t i; } cand;

/* canonical order: ascending distance, every NaN after +inf, ties by row index (the reference's heap
 * leaves NaN undefined: a NaN never satisfies `dist < root`, adaptive_index.go:206) */
static int cand_less(const cand *a, const cand *b)
{
    const int an = a->d != a->d, bn = b->d != b->d;
    if (an || bn) {
        if (an != bn) return bn;
        return a->i < b->i;
    }
    if (a->d < b->d) return 1;
    if (a->d > b->d) return 0;
    return a->i < b->i;
}

static int cand_cmp(const void *pa, const void *pb)
{
    const cand *a = (const cand *)pa, *b = (const cand *)pb;
    if (cand_less(a, b)) return -1;
    if (cand_less(b, a)) return 1;
    return 0;
}

/* bounded max-heap keyed by (d, i): keeps the k smallest pairs */
static void ck_sift_down(cand *h, int n, int i)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && cand_less(&h[m], &h[l])) m = l;
        if (r < n && cand_less(&h[m], &h[r])) m = r;
        if (m == i) break;
        cand t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

static void ck_sift_up(cand *h, int i)
{
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!cand_less(&h[p], &h[i])) break;
        cand t = h[i]; h[i] = h[p]; h[p] = t;
        i = p;
    }
}

typedef struct { cand *h; int len; int k; } topk_acc;

static void topk_offer(topk_acc *t, float d, int64_t idx)
{
    cand c = { d, idx };
    if (t->len < t->k) {
        t->h[t->len] = c;
        ck_sift_up(t->h, t->len);
        t->len++;
    } else if (cand_less(&c, &t->h[0])) {
        t->h[0] = c;
        ck_sift_down(t->h, t->len, 0);
    }
}

static int topk_finish(topk_acc *t, int k, int64_t *out_ids, float *out_dist)
{
    qsort(t->h, (size_t)t->len, sizeof(cand), cand_cmp);
    for (int i = 0; i < t->len; i++) { out_ids[i] = t->h[i].i; out_dist[i] = t->h[i].d; }
    for (int i = t->len; i < k; i++) { out_ids[i] = -1; out_dist[i] = FLT_MAX; }
    return t->len;
}

int lbo_topk_canonical(const float *dist, int64_t n, int k, int64_t *out_ids, float *out_dist)
{
    if (k <= 0) return 0;
    topk_acc t = { (cand *)malloc(sizeof(cand) * (size_t)k), 0, k };
    for (int64_t i = 0; i < n; i++) topk_offer(&t, dist[i], i);
    int cnt = topk_finish(&t, k, out_ids, out_dist);
    free(t.h);
    return cnt;
}

typedef struct {
    int metric, order, dims, k, q0, q1;
    const float *queries, *flat;
    int64_t n;
    const uint8_t *mask;
    const int64_t *ids;
    int64_t *out_ids;
    float *out_dist;
} search_job;

static void *search_worker(void *arg)
{
    search_job *j = (search_job *)arg;
    topk_acc t = { (cand *)malloc(sizeof(cand) * (size_t)(j->k > 0 ? j->k : 1)), 0, j->k };
    for (int qi = j->q0; qi < j->q1; qi++) {
        const float *q = j->queries + (int64_t)qi * j->dims;
        t.len = 0;
        for (int64_t i = 0; i < j->n; i++) {
            if (j->mask && !j->mask[i]) continue;
            float d = lbo_distance(j->metric, q, j->flat + i * (int64_t)j->dims, j->dims, j->order);
            topk_offer(&t, d, i);
        }
        int64_t *oi = j->out_ids + (int64_t)qi * j->k;
        float *od = j->out_dist + (int64_t)qi * j->k;
        int cnt = topk_finish(&t, j->k, oi, od);
        if (j->ids)
            for (int r = 0; r < cnt; r++) oi[r] = j->ids[oi[r]];
    }
    free(t.h);
    return NULL;
}

void lbo_search_batch(int metric, int order, const float *queries, int nq,
                      const float *flat, int64_t n, int dims, int k,
                      const uint8_t *mask, const int64_t *ids,
                      int64_t *out_ids, float *out_dist, int nthreads)
{
    if (nq <= 0 || k <= 0) return;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nq) nthreads = nq;
    search_job *jobs = (search_job *)calloc((size_t)nthreads, sizeof(search_job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; t++) {
        search_job j = { metric, order, dims, k,
                         (int)((int64_t)nq * t / nthreads), (int)((int64_t)nq * (t + 1) / nthreads),
                         queries, flat, n, mask, ids, out_ids, out_dist };
        jobs[t] = j;
        if (nthreads == 1) search_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, search_worker, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
}

/* ======================================================================
 * Product quantisation
 * ====================================================================== */
void lbo_build_adc_table(const float *codebooks, int M, int K, int sub,
                         const float *query, float *table)
{
    for (int i = 0; i < M; i++) {
        const float *qs = query + (size_t)i * sub;
        const float *cb = codebooks + (size_t)i * K * sub;
        for (int j = 0; j < K; j++) /* simd.L2Squared -> L2SquaredFloat32: UNROLL4 */
            table[(size_t)i * K + j] = lbo_l2sq(qs, cb + (size_t)j * sub, sub, LBO_ORDER_UNROLL4);
    }
}

void lbo_adc_batch(const float *table, const uint8_t *codes, int m, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; i++) {
        float sum = 0.0f;
        const uint8_t *c = codes + i * (int64_t)m;
        for (int j = 0; j < m; j++) sum += table[j * 256 + (int)c[j]];
        out[i] = (float)sqrt((double)sum);
    }
}

float lbo_adc_single(const float *table, const uint8_t *code, int M, int K)
{
    float sum = 0.0f;
    for (int m = 0; m < M; m++) sum += table[m * K + (int)code[m]];
    return sum;
}

void lbo_pq_encode(const float *codebooks, int M, int K, int sub,
                   const float *vec, uint8_t *codes)
{
    for (int m = 0; m < M; m++) {
        const float *sv = vec + (size_t)m * sub;
        const float *cb = codebooks + (size_t)m * K * sub;
        int best = 0;
        if (K <= 16) {
            /* encodeSequential (encoder.go:92-119): L2Squared, bestDist = MaxFloat32, strict < */
            float bd = FLT_MAX;
            for (int k = 0; k < K; k++) {
                float d = lbo_l2sq(sv, cb + (size_t)k * sub, sub, LBO_ORDER_UNROLL4);
                if (d < bd) { bd = d; best = k; }
            }
        } else {
            /* FindNearestCentroid, K > 8 branch (simd.go:305-326): batch-flat Euclidean
             * (sqrt'd, UNROLL4 order) then first strict minimum. */
            float bd = lbo_euclidean(sv, cb, sub, LBO_ORDER_UNROLL4);
            for (int k = 1; k < K; k++) {
                float d = lbo_euclidean(sv, cb + (size_t)k * sub, sub, LBO_ORDER_UNROLL4);
                if (d < bd) { bd = d; best = k; }
            }
        }
        codes[m] = (uint8_t)best;
    }
}

void lbo_pq_decode(const float *codebooks, int M, int K, int sub,
                   const uint8_t *codes, float *vec)
{
    for (int m = 0; m < M; m++)
        memcpy(vec + (size_t)m * sub,
               codebooks + ((size_t)m * K + codes[m]) * sub, sizeof(float) * (size_t)sub);
}

static uint32_t rd_u32le(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

int lbo_pq_parse_blob(const uint8_t *blob, size_t len, int *dims, int *M, int *K)
{
    if (len < 12) return -1;                       /* "invalid PQ data: too short" */
    uint32_t d = rd_u32le(blob), m = rd_u32le(blob + 4), k = rd_u32le(blob + 8);
    if (m == 0 || d % m != 0) return -2;           /* "invalid PQ parameters" */
    size_t sub = d / m;
    size_t expect = 12 + (size_t)m * k * sub * 4;
    if (len != expect) return -3;                  /* "size mismatch" */
    *dims = (int)d; *M = (int)m; *K = (int)k;
    return 0;
}

/* ======================================================================
 * RingSharder + MergeSortedStreams
 * ====================================================================== */
uint32_t lbo_fnv1a32(const uint8_t *p, size_t n)
{
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 16777619u; }
    return h;
}

static int u32_cmp(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

int lbo_ring_build(int num_shards, int vnodes, uint32_t *hashes, int *owners)
{
    if (vnodes <= 0) vnodes = 20; /* sharding_strategy.go:51-53 */
    int np = 0;
    /* insertion order = shard ascending, vnode ascending; a later identical hash
     * overwrites ring[h] (Go map assignment) but is appended again to sortedHashes */
    uint32_t *ins_h = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)num_shards * vnodes);
    int *ins_o = (int *)malloc(sizeof(int) * (size_t)num_shards * vnodes);
    for (int s = 0; s < num_shards; s++) {
        for (int v = 0; v < vnodes; v++) {
            char key[48];
            /* strconv.Itoa(shard) ":" strconv.Itoa(vnode)  (sharding_strategy.go:77-83) */
            int len = snprintf(key, sizeof key, "%d:%d", s, v);
            ins_h[np] = lbo_fnv1a32((const uint8_t *)key, (size_t)len);
            ins_o[np] = s;
            np++;
        }
    }
    memcpy(hashes, ins_h, sizeof(uint32_t) * (size_t)np);
    qsort(hashes, (size_t)np, sizeof(uint32_t), u32_cmp);
    for (int i = 0; i < np; i++) {
        int owner = -1;
        for (int j = 0; j < np; j++)
            if (ins_h[j] == hashes[i]) owner = ins_o[j]; /* last writer wins */
        owners[i] = owner;
    }
    free(ins_h);
    free(ins_o);
    return np;
}

int lbo_ring_get_shard(const uint32_t *hashes, const int *owners, int npoints, uint64_t id)
{
    if (npoints == 0) return 0;
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(id >> (8 * i));
    uint32_t h = lbo_fnv1a32(b, 8);
    int lo = 0, hi = npoints; /* sort.Search: first i with hashes[i] >= h */
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (!(hashes[mid] >= h)) lo = mid + 1; else hi = mid;
    }
    if (lo == npoints) lo = 0;
    return owners[lo];
}

int lbo_merge_sorted_streams(const int64_t *ids, const float *scores, const int *lens,
                             int nlists, int k, int64_t *out_ids, float *out_scores)
{
    hitem *h = (hitem *)malloc(sizeof(hitem) * (size_t)(nlists + 1));
    int *pos = (int *)calloc((size_t)nlists, sizeof(int));
    int *base = (int *)calloc((size_t)nlists, sizeof(int));
    int len = 0, off = 0;
    for (int s = 0; s < nlists; s++) {
        base[s] = off;
        off += lens[s];
        if (lens[s] > 0) {
            hitem it = { ids[base[s]], scores[base[s]], s };
            h_push(h, &len, it, less_min);
        }
    }
    int count = 0;
    while (len > 0 && (k <= 0 || count < k)) {
        hitem it = h_pop(h, &len, less_min);
        out_ids[count] = it.id;
        out_scores[count] = it.score;
        count++;
        int s = it.src;
        pos[s]++;
        if (pos[s] < lens[s]) {
            hitem nx = { ids[base[s] + pos[s]], scores[base[s] + pos[s]], s };
            h_push(h, &len, nx, less_min);
        }
    }
    free(h); free(pos); free(base);
    return count;
}

/* ======================================================================
 * Predicate masks
 * ====================================================================== */
#define LBO_MATCH_BODY                                   \
    for (int64_t i = 0; i < n; i++) {                    \
        int r;                                           \
        switch (op) {                                    \
        case 0: r = src[i] == val; break;                \
        case 1: r = src[i] != val; break;                \
        case 2: r = src[i] > val; break;                 \
        case 3: r = src[i] >= val; break;                \
        case 4: r = src[i] < val; break;                 \
        default: r = src[i] <= val; break;               \
        }                                                \
        dst[i] = (uint8_t)r;                             \
    }

void lbo_match_int64(const int64_t *src, int64_t n, int64_t val, int op, uint8_t *dst) { LBO_MATCH_BODY }
void lbo_match_float32(const float *src, int64_t n, float val, int op, uint8_t *dst) { LBO_MATCH_BODY }
void lbo_and_bytes(uint8_t *dst, const uint8_t *src, int64_t n)
{
    for (int64_t i = 0; i < n; i++) dst[i] &= src[i];
}

/* ======================================================================
 * Reciprocal-rank fusion (internal/store/rrf.go:10-51)
 * ====================================================================== */
typedef struct { int64_t id; double s; float f; } rrf_item;
static int rrf_cmp(const void *pa, const void *pb)
{
    const rrf_item *a = (const rrf_item *)pa, *b = (const rrf_item *)pb;
    if (a->f > b->f) return -1;
    if (a->f < b->f) return 1;
    return (a->id > b->id) - (a->id < b->id);
}

int lbo_rrf(const int64_t *dense, int nd, const int64_t *sparse, int ns, int k, int limit,
            int64_t *out_ids, float *out_scores)
{
    if (k <= 0) k = 60;
    rrf_item *it = (rrf_item *)malloc(sizeof(rrf_item) * (size_t)(nd + ns + 1));
    int n = 0;
    for (int pass = 0; pass < 2; pass++) {
        const int64_t *l = pass == 0 ? dense : sparse;
        const int len = pass == 0 ? nd : ns;
        for (int rank = 0; rank < len; rank++) {
            if (l[rank] < 0) continue;
            int j = 0;
            for (; j < n; j++)
                if (it[j].id == l[rank]) break;
            if (j == n) { it[n].id = l[rank]; it[n].s = 0.0; n++; }
            it[j].s += 1.0 / (double)(k + rank + 1);
        }
    }
    for (int j = 0; j < n; j++) it[j].f = (float)it[j].s;
    qsort(it, (size_t)n, sizeof(rrf_item), rrf_cmp);
    int cnt = (limit > 0 && n > limit) ? limit : n;
    for (int j = 0; j < cnt; j++) { out_ids[j] = it[j].id; out_scores[j] = it[j].f; }
    free(it);
    return cnt;
}

/* ======================================================================
 * Synthetic data (shared definition with lb_gpu_fill_uniform)
 * ====================================================================== */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed + (idx + 1) * 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

void lbo_fill_uniform(float *dst, int64_t n, uint64_t seed, int64_t offset)
{
    for (int64_t i = 0; i < n; i++)
        dst[i] = (float)(splitmix64_at(seed, (uint64_t)(offset + i)) >> 40) * (1.0f / 16777216.0f);
}

void lbo_fill_codes(uint8_t *dst, int64_t n, uint64_t seed, int64_t offset)
{
    for (int64_t i = 0; i < n; i++)
        dst[i] = (uint8_t)(splitmix64_at(seed, (uint64_t)(offset + i)) >> 56);
}

/* internal/store/adaptive_search.go:7-39 */
int lbo_adaptive_limit(int k, uint64_t matches, int total)
{
    if (total == 0 || matches == 0) return k;
    const double selectivity = (double)matches / (double)total;
    double factor = 1.0 / selectivity;
    if (factor > 50.0) factor = 50.0;
    if (factor < 2.0) factor = 2.0;
    int limit = (int)((double)k * factor);
    if (limit > total) limit = total;
    if (limit < k) limit = k;
    return limit;
}
