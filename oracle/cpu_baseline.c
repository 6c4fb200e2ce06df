/*
 * cpu_baseline.c -- the reference's CPU brute-force idiom, timed beside the GPU.
 * TEST/BENCH INFRASTRUCTURE ONLY (bench.py "cpu_baseline" leg).  kind = "port":
 * the Go reference cannot be built in this image (no Go toolchain), so this is a
 * C port of its structure, not the reference binary.
 *
 * Driver idiom: internal/store/recall_validation_test.go:237-300 -- queries are
 * partitioned over worker threads; each worker streams the WHOLE corpus per
 * query (flat-batch distance), then selects top-k (bounded max-heap, strict <,
 * internal/store/adaptive_index.go:176-222).
 *
 * simd=1 kernels follow the shape of the reference's AVX2/AVX-512 wrappers
 * (internal/simd/simd_amd64.go:21-63,125-140; distance_amd64.s:36-71): blocks of
 * 4 accumulators x 8/16 lanes with FMA, horizontal reduce at the end.  They are
 * written as fixed-width lane loops that gcc vectorises; target_clones picks
 * AVX-512 / AVX2 / baseline at load time on whatever host runs the bench.
 */
#include "longbow_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define LANES 16
#define NACC 4

#define CLONES __attribute__((target_clones("avx512f", "avx2,fma", "default")))

/* dot, ||q-x||^2 or (dot, nb) in blocked-lane order */
CLONES static float simd_l2sq(const float *a, const float *b, int n)
{
    float acc[NACC][LANES];
    memset(acc, 0, sizeof acc);
    int i = 0;
    for (; i + NACC * LANES <= n; i += NACC * LANES)
        for (int u = 0; u < NACC; u++)
            for (int l = 0; l < LANES; l++) {
                float d = a[i + u * LANES + l] - b[i + u * LANES + l];
                acc[u][l] += d * d;
            }
    float s = 0.0f;
    for (int u = 0; u < NACC; u++)
        for (int l = 0; l < LANES; l++) s += acc[u][l];
    for (; i < n; i++) { float d = a[i] - b[i]; s += d * d; }
    return s;
}

CLONES static void simd_dot_nb(const float *a, const float *b, int n, float *dot, float *nb)
{
    float ad[NACC][LANES], an[NACC][LANES];
    memset(ad, 0, sizeof ad);
    memset(an, 0, sizeof an);
    int i = 0;
    for (; i + NACC * LANES <= n; i += NACC * LANES)
        for (int u = 0; u < NACC; u++)
            for (int l = 0; l < LANES; l++) {
                float x = a[i + u * LANES + l], y = b[i + u * LANES + l];
                ad[u][l] += x * y;
                an[u][l] += y * y;
            }
    float s = 0.0f, t = 0.0f;
    for (int u = 0; u < NACC; u++)
        for (int l = 0; l < LANES; l++) { s += ad[u][l]; t += an[u][l]; }
    for (; i < n; i++) { s += a[i] * b[i]; t += b[i] * b[i]; }
    *dot = s; *nb = t;
}

CLONES static float simd_dot(const float *a, const float *b, int n)
{
    float ad[NACC][LANES];
    memset(ad, 0, sizeof ad);
    int i = 0;
    for (; i + NACC * LANES <= n; i += NACC * LANES)
        for (int u = 0; u < NACC; u++)
            for (int l = 0; l < LANES; l++) ad[u][l] += a[i + u * LANES + l] * b[i + u * LANES + l];
    float s = 0.0f;
    for (int u = 0; u < NACC; u++)
        for (int l = 0; l < LANES; l++) s += ad[u][l];
    for (; i < n; i++) s += a[i] * b[i];
    return s;
}

typedef struct { float d; int64_t i; } cand;

static inline int cand_less(cand a, cand b) { return a.d < b.d || (a.d == b.d && a.i < b.i); }

static void sift_down(cand *h, int n, int i)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && cand_less(h[m], h[l])) m = l;
        if (r < n && cand_less(h[m], h[r])) m = r;
        if (m == i) break;
        cand t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

static void sift_up(cand *h, int i)
{
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!cand_less(h[p], h[i])) break;
        cand t = h[i]; h[i] = h[p]; h[p] = t;
        i = p;
    }
}

static int cand_cmp(const void *pa, const void *pb)
{
    cand a = *(const cand *)pa, b = *(const cand *)pb;
    return cand_less(a, b) ? -1 : (cand_less(b, a) ? 1 : 0);
}

typedef struct {
    int metric, dims, k, q0, q1, simd;
    const float *queries, *flat;
    int64_t n;
    int64_t *out_ids;
    float *out_dist;
} job;

static void *worker(void *arg)
{
    job *j = (job *)arg;
    cand *h = (cand *)malloc(sizeof(cand) * (size_t)j->k);
    const int D = j->dims;
    for (int qi = j->q0; qi < j->q1; qi++) {
        const float *q = j->queries + (int64_t)qi * D;
        float nq = 0.0f;
        if (j->metric == LBO_METRIC_COSINE) {
            if (j->simd) nq = simd_dot(q, q, D);
            else for (int i = 0; i < D; i++) nq += q[i] * q[i];
        }
        int len = 0;
        for (int64_t r = 0; r < j->n; r++) {
            const float *x = j->flat + r * (int64_t)D;
            float d;
            if (!j->simd) {
                d = lbo_distance(j->metric, q, x, D, LBO_ORDER_SEQ);
            } else if (j->metric == LBO_METRIC_EUCLIDEAN) {
                d = (float)sqrt((double)simd_l2sq(q, x, D));
            } else if (j->metric == LBO_METRIC_COSINE) {
                float dot, nb;
                simd_dot_nb(q, x, D, &dot, &nb);
                d = (nq == 0.0f || nb == 0.0f) ? 1.0f
                    : 1.0f - dot / (float)sqrt((double)nq * (double)nb);
            } else {
                d = -simd_dot(q, x, D);
            }
            cand c = { d, r };
            if (len < j->k) { h[len] = c; sift_up(h, len); len++; }
            else if (cand_less(c, h[0])) { h[0] = c; sift_down(h, len, 0); }
        }
        qsort(h, (size_t)len, sizeof(cand), cand_cmp);
        for (int t = 0; t < len; t++) {
            j->out_ids[(int64_t)qi * j->k + t] = h[t].i;
            j->out_dist[(int64_t)qi * j->k + t] = h[t].d;
        }
        for (int t = len; t < j->k; t++) {
            j->out_ids[(int64_t)qi * j->k + t] = -1;
            j->out_dist[(int64_t)qi * j->k + t] = FLT_MAX;
        }
    }
    free(h);
    return NULL;
}

double lbo_cpu_baseline(int metric, const float *queries, int nq, const float *flat,
                        int64_t n, int dims, int k, int nthreads, int simd,
                        int64_t *out_ids, float *out_dist)
{
    if (nq <= 0 || k <= 0) return 0.0;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nq) nthreads = nq;
    job *jobs = (job *)calloc((size_t)nthreads, sizeof(job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; t++) {
        job jb = { metric, dims, k, (int)((int64_t)nq * t / nthreads),
                   (int)((int64_t)nq * (t + 1) / nthreads), simd,
                   queries, flat, n, out_ids, out_dist };
        jobs[t] = jb;
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(jobs);
    free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
