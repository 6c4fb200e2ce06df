/*
 * longbow_oracle.h -- CPU restatement of the Longbow k-NN hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under longbow_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the baseline.
 *
 * Parity status: the reference is Go (+ Go assembly); no Go toolchain exists in
 * the build image, so oracle/_ref cannot be built.  This restatement is pinned
 * by the reference's own known-answer tests (tests/golden/ *.json, produced by
 * oracle/gen_golden.py from the literals/generators in the reference's tests)
 * and cross-checked by an independent numpy restatement (oracle/oracle_np.py).
 * Two items are "parity unpinned" (no reference test fixes them): the
 * RingSharder assignment and the sqrt form of ADCDistanceBatch; see DESIGN.md.
 *
 * Citations are relative to the reference tree (23skdu/longbow @ 2026-01-30).
 * All float arithmetic is IEEE binary32, one rounding per operation, no FMA
 * contraction (build with -ffp-contract=off), sqrt done in binary64 and
 * narrowed, exactly as the Go source spells it.
 */
#ifndef LONGBOW_ORACLE_H
#define LONGBOW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* simd.MetricType  (internal/simd/registry.go:8-15) */
enum { LBO_METRIC_EUCLIDEAN = 0, LBO_METRIC_COSINE = 1, LBO_METRIC_DOT = 2 };

/* Accumulation order of the f32 sums.
 * SEQ      = one accumulator, i ascending: the reference's test oracles
 *            referenceEuclidean/referenceCosine (internal/simd/simd_test.go:13-33),
 *            cosineGeneric / dotGeneric (internal/simd/simd.go:138-163).
 * UNROLL4  = four accumulators over i mod 4, tail into acc0, combined
 *            ((s0+s1)+s2)+s3: euclideanUnrolled4x / cosineUnrolled4x /
 *            dotUnrolled4x (internal/simd/simd.go:365-479) and
 *            L2SquaredFloat32 (internal/simd/distance_functions.go:195-227).
 */
enum { LBO_ORDER_SEQ = 0, LBO_ORDER_UNROLL4 = 1 };

/* ---- per-pair metrics ------------------------------------------------- */
float lbo_l2sq(const float *a, const float *b, int n, int order);      /* no sqrt */
float lbo_euclidean(const float *a, const float *b, int n, int order); /* sqrt applied */
float lbo_cosine(const float *a, const float *b, int n, int order);    /* 1 - cos */
float lbo_dot(const float *a, const float *b, int n, int order);       /* raw dot */

/* The value Longbow ranks by, ascending: L2 (sqrt), 1-cos, NEGATED dot
 * (internal/store/distance_resolvers.go:12-16,74-83; docs/distance_metrics.md:44-46). */
float lbo_distance(int metric, const float *q, const float *x, int n, int order);

/* simd.EuclideanDistanceBatchFlat and its cosine/dot analogues over a row-major
 * flat[n*dims] buffer (internal/simd/batch_operations.go:64-87,
 * internal/simd/simd.go:203-229).  out[i] = lbo_distance(metric, q, row i). */
void lbo_batch_flat(int metric, int order, const float *q, const float *flat,
                    int64_t n, int dims, float *out);

/* ---- brute-force k-NN -------------------------------------------------- */
/* BruteForceIndex.SearchVectors (internal/store/adaptive_index.go:161-225) with
 * Go's container/heap (up/down/Push/Pop restated): bounded max-heap, insert if
 * len<k else replace root iff dist < root (strict).  Output ascending.  Returns
 * the number of results (min(k, n)).  Ties at the k-th boundary follow the heap
 * layout exactly as in the reference. */
int lbo_bruteforce_goheap(int metric, int order, const float *q, const float *flat,
                          int64_t n, int dims, int k, int64_t *out_ids, float *out_dist);

/* Canonical top-k: ascending by (distance, row index): the specification the
 * GPU path implements ("lowest index wins" on ties).  Equal to the go-heap form
 * on tie-free data.  Returns min(k, n); unused tail entries get id -1 and
 * dist FLT_MAX (FAISS padding convention). */
int lbo_topk_canonical(const float *dist, int64_t n, int k, int64_t *out_ids, float *out_dist);

/* Batched canonical search: nq queries, each lbo_batch_flat + lbo_topk_canonical.
 * mask (nullable): byte per row, 0 = row excluded (metadata predicate filter).
 * ids (nullable): user ids reported instead of row positions.
 * nthreads <= 1 -> scalar single thread. */
void lbo_search_batch(int metric, int order, const float *queries, int nq,
                      const float *flat, int64_t n, int dims, int k,
                      const uint8_t *mask, const int64_t *ids,
                      int64_t *out_ids, float *out_dist, int nthreads);

/* ---- product quantisation ---------------------------------------------- */
/* pq.BuildADCTable (internal/pq/adc_table.go:15-51): table[i*K+j] =
 * L2SquaredFloat32(q_sub_i, centroid_ij).  codebooks flat [M][K][sub]. */
void lbo_build_adc_table(const float *codebooks, int M, int K, int sub,
                         const float *query, float *table);

/* simd.adcBatchGeneric (internal/simd/simd.go:345-355): out[i] =
 * float32(sqrt(float64(sum_j table[j*256 + codes[i*m+j]]))), f32 sum in j order.
 * NOTE stride 256 regardless of K (the reference's own inconsistency). */
void lbo_adc_batch(const float *table, const uint8_t *codes, int m, int64_t n, float *out);

/* pq.ADCDistance (internal/pq/adc_table.go:77-92): the sum WITHOUT sqrt, stride K. */
float lbo_adc_single(const float *table, const uint8_t *code, int M, int K);

/* pq.Encode (internal/pq/encoder.go:76-136) + simd.FindNearestCentroid
 * (internal/simd/simd.go:278-326): K<=16 sequential argmin of L2^2 with strict <;
 * K>16 argmin over euclideanUnrolled4x (sqrt'd) distances, strict <, first wins. */
void lbo_pq_encode(const float *codebooks, int M, int K, int sub,
                   const float *vec, uint8_t *codes);
/* pq.Decode (internal/pq/encoder.go:139-158) */
void lbo_pq_decode(const float *codebooks, int M, int K, int sub,
                   const uint8_t *codes, float *vec);
/* pq.DeserializePQEncoder header checks (internal/pq/persistence.go:38-73):
 * returns 0 and fills dims/M/K on success, <0 on the reference's error cases. */
int lbo_pq_parse_blob(const uint8_t *blob, size_t len, int *dims, int *M, int *K);

/* ---- sharding / merge --------------------------------------------------- */
/* RingSharder (internal/store/sharding_strategy.go:40-127): FNV-1a-32.
 * lbo_ring_build fills hashes[numShards*vnodes] sorted ascending and
 * owners[] = ring[hash] with Go map semantics (later shard overwrites on hash
 * collision).  Returns the number of ring points. */
uint32_t lbo_fnv1a32(const uint8_t *p, size_t n);
int lbo_ring_build(int num_shards, int vnodes, uint32_t *hashes, int *owners);
int lbo_ring_get_shard(const uint32_t *hashes, const int *owners, int npoints, uint64_t id);

/* store.MergeSortedStreams (internal/store/result_merger.go:34-101) with
 * container/heap: k-way merge of ascending lists; k<=0 = all.  lists are given
 * as concatenated arrays with per-list lengths.  Returns count written. */
int lbo_merge_sorted_streams(const int64_t *ids, const float *scores, const int *lens,
                             int nlists, int k, int64_t *out_ids, float *out_scores);

/* ---- predicate masks -------------------------------------------------------- */
/* simd.MatchInt64 / MatchFloat32 (internal/simd/simd.go:570-761): dst[i] = src[i] OP val ? 1 : 0;
 * op = simd.CompareOp (Eq=0, Neq, Gt, Ge, Lt, Le; simd.go:38-45).  simd.AndBytes (simd.go:119-125). */
void lbo_match_int64(const int64_t *src, int64_t n, int64_t val, int op, uint8_t *dst);
void lbo_match_float32(const float *src, int64_t n, float val, int op, uint8_t *dst);
void lbo_and_bytes(uint8_t *dst, const uint8_t *src, int64_t n);

/* ---- hybrid fusion ------------------------------------------------------------- */
/* store.ReciprocalRankFusion (internal/store/rrf.go:10-51): f64 accumulation dense then sparse,
 * Score = float32(sum); sorted by score descending (canonical tie order: lower id first).
 * ids < 0 are padding.  Returns the number of fused results written (<= limit if limit > 0). */
/* calculateAdaptiveLimit (internal/store/adaptive_search.go:7-39): search depth for post-filtered
 * search from the filter's selectivity: k * clamp(total/matches, 2, 50), clamped to [k, total];
 * k when total == 0 or matches == 0. */
int lbo_adaptive_limit(int k, uint64_t matches, int total);

int lbo_rrf(const int64_t *dense, int nd, const int64_t *sparse, int ns, int k, int limit,
            int64_t *out_ids, float *out_scores);

/* ---- synthetic data ------------------------------------------------------ */
/* Counter-based uniform [0,1) f32 generator shared bit-for-bit with the HIP
 * library (lb_gpu_fill_uniform): value(idx) = (splitmix64(seed ^ mix(idx)) >> 40) * 2^-24.
 * Matches the distribution of Go's rand.Float32() used by the reference's
 * benches (cmd/bench-tool/main.go:143,176). */
void lbo_fill_uniform(float *dst, int64_t n, uint64_t seed, int64_t offset);
void lbo_fill_codes(uint8_t *dst, int64_t n, uint64_t seed, int64_t offset);

/* ---- CPU baseline (bench.py cpu_baseline leg only) ----------------------- */
/* The reference's brute-force ground-truth idiom
 * (internal/store/recall_validation_test.go:237-300): queries partitioned over
 * worker threads, each streams the whole corpus per query, then top-k.
 * simd=0: scalar canonical order.  simd=1: 8-lane x 4-accumulator blocked
 * loops in the shape of the reference's AVX2 wrappers
 * (internal/simd/simd_amd64.go:21-63,125-140), auto-vectorised by the compiler.
 * Returns wall seconds. */
double lbo_cpu_baseline(int metric, const float *queries, int nq, const float *flat,
                        int64_t n, int dims, int k, int nthreads, int simd,
                        int64_t *out_ids, float *out_dist);

#ifdef __cplusplus
}
#endif
#endif
