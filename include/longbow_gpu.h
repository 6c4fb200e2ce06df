/*
 * longbow_gpu.h -- C ABI of liblongbow_gpu.so: the MI355X (gfx950) k-NN
 * distance / top-k / PQ-ADC backend that drops in behind Longbow's
 * internal/gpu plug-point and keeps internal/simd's metric semantics.
 *
 * Plain pointers and sizes only; every call returns an lb_status code (0 = ok)
 * and never aborts the process (reference convention: non-zero -> Go
 * fmt.Errorf("... code %d"), NULL handle = init failure;
 * internal/gpu/faiss_gpu.go:56-66,99-101,134-137).
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to 23skdu/longbow).  The cgo stub a maintainer adds is in
 * INTEGRATION.md and go/internal/gpu/hip_gpu.go.
 *
 * Pointer naming: plain names are HOST pointers, borrowed for the call only
 * (cgo pointer rules: the library copies/DMAs during the call and retains
 * nothing).  `d_` names are DEVICE (HBM) pointers on the index's device.
 */
#ifndef LONGBOW_GPU_H
#define LONGBOW_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* simd.MetricType values (internal/simd/registry.go:8-15); search returns the
 * value Longbow ranks by, ascending: L2 with sqrt, 1-cos, NEGATED dot
 * (internal/store/distance_resolvers.go:12-16,74-83). */
typedef enum {
    LB_METRIC_EUCLIDEAN = 0,
    LB_METRIC_COSINE = 1,
    LB_METRIC_DOT = 2
} lb_metric;

/* f32 accumulation order of the reported distances (both exist in the reference):
 * SEQ     one accumulator, i ascending (internal/simd/simd_test.go:13-33,
 *         simd.go:138-163) -- the canonical oracle order, default;
 * UNROLL4 four accumulators (internal/simd/simd.go:365-479,
 *         distance_functions.go:195-227) -- what EuclideanDistanceBatchFlat runs. */
typedef enum { LB_ORDER_SEQ = 0, LB_ORDER_UNROLL4 = 1 } lb_order;

typedef enum {
    LB_OK = 0,
    LB_ERR_INVALID_ARG = 1, /* NULL / negative / mismatched sizes                  */
    LB_ERR_CLOSED = 2,      /* "index is closed" (faiss_gpu.go:79,111)             */
    LB_ERR_NO_DEVICE = 3,   /* no HIP device / bad device id (ErrGPUNotAvailable)   */
    LB_ERR_HIP = 4,         /* a HIP runtime call failed; see lb_gpu_last_error     */
    LB_ERR_OOM = 5,         /* HBM or pinned-host allocation failed                 */
    LB_ERR_UNSUPPORTED = 6, /* e.g. PQ with K != 256 (simd.go:350 stride)           */
    LB_ERR_INTERNAL = 7,
    LB_ERR_CANCELLED = 8,   /* the call's lb_cancel fired: ctx.Err() == context.Canceled          */
    LB_ERR_DEADLINE = 9     /* its deadline passed:        ctx.Err() == context.DeadlineExceeded  */
} lb_status;

typedef struct lb_gpu_index lb_gpu_index; /* opaque; replaces FaissGpuResourcesPtr +
                                             FaissGpuIndexFlatL2Ptr (faiss_gpu.go:12-13) */

/* ---- library ------------------------------------------------------------ */
int lb_gpu_device_count(void);          /* 0 when no GPU is visible; never fails */
const char *lb_gpu_version(void);
const char *lb_gpu_status_string(int status);

/* ---- gpu.Index: NewIndexWithConfig / Add / Search / Close ------------------
 * Replaces faiss_gpu_resources_new + faiss_gpu_index_flat_l2_new
 * (internal/gpu/faiss_gpu.go:16-19,44-72; interface.go:3-19).  dim <= 0 or an
 * unknown metric -> NULL (gpu_test.go:49-55).  out_status (nullable) receives
 * the reason.  dim > LB_MAX_DIM -> NULL / LB_ERR_UNSUPPORTED (the kernels stage one query row in LDS;
 * the reference has no cap, embedding widths in practice are <= 4096). */
#define LB_MAX_DIM 8192
#define LB_MAX_K 2048 /* largest k of any search entry point (LB_ERR_UNSUPPORTED beyond) */
lb_gpu_index *lb_gpu_index_new(int device, int dim, int metric, int *out_status);

/* Close (faiss_gpu.go:147-167): frees HBM; idempotent on NULL.  Must not race with any other call on the same
 * handle: the caller's lock orders Close after the last Add / Search (the Go shim's RWMutex does, as
 * faiss_gpu.go:148 does); a call on a freed handle is a use-after-free like with any C handle. */
void lb_gpu_index_free(lb_gpu_index *h);

/* Text of the last failure on this handle ("" if none).  Valid until the next
 * failing call on the same handle.  Concurrent host-pointer searches of a few queries are combined into one
 * device batch (lb_gpu_index_set_combining): this text, last_fallbacks and last_route describe the last device
 * BATCH on the handle, which may have held other callers' queries.  Return codes are per caller: a combined
 * batch that fails is searched again request by request before anybody is told. */
const char *lb_gpu_last_error(const lb_gpu_index *h);

int lb_gpu_index_set_order(lb_gpu_index *h, int order);

/* How the batched path generates candidates before the exact f32 re-rank.  Results are identical in every mode:
 * the re-rank recomputes every reported distance in the reference's f32 order and a rounding-error bound proves
 * that the candidate set contains the true top-k, else the query is re-done by the exact scan.
 *   LB_CAND_AUTO       (default) the cheapest route at every batch size: narrow tiles (HBM-bound passes) for small
 *                      batches, beyond them q.x as hi*hi + hi*lo + lo*hi on the bf16 MFMA with BOTH f32 operands split
 *                      into bf16 pairs in registers after the LDS read (no second copy of the corpus).
 *   LB_CAND_F32_MFMA   strict: beyond 384 queries q.x on the f32 MFMA (v_mfma_f32_32x32x2_f32); batches of
 *                      5..384 queries as AUTO.  The headline of bench.py is measured in this mode.
 *   LB_CAND_SPLIT_BF16 the split contraction over a pre-split image of the corpus (x = hi + lo + O(2^-18));
 *                      costs a second N*dim*4-byte copy in HBM; dim % 32 == 0.
 *   LB_CAND_SPLIT_BF16_INREG the in-register split for every batch beyond the narrow tiles.
 *   LB_CAND_F16        beyond 64 queries q.x as ONE fp16 product per element (corpus rounded to fp16 in registers, each
 *                      query scaled by a power of two): a third of the matrix work, error bound ~1.1e-3 |q||x| -- still
 *                      inside what the containment proof needs on embedding-like data; AUTO takes this route by itself
 *                      while the corpus norms allow it (max |x| <= 2^13, smallest non-zero |x| >= 2^-6), otherwise the
 *                      split contraction. */
typedef enum { LB_CAND_F32_MFMA = 0, LB_CAND_SPLIT_BF16 = 1, LB_CAND_SPLIT_BF16_INREG = 2, LB_CAND_AUTO = 3, LB_CAND_F16 = 4 } lb_candidate_mode;
int lb_gpu_index_set_candidate_mode(lb_gpu_index *h, int mode);
/* The fp16 route's own copy of the corpus (2 bytes per element, K-blocked): half the bytes to stage per batched search of
 * more than 64 queries (1M x 768, 1024 queries: see LABNOTES.md 4.2).  mode 1 (default): kept while the route is on offer for
 * this index (LB_CAND_AUTO from 16,384 rows, or LB_CAND_F16; norms in range; any dimension: the copy is laid out in
 * zero-padded planes of 32 dimensions, which is also what opens the matrix-core routes to dimensions like 100 or 300) and the copy leaves
 * max(2 GiB, 1/16 of the device) free -- brought up to date inside Add, dropped when the conditions end; mode 0: never (the
 * route then rounds the f32 rows in registers).  Results do not depend on it: both forms see the same fp16 values, and every
 * reported distance comes from the exact f32 re-rank.  There is no reference counterpart (a memory-for-speed knob of this
 * backend, like the index's capacity reservation). */
int lb_gpu_index_set_f16_image(lb_gpu_index *h, int mode);
int64_t lb_gpu_index_f16_image_bytes(const lb_gpu_index *h); /* HBM held by that copy right now (0: none) */

/* Concurrent host-pointer searches (lb_gpu_index_search with at most 16 queries and no cancellation context -- the reference's
 * gpu.Index.Search is ONE query per call, from many goroutines: internal/gpu/faiss_gpu.go:108-145) are COMBINED: one or two
 * callers search at once; callers that arrive while the device is taken queue, and the first of them then answers everybody who
 * queued with the same k by ONE batched search (at most 256 queries); after a combined batch the next caller waits up to 50 us
 * for the others to come back before it launches.  A batch's lists are the single searches' lists bit for bit, so only the clock
 * can tell: sixteen threads of single-query searches on 1M x 768 go from 4.2 k to 40 k queries/s, p50 3.7 -> 0.40 ms.  On by
 * default; 0 switches it off for the handle.  stats: out[0] = combined batches run, out[1] = requests they answered. */
int lb_gpu_index_set_search_combining(lb_gpu_index *h, int enable);
int lb_gpu_index_combining_stats(const lb_gpu_index *h, int64_t out[2]);
/* rows committed so far; safe beside Add / Search from other threads (read under the handle's reader lock, so it waits out an Add
 * in progress and never reports half a chunk) */
int64_t lb_gpu_index_ntotal(const lb_gpu_index *h);
int lb_gpu_index_dim(const lb_gpu_index *h);
int lb_gpu_index_device(const lb_gpu_index *h); /* GPUConfig.DeviceID (interface.go:15-19); -1 on NULL */

/* Pre-size HBM for n_total rows (optional; add grows geometrically otherwise). */
int lb_gpu_index_reserve(lb_gpu_index *h, int64_t n_total);

/* Add (replaces faiss_gpu_index_add, faiss_gpu.go:20,75-104): APPENDS n rows of
 * row-major f32[n*dim] -- the values buffer of an Arrow FixedSizeList<float32>
 * column (internal/store/adaptive_index.go:276-315).  ids nullable: labels are
 * then insertion positions (what the FAISS binding does, faiss_gpu.go:93-97).
 * The host buffer is staged through library-owned pinned memory and DMA'd. */
int lb_gpu_index_add(lb_gpu_index *h, int64_t n, const float *vectors, const int64_t *ids);
/* Same with the rows already in HBM on the index's device (D2D append). */
int lb_gpu_index_add_device(lb_gpu_index *h, int64_t n, const float *d_vectors,
                            const int64_t *d_ids);

/* Search (replaces faiss_gpu_index_search, faiss_gpu.go:21,107-144; semantics of
 * BruteForceIndex.SearchVectors, internal/store/adaptive_index.go:161-225):
 * exact k-NN of nq queries, results ascending by (distance, row position).
 * dist/labels are nq*k; fewer than k hits -> label -1 / dist FLT_MAX padding.
 * Thread-safe for concurrent calls on one handle (faiss_gpu.go:108 RLock). */
int lb_gpu_index_search(lb_gpu_index *h, int64_t nq, const float *queries, int k,
                        float *dist, int64_t *labels);
/* Queries and results resident in HBM; `stream` is a hipStream_t (NULL = the
 * library's own).  Results are complete when the call returns. */
int lb_gpu_index_search_device(lb_gpu_index *h, int64_t nq, const float *d_queries, int k,
                               float *d_dist, int64_t *d_labels, void *stream);

/* ---- cancellation: the ctx of SearchVectors(ctx, ...) ------------------------------------------
 * The reference's brute-force loop polls ctx.Err() every 1000 rows (internal/store/adaptive_index.go:182).
 * A cgo caller makes one lb_cancel per call, fires it from `go func(){ <-ctx.Done(); C.lb_cancel_fire(c) }()`
 * (or sets the ctx's deadline on it) and passes it to a *_ctx entry point.  The library polls it on the host
 * before every kernel launch of the search -- a launch covers at most one pass over <= 2.5M rows of one
 * batch (dense) or one query's pass over the codes (PQ), i.e. <= ~12 ms on 1M x 768 x 1024 queries -- stops
 * enqueuing, waits for what is already on the stream and returns LB_ERR_CANCELLED / LB_ERR_DEADLINE; the
 * output buffers then hold unspecified values.  NULL ctx = never cancelled.  All calls are thread-safe. */
typedef struct lb_cancel lb_cancel;
lb_cancel *lb_cancel_new(void);
void lb_cancel_fire(lb_cancel *c);                            /* idempotent                           */
void lb_cancel_set_deadline_ms(lb_cancel *c, int64_t ms_from_now); /* < 0 clears the deadline          */
int lb_cancel_state(const lb_cancel *c);                      /* 0, LB_ERR_CANCELLED or LB_ERR_DEADLINE */
void lb_cancel_free(lb_cancel *c);                            /* after the call that used it returned */
int lb_gpu_index_search_ctx(lb_gpu_index *h, int64_t nq, const float *queries, int k, float *dist, int64_t *labels,
                            const lb_cancel *ctx);
int lb_gpu_index_search_device_ctx(lb_gpu_index *h, int64_t nq, const float *d_queries, int k, float *d_dist,
                                   int64_t *d_labels, void *stream, const lb_cancel *ctx);

/* Metadata predicate mask for filtered search (SURVEY f-3; byte-per-row 0/1 as
 * internal/query/filter_evaluator.go:79-115 produces).  mask has ntotal bytes;
 * NULL clears it.  Rows with mask 0 never appear in results. */
int lb_gpu_index_set_filter(lb_gpu_index *h, const uint8_t *mask, int64_t n);

/* simd.CompareOp values (internal/simd/simd.go:38-45) */
typedef enum { LB_CMP_EQ = 0, LB_CMP_NEQ = 1, LB_CMP_GT = 2, LB_CMP_GE = 3, LB_CMP_LT = 4, LB_CMP_LE = 5 } lb_compare_op;

/* Evaluate a metadata predicate ON THE DEVICE straight into the index's row mask: the GPU form of
 * query.int64FilterOp/float32FilterOp.MatchBitmap (internal/query/filter_evaluator.go:79-115,205-241).
 * column has ntotal values (host pointer, e.g. an Arrow Int64/Float32 values buffer); validity is the
 * Arrow validity bitmap (LSB first; NULL = no nulls; nulls never match); combine 0 replaces the
 * mask, 1 ANDs into it (FilterEvaluator's AND chain, simd.AndBytes). */
int lb_gpu_index_filter_int64(lb_gpu_index *h, const int64_t *column, int64_t n, int64_t value, int op,
                              const uint8_t *validity, int64_t validity_offset, int combine);
int lb_gpu_index_filter_float32(lb_gpu_index *h, const float *column, int64_t n, float value, int op,
                                const uint8_t *validity, int64_t validity_offset, int combine);

/* Per-search telemetry of the most recent search on this handle (for benches):
 * number of queries that needed the exact-scan fallback. */
int64_t lb_gpu_index_last_fallbacks(const lb_gpu_index *h);
/* Cumulative count of small-batch searches (1..32 queries) whose in-launch threshold hand-off gave up after its
 * ~1 ms bound (the launch's workgroups were not co-resident, e.g. many such launches from many streams at once) and
 * were redone on the exact path: a latency event worth a metric, never a correctness one. */
int64_t lb_gpu_index_fused_giveups(const lb_gpu_index *h);
/* Which kernel generated the candidates of the most recent batched search on this handle: kind * 10 + operand form.
 * kind: 0 exact scan path (<= 4 queries, non-finite data), 1 / 2 narrow tile (32 / 64 queries per pass), (3: retired
 * in round 4), 4 128 x 128 f32-MFMA tile, 5 256 x 256 split-bf16 tile, 6 256 x 256 fp16 single-product tile;
 * form: 0 f32 operands, 1 pre-split corpus image, 2 split in registers, 3 fp16.  Telemetry only. */
int lb_gpu_index_last_route(const lb_gpu_index *h);

/* Candidate re-rank: the distance step of processChunkInternal
 * (internal/store/parallel_search.go:274-364).  The reference gathers the candidates' vectors into a
 * flat buffer and calls simd.EuclideanDistanceBatchFlat (:347), then emits
 * SearchResult{ID, Distance: d, Score: 1/(1+d)} (:355-362).  Here the candidate rows are gathered from the
 * corpus already resident in HBM (no PCIe re-upload): rows[i] is a row POSITION (the label Search returns
 * when Add got no ids, faiss_gpu.go:93-97); dist[i] is the index metric's distance in the requested
 * accumulation order (LB_ORDER_UNROLL4 is what the reference runs here; -1 = the index's own order),
 * score[i] = 1/(1+dist[i]) in f32 (nullable).  A row outside [0, ntotal) reports FLT_MAX / 0 -- the value
 * EuclideanDistanceBatch gives a nil vector (internal/simd/batch_operations.go:39-42).  The caller sorts
 * and truncates (parallel_search.go:123-130). */
int lb_gpu_index_rerank(lb_gpu_index *h, const float *query, const int64_t *rows, int64_t n, int order,
                        float *dist, float *score);
int lb_gpu_index_rerank_device(lb_gpu_index *h, const float *d_query, const int64_t *d_rows, int64_t n, int order,
                               float *d_dist, float *d_score, void *stream);

/* ---- internal/simd batch interface on the GPU --------------------------------
 * simd.EuclideanDistanceBatchFlat / CosineDistanceBatch / DotProductBatch
 * (internal/simd/batch_operations.go:64-87,131-157): one query x n rows of
 * flat[n*dims] -> results[n], in the requested accumulation order, bit-exact
 * with the scalar formulas.  metric DOT returns the RAW dot product here (as
 * simd.DotProduct does, distance_functions.go:59-73), not the negated value. */
int lb_simd_distance_batch_flat(int device, int metric, int order, const float *query,
                                const float *flat, int64_t n, int dims, float *results);
int lb_simd_distance_batch_flat_device(int device, int metric, int order, const float *d_query,
                                       const float *d_flat, int64_t n, int dims,
                                       float *d_results, void *stream);
/* simd.EuclideanDistanceBatch / CosineDistanceBatch / DotProductBatch over a SLICE OF SLICES
 * (internal/simd/batch_operations.go:29-60,131-157): vectors[i] points at vector i (host), lens[i] is its
 * length.  Per-vector rules of the reference:
 *   Euclidean  a nil or length-mismatched vector yields math.MaxFloat32 (batch_operations.go:39-42,51;
 *              simd_amd64.go:217-220);
 *   Cosine/Dot a nil vector is skipped -- results[i] keeps the value the caller put there (simd.go:241-267) --
 *              and a length mismatch makes the reference's loop stop at that vector with the error swallowed
 *              (batch_operations.go:140,155): results[i..] keep the caller's values.
 * The valid vectors are packed into one pinned block, DMA'd and scored by the same kernel as the flat call. */
int lb_simd_distance_batch(int device, int metric, int order, const float *query, int dims,
                           const float *const *vectors, const int *lens, int64_t n, float *results);

/* simd.MatchInt64 / simd.MatchFloat32 (internal/simd/simd.go:570-761): dst[i] = src[i] OP val ? 1 : 0,
 * and simd.AndBytes (simd.go:119-125): dst[i] &= src[i].  Host pointers. */
int lb_simd_match_int64(int device, const int64_t *src, int64_t n, int64_t value, int op, uint8_t *dst);
int lb_simd_match_float32(int device, const float *src, int64_t n, float value, int op, uint8_t *dst);
int lb_simd_and_bytes(int device, uint8_t *dst, const uint8_t *src, int64_t n);

/* ---- internal/pq on the GPU ---------------------------------------------------
 * Codebooks arrive as the reference's serialised blob (internal/pq/persistence.go:9-35:
 * u32 LE dims, M, K, then M*K*SubDim f32 LE).  K must be 256 (simd.go:350). */
typedef struct lb_gpu_pq lb_gpu_pq;
lb_gpu_pq *lb_gpu_pq_new(int device, const uint8_t *codebook_blob, size_t len, int *out_status);
void lb_gpu_pq_free(lb_gpu_pq *p);
const char *lb_gpu_pq_last_error(const lb_gpu_pq *p);
int lb_gpu_pq_m(const lb_gpu_pq *p);
int lb_gpu_pq_dims(const lb_gpu_pq *p);
int64_t lb_gpu_pq_ntotal(const lb_gpu_pq *p);
/* Append n codes u8[n*M] (row-major, as pq.Encode emits them). */
int lb_gpu_pq_add_codes(lb_gpu_pq *p, int64_t n, const uint8_t *codes);
int lb_gpu_pq_add_codes_device(lb_gpu_pq *p, int64_t n, const uint8_t *d_codes);
int lb_gpu_pq_reserve(lb_gpu_pq *p, int64_t n_total);
/* Read back stored codes rows [row0, row0+n) as u8[n*M] (e.g. to persist what add_vectors_device encoded). */
int lb_gpu_pq_get_codes(lb_gpu_pq *p, int64_t row0, int64_t n, uint8_t *codes);
/* pq.(*PQEncoder).Encode (internal/pq/encoder.go:76-136) for n vectors of f32[dims] -> codes u8[n*M]:
 * per subspace the FIRST centroid with the strictly smallest float32(sqrt(float64(L2^2))) in the
 * 4-accumulator order, as simd.FindNearestCentroid's K > 8 branch computes it
 * (internal/simd/simd.go:305-326, 365-396).  (K <= 16 codebooks -- the encodeSequential branch -- are
 * rejected at lb_gpu_pq_new with LB_ERR_UNSUPPORTED: K must be 256.) */
int lb_gpu_pq_encode(lb_gpu_pq *p, int64_t n, const float *vectors, uint8_t *codes);
int lb_gpu_pq_encode_device(lb_gpu_pq *p, int64_t n, const float *d_vectors, uint8_t *d_codes, void *stream);
/* Encode n device-resident vectors and append their codes (the ingest step that produces config 4's codes). */
int lb_gpu_pq_add_vectors_device(lb_gpu_pq *p, int64_t n, const float *d_vectors);
/* pq.(*PQEncoder).Decode (encoder.go:139-158): codes u8[n*M] -> vectors f32[n*dims]. */
int lb_gpu_pq_decode(lb_gpu_pq *p, int64_t n, const uint8_t *codes, float *vectors);
int lb_gpu_pq_decode_device(lb_gpu_pq *p, int64_t n, const uint8_t *d_codes, float *d_vectors, void *stream);
/* processChunkInternal's PQ branch (internal/store/parallel_search.go:292-345): ADC distance of the stored
 * code rows rows[0..n) to `query` (table built per call) + score = 1/(1+d) (nullable); rows outside
 * [0, ntotal) report FLT_MAX / 0. */
int lb_gpu_pq_rerank(lb_gpu_pq *p, const float *query, const int64_t *rows, int64_t n, float *dist, float *score);
int lb_gpu_pq_rerank_device(lb_gpu_pq *p, const float *d_query, const int64_t *d_rows, int64_t n, float *d_dist,
                            float *d_score, void *stream);
/* pq.BuildADCTable (internal/pq/adc_table.go:15-51): table f32[M*K] for one query. */
int lb_gpu_pq_build_adc_table(lb_gpu_pq *p, const float *query, float *table);
/* simd.ADCDistanceBatch (internal/simd/batch_operations.go:119-127, simd.go:345-355):
 * results[i] = float32(sqrt(float64(sum_j table[j*256+codes[i*M+j]]))) over the stored codes
 * rows [row0, row0+n). */
int lb_gpu_pq_adc_distance_batch(lb_gpu_pq *p, const float *table, int64_t row0, int64_t n,
                                 float *results);
/* ADC k-NN over all stored codes: builds the table per query, scans, top-k ascending by
 * (distance, position).  nq queries of f32[dims]. */
int lb_gpu_pq_search(lb_gpu_pq *p, int64_t nq, const float *queries, int k, float *dist,
                     int64_t *labels);
int lb_gpu_pq_search_device(lb_gpu_pq *p, int64_t nq, const float *d_queries, int k,
                            float *d_dist, int64_t *d_labels, void *stream);
/* the same with a cancellation context, polled between the queries of the batch and the launches of one query */
int lb_gpu_pq_search_ctx(lb_gpu_pq *p, int64_t nq, const float *queries, int k, float *dist, int64_t *labels,
                         const lb_cancel *ctx);
/* Concurrent host-pointer ADC searches are combined like lb_gpu_index_search's (lb_gpu_index_set_search_combining): queued calls
 * with the same k are answered by one batch, in which two queries share each pass over the codes.  On by default. */
int lb_gpu_pq_set_search_combining(lb_gpu_pq *p, int enable);
int lb_gpu_pq_combining_stats(const lb_gpu_pq *p, int64_t out[2]);
int lb_gpu_pq_search_device_ctx(lb_gpu_pq *p, int64_t nq, const float *d_queries, int k, float *d_dist,
                                int64_t *d_labels, void *stream, const lb_cancel *ctx);
/* 1 (default): the search runs the byte-table prefilter + exact survivors; 0: the exact f32-table
 * pass over every row.  Results are bit-identical; the switch exists for A/B timing and for the parity tests. */
int lb_gpu_pq_set_prefilter(lb_gpu_pq *p, int enable);

/* instrumentation (bench.py): HIP-event times of the most recent profiled search on this handle, recorded on the
 * search stream: ms[0] = the pass over the codes of the LAST query (prefilter kernel, or the exact kernel when
 * the prefilter is off), ms[1] = the whole search on the device.  Enable before the search; one search at a time. */
int lb_gpu_pq_set_profiling(lb_gpu_pq *p, int enable);
int lb_gpu_pq_last_timing(const lb_gpu_pq *p, float ms[2]);

/* ---- cross-shard merge ---------------------------------------------------------
 * store.MergeSortedStreams (internal/store/result_merger.go:34-101) for S shards:
 * inputs [S][nq][k] ascending per (shard, query) (padding label -1 / FLT_MAX allowed),
 * output [nq][k] ascending by (distance, label); padding sorts last, every NaN after +inf.
 * nshards * k <= 16384 (k = 2048 on 8 GPUs).  Device pointers. */
int lb_gpu_merge_topk_device(int device, int nshards, int64_t nq, int k, const float *d_dist_in,
                             const int64_t *d_labels_in, float *d_dist_out, int64_t *d_labels_out,
                             void *stream);

/* Same merge over ONE gathered buffer (a single RCCL all-gather per batch): shard s's block starts at
 * d_packed + s*block_bytes and holds nq*k int64 labels followed by nq*k f32 distances, the block
 * size rounded up to 8 bytes: block_bytes = nq*k*8 + ceil(nq*k*4 / 8)*8. */
int lb_gpu_merge_topk_packed_device(int device, int nshards, int64_t nq, int k, const void *d_packed,
                                    float *d_dist_out, int64_t *d_labels_out, void *stream);

/* ---- multi-GPU search -------------------------------------------------------------------
 * One GPU per shard; per batch: shard search -> ONE all-gather of the packed per-shard top-k (RCCL over xGMI,
 * nq*k*12 bytes per rank) -> device merge.  Semantics of ShardedHNSW.SearchVectors' fan-out + concat + sort
 * (internal/store/sharded_hnsw.go:414-503) / MergeSortedStreams (result_merger.go:34-101); which rows a
 * shard holds is the host's business (store.RingSharder, sharding_strategy.go:40-127).  SURVEY 8(b)'s
 * lb_gpu_comm_init(ndev) is lb_gpu_comm_init_all.  ranks * k <= 16384. */
typedef struct lb_gpu_comm lb_gpu_comm;
/* host-supplied all-gather between HOST buffers: recv holds nranks blocks of `bytes` in rank order; 0 = ok */
typedef int (*lb_allgather_fn)(void *ctx, const void *send, void *recv, size_t bytes);
#define LB_COMM_UNIQUE_ID_BYTES 128
/* ONE process drives ndev GPUs (the Go server): devices NULL = 0..ndev-1.  RCCL (ncclCommInitAll). */
lb_gpu_comm *lb_gpu_comm_init_all(int ndev, const int *devices, int *out_status);
/* one process (or thread) per GPU: rank 0 calls get_unique_id and ships the 128 bytes to the others */
int lb_gpu_comm_get_unique_id(void *out128);
lb_gpu_comm *lb_gpu_comm_init_rank(int device, int nranks, int rank, const void *unique_id, int *out_status);
/* one process per GPU with the exchange done by the host (gloo / MPI / gRPC), staged through pinned memory */
lb_gpu_comm *lb_gpu_comm_init_host(int device, int nranks, int rank, lb_allgather_fn fn, void *ctx, int *out_status);
void lb_gpu_comm_free(lb_gpu_comm *c);
int lb_gpu_comm_nranks(const lb_gpu_comm *c);
int lb_gpu_comm_rank(const lb_gpu_comm *c);
const char *lb_gpu_comm_last_error(const lb_gpu_comm *c);
/* Size the exchange buffers (device blocks, pinned host blocks of the host transport, per-device query buffers of init_all)
 * for searches of up to nq_max queries and k_max results.  NOT collective: each rank calls it on its own, right after init,
 * and acts on its return code before any search is issued.  A search within the prepared size allocates nothing between
 * the shard search and the exchange, so an out-of-memory condition on one rank cannot leave its peers waiting in the
 * all-gather.  (A larger search still grows the buffers on the fly, with that risk.) */
int lb_gpu_comm_prepare(lb_gpu_comm *c, int64_t nq_max, int k_max);
/* init_rank / init_host communicators: this rank's shard h, the same nq queries on every rank (device
 * pointer), global top-k on every rank.  Collective: every rank must call it with the same nq and k. */
int lb_gpu_comm_search_device(lb_gpu_comm *c, lb_gpu_index *h, int64_t nq, const float *d_queries, int k, float *d_dist,
                              int64_t *d_labels, void *stream);
/* init_all communicators: shards[i] lives on the communicator's i-th device; host queries in, host results out */
int lb_gpu_comm_search_all(lb_gpu_comm *c, lb_gpu_index *const *shards, int64_t nq, const float *queries, int k, float *dist,
                           int64_t *labels);

/* ---- Arrow Flight framing (SURVEY f-1 / f-2) -----------------------------------------------------
 * The bytes of the Flight messages in, the bytes of the response out: a Go (or C) host needs no Arrow glue.
 * Return values of the lb_flight_* calls that parse requests are gRPC status codes, as the reference handler
 * returns them (0 OK, 3 InvalidArgument, 5 NotFound, 9 FailedPrecondition, 13 Internal, 14 Unavailable), with
 * the reference's message text in errbuf (nullable). */
typedef struct lb_flight_datasets lb_flight_datasets; /* name -> index (VectorStore.getDataset) */
lb_flight_datasets *lb_flight_datasets_new(void);
void lb_flight_datasets_free(lb_flight_datasets *r);
/* register (h != NULL) or remove (h == NULL) a dataset; the index stays owned by the caller */
int lb_flight_datasets_put(lb_flight_datasets *r, const char *name, lb_gpu_index *h);
/* VectorStore.handleVectorSearchExchange (internal/store/vector_search_exchange.go:31-217): ipc_in = an Arrow
 * IPC stream with ONE request batch {dataset utf8, k int32 (default 10), ef int32 (ignored), query_vector
 * FixedSizeList<float32> | List<float32>}, row 0 only; *ipc_out = an IPC stream with the response batch
 * {id uint64, score float32} (min(k, N) rows), to be released with lb_flight_free_buffer. */
int lb_flight_vector_search_exchange(lb_flight_datasets *reg, const uint8_t *ipc_in, size_t len_in, uint8_t **ipc_out,
                                     size_t *len_out, char *errbuf, size_t errcap);
/* One result batch {id uint64, score float32} as IPC stream bytes (trailing -1 labels trimmed): the per-query
 * flight.Result of DoAction("VectorSearch") (internal/store/vector_search_action.go:180-231).  lb_status. */
int lb_flight_encode_results(const int64_t *ids, const float *scores, int64_t n, uint8_t **ipc_out, size_t *len_out);
void lb_flight_free_buffer(uint8_t *p);
/* Append every record batch of an IPC stream: column "vector" FixedSizeList<float32>[dim] (its values buffer
 * goes to lb_gpu_index_add as it lies in the message body) and optional "id" (uint32 / uint64 / int64,
 * truncated to the reference's uint32 VectorID).  internal/store/store_lifecycle.go:66-76. */
int lb_flight_index_add_ipc(lb_gpu_index *h, const uint8_t *ipc, size_t len, int64_t *rows_added, char *errbuf, size_t errcap);

/* ---- hybrid fusion ---------------------------------------------------------------------
 * store.ReciprocalRankFusion (internal/store/rrf.go:10-51) for nq queries at once: per query a dense
 * ranking ids[kd] and a sparse ranking ids[ks] (best first, -1 = padding, ids unique within a list);
 * score(id) = sum 1/float64(k + rank + 1) (k <= 0 -> 60), narrowed to f32; output limit ids/scores per
 * query, score descending (ties: lower id first), padded with -1 / 0.  kd + ks <= 8192. */
int lb_gpu_rrf_fuse_device(int device, int64_t nq, int kd, const int64_t *d_dense_ids, int ks,
                           const int64_t *d_sparse_ids, int k, int limit, int64_t *d_out_ids,
                           float *d_out_scores, void *stream);
int lb_gpu_rrf_fuse(int device, int64_t nq, int kd, const int64_t *dense_ids, int ks, const int64_t *sparse_ids,
                    int k, int limit, int64_t *out_ids, float *out_scores);

/* ---- synthetic data (bench / tests) ------------------------------------------------
 * Counter-based uniform [0,1) f32 / uniform u8, bit-identical to the oracle's
 * lbo_fill_uniform / lbo_fill_codes. */
int lb_gpu_fill_uniform_device(int device, float *d_dst, int64_t n, uint64_t seed, int64_t offset,
                               void *stream);
int lb_gpu_fill_codes_device(int device, uint8_t *d_dst, int64_t n, uint64_t seed, int64_t offset,
                             void *stream);
/* rows of a global synthetic corpus picked by id: dst[r][j] = uniform(seed, ids[r]*dim + j) -- what a shard of
 * a ring-partitioned corpus holds (bench.py) */
int lb_gpu_fill_uniform_rows_device(int device, float *d_dst, const int64_t *d_ids, int64_t nrows, int dim, uint64_t seed,
                                    void *stream);

/* Shader clock of `device` right now, in MHz: every CU spins for `spin_us` microseconds and compares its cycle counter
 * (s_memtime) with the constant 100 MHz counter (s_memrealtime).  bench.py calls it straight behind its timed steps so a
 * line says at which clock state it was measured (the same search runs 1.67-1.95 ms by clock state on one box).
 * Returns a negative status code (-LB_ERR_*) on failure. */
double lb_gpu_shader_clock_mhz(int device, int spin_us);

/* ---- instrumentation (bench.py roofline leg) ----------------------------------------
 * HIP-event timing of the dominant kernels of the most recent search on this handle,
 * recorded on the stream the kernels ran on.  Enable before the search. */
int lb_gpu_index_set_profiling(lb_gpu_index *h, int enable);
/* ms spent in: [0] candidate GEMM kernels, [1] select kernels, [2] re-rank, [3] scan
 * kernels, [4] whole search (device side).  n_launch[i] = launches of that class. */
int lb_gpu_index_last_timing(const lb_gpu_index *h, float ms[5], int n_launch[5]);

#ifdef __cplusplus
}
#endif
#endif
