// lb_device.h -- shared device helpers and kernel-launcher prototypes (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace lb {

// A/B tunables.  In the product build every tunable IS its compiled-in default (the measured winner);
// only a -DLB_DIAG build (python -m longbow_amd.build --diag -> liblongbow_gpu_diag.so, used by tools/)
// reads the LB_* environment variables and carries the timing-only ablation kernels.
inline int lb_tunable(const char *name, int dflt)
{
#ifdef LB_DIAG
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// ---------------------------------------------------------------------------
// Candidate entries: (key f32, row u32) packed into one u64 so that unsigned
// integer order == ascending (key, row).  All selection works on these.
// ---------------------------------------------------------------------------
constexpr uint64_t kEntryMax = ~0ull;

__host__ __device__ __forceinline__ uint32_t f32_sortable(float f)
{
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float sortable_f32(uint32_t s)
{
    uint32_t u = (s & 0x80000000u) ? (s & 0x7fffffffu) : ~s;
    return __builtin_bit_cast(float, u);
}
// Canonical order of the reported lists: ascending distance, every NaN after +inf (whatever its sign
// bit), ties by row.  (The reference's heap has no defined behaviour for NaN distances -- a NaN never
// displaces anything, adaptive_index.go:206 -- so this only fixes what it leaves open.)
__host__ __device__ __forceinline__ uint64_t pack_entry(float key, uint32_t row)
{
    key = key + 0.0f; // -0 -> +0
    const uint32_t sk = (key != key) ? 0xffc00000u /* sortable image of the positive quiet NaN */ : f32_sortable(key);
    return ((uint64_t)sk << 32) | row;
}
__host__ __device__ __forceinline__ float entry_key(uint64_t e) { return sortable_f32((uint32_t)(e >> 32)); }
__host__ __device__ __forceinline__ uint32_t entry_row(uint64_t e) { return (uint32_t)e; }
// Threshold in the (key, row) form the MFMA kernels test in the float domain.  "No threshold yet"
// (kEntryMax: e.g. a bootstrap chunk with fewer visible rows than candidates to keep) decodes to
// a NaN key, which would admit nothing -- it must admit every row instead.
__host__ __device__ __forceinline__ float tau_key_of(uint64_t tau)
{
    return tau == ~0ull ? __builtin_huge_valf() : entry_key(tau);
}

inline uint32_t next_pow2_host(uint32_t v)
{
    uint32_t p = 2;
    while (p < v) p <<= 1;
    return p;
}

// Chunk schedule shared by every top-k pipeline (host side).  Chunk 0 is the bootstrap chunk
// (all rows stored, no atomics); every later chunk is sized so that, with the threshold left by
// the previous select (pass rate ~ keep/pos on exchangeable data), about (cap-keep)/4 entries
// are admitted.  safe = chunks that cannot overflow whatever the data order.
// big_boot: use the whole list as bootstrap chunk (scan pipelines: one select level fewer).
inline int64_t chunk_end_host(int step, int64_t pos, int64_t n, int64_t keep, int64_t cap, bool safe,
                              bool big_boot = false)
{
    int64_t end;
    if (step == 0) {
        int64_t boot = 4 * keep > 2048 ? 4 * keep : 2048;
        if (boot > cap || big_boot) boot = cap;
        end = boot;
    } else if (safe) {
        end = pos + (cap - keep);
    } else {
        const int64_t target = (cap - keep) / 4;
        // rows = target * pos / keep, in 128-bit-safe steps
        const double rows = (double)target * (double)pos / (double)(keep > 0 ? keep : 1);
        end = rows > 4e18 ? n : pos + (int64_t)rows;
        if (end <= pos) end = pos + 1;
        if (n - end < end - pos) end = n; // do not leave a tail smaller than this chunk
    }
    return end < n ? end : n;
}

enum : int { METRIC_L2 = 0, METRIC_COS = 1, METRIC_DOT = 2 };
enum : int { ORDER_SEQ = 0, ORDER_UNROLL4 = 1 };

// Per-search candidate state, one slot per query (all in HBM).
struct CandState {
    uint64_t *lists; // [nq][cap] entries; [0,cnt) valid
    uint32_t *cnt;   // [nq] appended so far (may exceed cap -> overflow)
    uint64_t *tau;   // [nq] admission threshold: entry admitted iff e < tau
    uint32_t *flags; // [nq] bit0 = list overflowed, bit1 = containment bound failed, bit2 = sampled threshold too tight
    uint32_t cap;
    // Optional (scan path, sampled pass): 16 admission counters per slot, LB_STRIPE_PAD u32 apart.  Stripe s
    // hands out list positions s, s+16, s+32, ... so the admissions of one query do not serialise on a
    // single hot counter; the following select reads the counters and skips the holes.
    uint32_t *stripes;
};
constexpr int LB_STRIPES = 16;
constexpr int LB_STRIPE_PAD = 32; // u32 words between counters (128 B)

#ifdef __HIPCC__
// u64 minimum over lanes on the DPP path (no LDS round trips); shared by the threshold kernels
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t min_dpp_u64(uint64_t v)
{
    const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const uint64_t o = ((uint64_t)(uint32_t)ohi << 32) | (uint64_t)(uint32_t)olo;
    return o < v ? o : v;
}
// minimum over each row of 16 lanes, left in every lane of the row
__device__ __forceinline__ uint64_t row16_min_u64(uint64_t v)
{
    v = min_dpp_u64<0xB1, 0xf>(v);  // quad_perm [1,0,3,2]
    v = min_dpp_u64<0x4E, 0xf>(v);  // quad_perm [2,3,0,1]
    v = min_dpp_u64<0x141, 0xf>(v); // row_half_mirror
    v = min_dpp_u64<0x140, 0xf>(v); // row_mirror
    return v;
}
// minimum over the 64 lanes, returned wave-uniform
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
    v = row16_min_u64(v);
    v = min_dpp_u64<0x142, 0xa>(v); // row_bcast15: rows 1,3 <- lane 15 of rows 0,2
    v = min_dpp_u64<0x143, 0xc>(v); // row_bcast31: rows 2,3 <- lane 31
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}

// the same on 32-bit keys (one v_min_u32 with a DPP operand per step)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t min_dpp_u32(uint32_t v)
{
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
}
__device__ __forceinline__ uint32_t row16_min_u32(uint32_t v)
{
    v = min_dpp_u32<0xB1, 0xf>(v);
    v = min_dpp_u32<0x4E, 0xf>(v);
    v = min_dpp_u32<0x141, 0xf>(v);
    v = min_dpp_u32<0x140, 0xf>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = row16_min_u32(v);
    v = min_dpp_u32<0x142, 0xa>(v);
    v = min_dpp_u32<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

#endif

// ---------------------------------------------------------------------------
// Launchers (implemented in kernels_*.hip).  All are asynchronous on `s`.
// ---------------------------------------------------------------------------

// per-row ||x||^2 and 1/||x|| (0 if the norm is 0); max ||x||^2 folded into d_maxnorm2[0], the smallest NON-ZERO ||x||^2
// into d_maxnorm2[1] (float bits; initialise to {0, 0x7f800000})
// center (or null): norms of x - center instead (rnorm may then be null)
void launch_row_norms(const float *X, int64_t n, int D, float *norm2, float *rnorm,
                      uint32_t *d_maxnorm2, hipStream_t s, const float *center = nullptr);
// column means of X[0 .. n) in a fixed order -> center[0 .. Dpad) (zero beyond D); partial: [256][D] scratch
void launch_column_means(const float *X, int64_t n, int D, float *partial, float *center, int Dpad, hipStream_t s);

// candidate generation: f32 MFMA inner products of queries [nq][D] x rows [row_begin,row_end)
// -> metric key -> admit (key,row) < tau[q] into the query's list.
// boot: every row is stored at list[row - row_begin] (no admission test, no atomics).
void launch_gemm_filter(int metric, const float *X, const float *norm2, const float *rnorm,
                        int64_t row_begin, int64_t row_end, int D, const float *Q, int nq,
                        const uint8_t *mask, const uint32_t *rowmap, CandState cs, bool boot, int split,
                        hipStream_t s); // split: 0 f32 MFMA, 1 pre-split bf16 images, 2 f32 operands split in registers
// small/mid-size batches (5..384 queries): 256-row x 32-query tiles, HBM-bound; needs D % 32 == 0, 16-B aligned X/Q
void launch_gemm_filter_narrow(int metric, const float *X, const float *norm2, const float *rnorm,
                               int64_t row_begin, int64_t row_end, int D, const float *Q, int nq,
                               const uint8_t *mask, const uint32_t *rowmap, CandState cs, bool boot,
                               hipStream_t s, bool tile64 = false, // tile64: 128 rows x 64 queries per workgroup
                               bool split = false);                // split: 3 x bf16 MFMA on operands split in registers
// Sampled threshold fused into the candidate pass (kernels_gemm_narrow.hip, FUSED): the first `n_blocks` workgroups of
// the launch score `count` sampled rows (corpus rows smap[0..count)), then workgroup j < nq turns query j's sample keys
// into tau[j] (the m-th smallest), zeroes cnt[j] / flags[j], writes the exact ||q_j||^2 (cosine) and publishes
// ready[j] = epoch; the corpus tiles of the same launch pick the thresholds up in front of their epilogues.
// ticket: a counter that only grows (every sample workgroup adds 1); ticket_base = its value before this launch.
struct FusedSample {
    const uint32_t *smap = nullptr;
    uint32_t count = 0;
    int m = 0;
    uint32_t n_blocks = 0;
    uint32_t *ticket = nullptr;
    uint32_t ticket_base = 0;
    uint32_t *ready = nullptr;
    uint32_t epoch = 0;
    float *qna = nullptr; // or null (not cosine)
    int order = 0;
    uint32_t *fail_host = nullptr; // pinned: set to `epoch` when a wait gave up (the host then redoes the batch exactly)
    int relaxed = 0;               // diagnostic build only: hand-off without release / acquire (A/B of their cost)
};
void launch_gemm_filter_narrow_fused(int metric, const float *X, const float *norm2, const float *rnorm, int64_t row_begin,
                                     int64_t row_end, int D, const float *Q, int nq, const uint8_t *mask,
                                     const uint32_t *rowmap, CandState cs, hipStream_t s, bool tile64, FusedSample fs);
uint32_t fused_sample_blocks(uint32_t count, int nq, bool tile64); // sample workgroups such a launch starts with
// f32 [rows][D] -> split-bf16 image (same byte shape; D % 32 == 0) consumed by the split GEMM
void launch_split_bf16(const float *src, float *dst, int64_t rows, int D, hipStream_t s);
// split-bf16 contraction on 256 x 256 tiles with whole 128-B lines per row and K-step (kernels_gemm_tall2.hip); Qs = split image
// of the batch (launch_split_bf16); asplit 1: X is the split image of the corpus, 2: X is the f32 corpus (split in registers)
void launch_gemm_filter_tall2(int metric, const float *X, const float *norm2, const float *rnorm, int64_t row_begin,
                              int64_t row_end, int D, const float *Qs, int nq, const uint8_t *mask, const uint32_t *rowmap,
                              CandState cs, bool boot, int asplit, hipStream_t s);

// ONE fp16 product per (row, query, k) on 256 x 256 tiles (kernels_gemm_tall16.hip): Qh / qinv from launch_queries_to_f16
// (fp16 image of the batch, each query scaled by a power of two to a norm in [1, 2); qinv = 1 / scale); X = f32 corpus
void launch_queries_to_f16(const float *Q, int nq, int D, void *Qh, float *qinv, hipStream_t s);
// The one-tile persistent kernel over the image (<= 128 queries) can turn the sample the launch before it left in
// lists[q][0 .. count) into the thresholds ITSELF (its first nq workgroups do, on shorter row ranges; everybody picks the
// thresholds up in front of its first epilogue): no threshold launch, no gap behind it.  The launch before must leave
// tau[q] = 0 ("not out yet": launch_query_prep / SamplePrep, tau_zero); a bounded wait that gives up stores `tag` to the
// pinned fail_host and admits nothing (the host redoes the batch).  Q / qna / order: the duty workgroups also compute the
// exact ||q||^2 (cosine; qna null otherwise).
struct Tall16Tin {
    uint32_t count;
    int m;
    uint32_t tag;
    uint32_t *fail_host;
    const float *Q;
    float *qna;
    int order;
};
bool tall16_tin_ok(int D, int nq, int64_t n_pos, bool img, bool mapped, bool masked, bool with_norm, uint32_t count, int m);
// the same image and scales, plus the exact ||q||^2 in `order` (qna, or null) and the reset of the queries' candidate state:
// one launch for what a search over this route needs from its batch (kernels_scan.hip)
void launch_query_prep(const float *Q, int nq, int D, void *Qh, float *qinv, float *qna, int order, CandState cs, hipStream_t s,
                       const float *center = nullptr, // center: the image of q - center (L2 over the centred corpus image)
                       float *qnrm = nullptr,         // [nq] upper bounds of |q| (the dot product's lower-bound key)
                       bool tau_zero = false,         // thresholds left at 0 = "not out yet" (a TAUIN candidate launch follows)
                       float *qrho = nullptr,         // [nq] |q - fp16 image of q| / |q|, padded: the query's share of the keys' error bound
                       float rho_gain = 0.f);         // qnrm[q] *= 1 + rho_gain * qrho[q] (dot product: folds that share into G)
// Xh (or null): the corpus's K-blocked fp16 image [Dp / 32][xh_cap][32] from launch_corpus_to_f16, in step with X; used by
// unfiltered searches (half the bytes to stage, a four-stage ring)
void launch_gemm_filter_tall16(int metric, const float *X, const float *norm2, const float *rnorm, int64_t row_begin,
                               int64_t row_end, int D, const void *Qh, const float *qinv, int nq, const uint8_t *mask,
                               const uint32_t *rowmap, CandState cs, bool boot, hipStream_t s, const void *Xh = nullptr,
                               int64_t xh_cap = 0,
                               uint32_t gstride = 0,  // boot launches of the persistent forms: positions = granules of 16 rows,
                                                      // gstride rows apart (an evenly spaced sample read in whole KiB)
                               const float *qnrm = nullptr, float gsum = 0.f, // dot product on the persistent forms: upper bounds
                                                      // of |q| (launch_query_prep) and gamma_a + gamma_o -- the lower-bound key
                               const struct Tall16Tin *tin = nullptr); // thresholds inside the launch (tall16_tin_ok)
void launch_corpus_to_f16(const float *X, int64_t row_begin, int64_t row_end, int D, void *Xh, int64_t cap, hipStream_t s,
                          const float *center = nullptr); // center (or null; [>= D + 8]): the image holds fp16(x - center)
// *stat = max(*stat, max over the rows of |x - fp16(x)|^2 / |x|^2) as float bits (x - center for the centred image): the
// measured loss of the image, from which the candidate keys' error bound is taken (index.hip: gamma)
void launch_f16_residual(const float *X, int64_t row_begin, int64_t row_end, int D, const float *center, uint32_t *stat, hipStream_t s);
int corpus_f16_plane_dims(); // dimensions per plane of that image (its rows are zero-padded to a multiple of it)
// under a row list (mapped) the persistent kernels gather out of the image and leave POSITIONS of the list in the candidate
// entries (the finish launch maps them back: posmap); true when a launch with these parameters does so
bool tall16_entries_are_positions(int D, int nq, bool img, bool mapped, bool masked);
bool tall16_runs_persistent(int D, int nq, bool img, bool mapped, bool masked); // the persistent kernels serve such a launch

// per query: sort the list, keep the best kc, tau = kc-th entry (or max), flag overflow.
// qsel (nullable): only these query slots.  boot_rows > 0: the list was filled by a bootstrap
// launch (one entry per row at index row - row_begin, no atomics; masked rows hold kEntryMax).
// tau_only: the list is a row *sample*; publish its kc-th entry as threshold and empty the list.
// need_at_least: flag bit 2 when fewer entries than this were admitted (sampled threshold too tight).
// emit: the search's last select also writes the k results per slot (what launch_emit_lists would do)
struct EmitArgs {
    int k;
    const int64_t *ids;
    float *out_dist;
    int64_t *out_labels;
    uint32_t *flags_host;
};
void launch_select(CandState cs, const int *qsel, int nsel, int kc, uint32_t boot_rows, hipStream_t s,
                   bool tau_only = false, uint32_t need_at_least = 0, const EmitArgs *emit = nullptr,
                   bool striped = false,
                   bool unsorted = false); // unsorted: the kept entries need not be ordered, only the worst one sits last
// approximate distances of `count` evenly spaced positions of [0, span) for up to 8 query slots, written
// as entries to lists[q][0..count) (also clears the slots' flags); cosine: `nsel` extra workgroups compute
// the slots' exact ||q||^2 into qna in the requested order.  norm2/rnorm given ("keys" mode, up to 64 slots):
// the entries carry the MFMA pipeline's candidate keys instead of distances
// prep (or null): the launch also does launch_query_prep's work for its queries (image, scales, |q| bounds; not the exact
// norms, not the state reset -- the threshold launch behind it sets the state): one launch and one gap less in front of a
// small search's candidate pass
struct SamplePrep {
    void *Qh;
    float *qinv, *qnrm;
    const float *center;
    bool tau_zero; // leave tau = 0 ("not out yet": a TAUIN candidate launch follows)
    float *qrho;   // as launch_query_prep
    float rho_gain;
};
void launch_sample_scores(int metric, int order, const float *X, int D, int64_t span, uint32_t count,
                          const uint32_t *rowmap, const uint8_t *mask, const float *Q, const int *qsel, int nsel,
                          CandState cs, float *qna, hipStream_t s, const float *norm2 = nullptr,
                          const float *rnorm = nullptr,
                          const float *center = nullptr,  // keys mode, L2: keys about this centre (norm2 = the centred norms)
                          const SamplePrep *prep = nullptr);
// tau[q] = m-th smallest of lists[q][0..count) with the row bits saturated, cnt[q] = 0
bool sample_tau_supported(uint32_t count, int m);
// zero_stripes: also reset the slots' striped admission counters (at most 8 slots: the scan path)
// qna != null: `nsel` extra workgroups compute the slots' exact ||q||^2 (order) alongside
void launch_sample_tau(CandState cs, const int *qsel, int nsel, uint32_t count, int m, bool zero_stripes, hipStream_t s,
                       const float *Q = nullptr, int D = 0, float *qna = nullptr, int order = 0);
// smap[i] = row behind the i-th of `count` evenly spaced positions of [0, span) (through rowmap if given)
void launch_sample_map(const uint32_t *rowmap, int64_t span, uint32_t count, uint32_t *smap, hipStream_t s);

// The last launch of a batched search (kernels_finish.hip): members of each query's candidate list within the key-space cut
// a_k + (1 + beta) E are re-ranked in the exact order, proven (or flagged: bit 1) and written.  posmap (or null): the lists
// carry positions of this row list.  done / xcnt: [nq] words that are zero between launches; xscratch: finish_scratch_bytes
// for the split form (up to nq_split_max queries); smax: members a query may have (power of two, >= 1024).
size_t finish_scratch_bytes(int nq_split_max, uint32_t smax);
void launch_finish(int metric, int order, const float *X, int D, const float *Q, int nq, const float *qna, CandState cs, int k,
                   const uint32_t *d_maxnorm2, float gamma, float beta, const int64_t *ids, const uint32_t *posmap, float *out_dist,
                   int64_t *out_labels, hipStream_t s, uint32_t *flags_host, uint32_t *done, uint32_t *xcnt, void *xscratch,
                   int nq_split_max, uint32_t smax,
                   const float *center = nullptr, // L2 keys taken about this centre: d_maxnorm2 = the centred maximum norm
                   const float *lb_norm2 = nullptr, const float *lb_qnrm = nullptr, float lb_gsum = 0.f, // dot: lower-bound keys
                   const float *qrho = nullptr, float qrho_k = 0.f); // gamma(q) = gamma + qrho_k * qrho[q] (launch_query_prep)

// ||q||^2 per selected query slot in the requested accumulation order (cosine).
void launch_query_norms(int order, const float *Q, const int *qsel, int nsel, int D, float *qna,
                        hipStream_t s);

// exact-order scan (nsel <= 8): distances of every row in [row_begin,row_end) for query slots
// qsel[0..nsel) (indices into Q, or identity when qsel==nullptr), admitted against tau.
// If all_out != nullptr writes every distance to all_out[slot*ld + row] instead (simd batch API).
void launch_scan(int metric, int order, bool raw_dot, const float *X, int64_t row_begin,
                 int64_t row_end, int D, const float *Q, const int *qsel, int nsel,
                 const float *qna, const uint8_t *mask, const uint32_t *rowmap, CandState cs, bool boot,
                 float *all_out, int64_t ld, hipStream_t s, bool striped = false);

// after the last select of the scan path: lists already hold exact distances.
void launch_emit_lists(CandState cs, const int *qsel, int nsel, int k, const int64_t *ids,
                       float *out_dist, int64_t *out_labels, uint32_t *flags_host, hipStream_t s);

void launch_init_cand(CandState cs, const int *qsel, int nsel, hipStream_t s);
// dist[0..n) = FLT_MAX, lab[0..n) = -1 (index.hip)
void launch_fill_empty(float *dist, int64_t *lab, int64_t n, hipStream_t s);

// shard s's [nq][k] blocks start at dist_in + s*dist_stride and lab_in + s*lab_stride (elements)
void launch_merge_topk(int nshards, int64_t nq, int k, const float *dist_in, const int64_t *lab_in,
                       int64_t dist_stride, int64_t lab_stride, float *dist_out, int64_t *lab_out, hipStream_t s);

void launch_fill_uniform(float *dst, int64_t n, uint64_t seed, int64_t offset, hipStream_t s);
void launch_fill_codes(uint8_t *dst, int64_t n, uint64_t seed, int64_t offset, hipStream_t s);
void launch_fill_uniform_rows(float *dst, const int64_t *ids, int64_t nrows, int dim, uint64_t seed, hipStream_t s);

// PQ
// minrng (nullable): [nq][M][4] = {min, max - min, 1.0 if the subtable holds a NaN / negative / infinite entry, 0}
void launch_build_adc_table(const float *codebooks, int M, int K, int sub, const float *Q, int nq,
                            float *tables, hipStream_t s, float *minrng = nullptr,
                            uint32_t *zero_a = nullptr, uint32_t *zero_b = nullptr); // zero_*: [nq] words cleared on the way
// one query per launch: `table` is that query's [M*256] table; entries go to cs slot `slot`,
// or (all_out != nullptr) every distance is written to all_out[row - out_base].
// skip_if_ok (nullable): {s_tau, ok} of the byte-table prefilter; the launch returns at once when ok != 0
// (the prefilter + exact-candidates kernels did the work)
void launch_adc_scan(const float *table, int M, const uint8_t *codes, int64_t row_begin, int64_t row_end,
                     int slot, const uint8_t *mask, CandState cs, bool boot, float *all_out,
                     int64_t out_base, hipStream_t s, const int *skip_if_ok = nullptr);

// sampled threshold for the ADC scan: exact ADC entries of `count` evenly spaced rows -> out[count];
// launch_sample_topm reduces them to groups*m entries at lists[slot] (returns groups, 0 = does not fit)
void launch_adc_sample(const float *table, int M, const uint8_t *codes, int64_t n, uint32_t count, uint64_t *out,
                       hipStream_t s);
uint32_t launch_sample_topm(const uint64_t *in, uint32_t count_total, int m, CandState cs, int slot, hipStream_t s);

// PQ codec + two-stage ADC search (kernels_pq2.hip)
void launch_pq_encode(const float *codebooks, int M, int K, int sub, const float *X, int64_t n, uint8_t *codes,
                      hipStream_t s);
void launch_pq_decode(const float *codebooks, int M, int K, int sub, const uint8_t *codes, int64_t n, float *out,
                      hipStream_t s);
// byte table + integer admission bound of one query: params = {s_tau, ok}; minrng from launch_build_adc_table
void launch_adc_quantise(const float *table, const float *minrng, int M, const uint64_t *tau, uint8_t *qtab, int *params,
                         hipStream_t s);
// rows whose lower bound can still pass the slot's threshold -> cand[0..*cand_cnt) (false: M too large for LDS)
bool launch_adc_prefilter(const uint8_t *qtab, const int *params, int M, const uint8_t *codes, int64_t n,
                          uint32_t *cand, uint32_t cand_cap, uint32_t *cand_cnt, hipStream_t s);
// two queries in ONE pass over the codes (false: no such form for this M -- run two single passes)
bool launch_adc_prefilter2(const uint8_t *qtab, const int *params, uint32_t *cand, uint32_t *cand_cnt, const uint8_t *qtab2,
                           const int *params2, uint32_t *cand2, uint32_t *cand_cnt2, int M, const uint8_t *codes, int64_t n,
                           uint32_t cand_cap, hipStream_t s);
// four queries in ONE pass (interleaved byte tables, one dword gather per code byte; false: no such form for this M)
bool launch_adc_prefilter4(const uint8_t *const qtab[4], const int *const params[4], uint32_t *const cand[4], uint32_t *const cand_cnt[4],
                           int M, const uint8_t *codes, int64_t n, uint32_t cand_cap, hipStream_t s);
void launch_adc_exact_candidates(const float *table, int M, const uint8_t *codes, const uint32_t *cand,
                                 const uint32_t *cand_cnt, uint32_t cand_cap, const int *params, int slot, CandState cs,
                                 hipStream_t s);
void launch_adc_rerank(const float *table, int M, const uint8_t *codes, int64_t n, const int64_t *rows, int64_t nrows,
                       float *out_dist, float *out_score, hipStream_t s);

// predicate masks (kernels_filter.hip): op = simd.CompareOp value; validity = Arrow LSB bitmap or null
void launch_match_int64(const int64_t *src, int64_t n, int64_t val, int op, const uint8_t *validity,
                        int64_t valid_offset, uint8_t *dst, int combine, hipStream_t s);
void launch_match_float32(const float *src, int64_t n, float val, int op, const uint8_t *validity,
                          int64_t valid_offset, uint8_t *dst, int combine, hipStream_t s);
void launch_and_bytes(uint8_t *dst, const uint8_t *src, int64_t n, hipStream_t s);
// ordered compaction of a byte mask into the ascending list of visible rows; scratch holds
// compact_scratch_words(n) u32 and ends with the visible-row count at [words-1]
int64_t compact_scratch_words(int64_t n);
void launch_compact_mask(const uint8_t *mask, int64_t n, uint32_t *rowmap, uint32_t *scratch, hipStream_t s);
// reciprocal-rank fusion of two ranked id lists per query (kernels_filter.hip)
void launch_rrf(int64_t nq, int kd, const int64_t *dense, int ks, const int64_t *sparse, int k, int limit,
                int64_t *out_ids, float *out_scores, hipStream_t s);

} // namespace lb
