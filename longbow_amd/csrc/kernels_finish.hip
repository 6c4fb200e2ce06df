// kernels_finish.hip -- the last launch of a batched search: prune the admitted candidates in KEY space, re-rank what is
// left in the reference's exact f32 order, prove the result, write it.
//
// It replaces "select the best kc keys -> re-rank all kc -> containment check" (two launches, kc = 256 / 512 exact rows per
// query whatever the data) by ONE launch that re-ranks only the rows that can still be among the k nearest:
//
//   1. the query's candidate list (every row whose key passed the admission threshold tau: ~2,300 entries at 1M rows) is
//      read into registers; a radix select finds its k-th smallest key a_k;
//   2. cut = a_k + (1 + beta) E, E = the rigorous error bound of a candidate key against the exact value (the contraction's
//      gamma of index.hip plus the exact sums' own rounding); S = { entries with key <= cut } -- a few hundred rows when the
//      keys are coarse (fp16 products), k + a handful when they are fine (split-bf16, f32);
//   3. the rows of S are gathered and scored in the reference's order (internal/simd/simd_test.go:13-33,
//      simd.go:138-163,365-479; BruteForceIndex.SearchVectors' (distance, row) ranking, adaptive_index.go:200-222);
//   4. proof: a row y outside S has key(y) > cut -- either it is in the list beyond the cut, or it was never admitted, and
//      then key(y) >= tau_key > cut (checked: otherwise the query is flagged).  Its exact value therefore exceeds
//      T = f(cut) - E_out, with f the metric's map from key to exact value and E_out the one-sided bound.  If at least k
//      members of S have an exact value < T (strictly), the k nearest rows are all in S and the sorted prefix is the answer;
//      otherwise flag bit 1, and the host widens the list or takes the exact scan (index.hip: search_batch_device).
//      Correctness never depends on beta: it only decides how often the proof closes at the first attempt.
//
// Two forms.  SPLIT (a few queries: latency): G workgroups per query share the members (list position mod G), each scores
// 16 rows at a time with every 16-B piece of the rows in flight at once (one memory round trip for D <= 1024); results go
// to a per-query scratch block with device-scope stores and the last workgroup to arrive (ticket) sorts, proves and emits.
// Tiled (many queries: throughput): one workgroup per query, 256 members at a time staged through LDS in 128-B pieces.
#include "lb_device.h"
#include "lb_select.h"

#include <float.h>
#include <algorithm>

#pragma clang fp contract(off)

namespace lb {

namespace {

constexpr int FN_THREADS = 256;
constexpr int FN_R16 = 16, FN_SD = 1024, FN_LD16 = FN_SD + 4; // SPLIT: 16 rows x up to 1024 dims per stage
constexpr int FN_DK = 32, FN_LDT = FN_DK + 4;                 // tiled: 256 rows x 32 dims per stage (one 128-B line per row)

#ifdef LB_DIAG
__device__ unsigned long long g_finish_probe[8]; // [0] members summed over queries, [1] queries, [2] list entries summed, [3] largest member count
#endif

// four LDS-DMA requests of 1 KiB behind one M0 write: the instruction offset moves the LDS destination and the global address
// alike (the caller pre-compensates the sources), as in the candidate kernels
__device__ __forceinline__ void fn_dma16x4(const void *g0, const void *g1, const void *g2, const void *g3, uint32_t lds_addr)
{
    uint32_t save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %4, off offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void fn_wait_vmcnt()
{
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

struct FinishArgs {
    const float *X;
    int D;
    const float *Q;
    const float *qna; // cosine: exact ||q||^2 per query in the requested order
    CandState cs;
    int k;
    const uint32_t *maxnorm2;
    float gamma, beta;
    const float *qrho; // or null: per-query share of the keys' error bound, gamma(q) = gamma + qrho_k * qrho[q] (index.hip: measured residuals)
    float qrho_k;
    const int64_t *ids;
    const uint32_t *posmap; // or null: the lists carry positions of this row list (a filtered view), not corpus rows
    const float *center;    // or null (L2 only): the keys were taken about this centre -- key + |q - c|^2 ~ d^2, the proof's norms
                            // are the centred ones (maxnorm2 then points at the centred maximum); the exact sums are untouched
    // dot product with LOWER-BOUND keys (the persistent fp16 kernels, kernels_gemm_tall16.hip): key' = -(q.x)~ / G - |x|,
    // G = lb_gsum * lb_qnrm[q], so that -q.x >= key' G for every row, however long.  lb_norm2 (or null: plain keys): the rows' |x|^2
    const float *lb_norm2, *lb_qnrm;
    float lb_gsum;
    float *out_dist;
    int64_t *out_labels;
    uint32_t *flags_host;
    uint32_t smax;    // members a query may have (power of two)
    int aligned;      // D % 4 == 0 and 16-B aligned rows / queries
    int split_ring;   // SPLIT form with the tiled form's row ring as its scoring scheme (17 .. 128 queries: a few workgroups per
                      // query so that every CU gathers; the 16-row scheme is the latency form for up to 16 queries)
    int nst;          // tiled form: stages of the row ring (2 .. 4, what the LDS beside the member arrays allows); 0 = the
                      // register-staged tile (several workgroups per CU)
    // SPLIT form
    uint32_t *done;   // [nq] arrival tickets (zero between launches)
    uint32_t *xcnt;   // [nq] members written to the scratch block (zero between launches)
    uint64_t *xent;   // [nq][smax] exact entries
    float *xcmp;      // [nq][smax] compare values
    int abl;          // diagnostic build: timing-only ablations (1 = no gather / exact sums, 2 = no radix select, 3 = return at once)
};

// exactly `need` (>= 1) of the real entries held in registers are <= the returned pivot (entries are unique).
// hist[256], wsum[4], scal[8], red[2]: LDS scratch.  All threads of the workgroup call it.
template <int PER>
__device__ __forceinline__ uint64_t radix_kth_regs(const uint64_t (&e)[PER], uint32_t need, uint32_t *hist, uint32_t *wsum,
                                                   uint32_t *scal, unsigned long long *red, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { red[0] = 0ull; red[1] = ~0ull; }
    __syncthreads();
    {
        uint64_t o = 0, an = ~0ull;
#pragma unroll
        for (int j = 0; j < PER; j++)
            if (e[j] != kEntryMax) { o |= e[j]; an &= e[j]; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            o |= __shfl_xor(o, off);
            an &= __shfl_xor(an, off);
        }
        if (lane == 0) {
            atomicOr(&red[0], (unsigned long long)o);
            atomicAnd(&red[1], (unsigned long long)an);
        }
    }
    __syncthreads();
    const uint64_t diff = red[0] ^ red[1];
    const uint64_t common = red[1];
    __syncthreads();
    if (diff == 0ull) return common | 0xffffffffull; // one real entry (or none): it is the pivot
    int first_shift = 56;
    uint64_t prefix = 0, mask = 0;
    {
        const int same_bytes = __builtin_clzll(diff) >> 3;
        first_shift = 56 - 8 * same_bytes;
        if (same_bytes > 0) {
            mask = ~0ull << (64 - 8 * same_bytes);
            prefix = common & mask;
        }
    }
    for (int shift = first_shift; shift >= 0; shift -= 8) {
        hist[tid] = 0; // (FN_THREADS == 256 bins)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; j++)
            if (e[j] != kEntryMax && (e[j] & mask) == prefix) atomicAdd(&hist[(uint32_t)(e[j] >> shift) & 0xffu], 1u);
        __syncthreads();
        const uint32_t h = hist[tid];
        uint32_t incl = wave_incl_scan(h, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) base += (w < wave) ? wsum[w] : 0u;
        incl += base;
        const uint32_t excl = incl - h;
        if (excl < need && need <= incl) {
            scal[0] = (uint32_t)tid;
            scal[1] = need - excl;
            scal[4] = (need == incl) ? 1u : 0u;
        }
        __syncthreads();
        prefix |= (uint64_t)scal[0] << shift;
        mask |= 0xffull << shift;
        need = scal[1];
        const bool whole_bucket = scal[4] != 0u;
        __syncthreads();
        if (whole_bucket) {
            prefix |= ~mask;
            break;
        }
    }
    return prefix;
}

template <int METRIC>
__device__ __forceinline__ void exact_values(float t, float nbt, float na, int D, float &dist, float &cmp)
{
    if (METRIC == METRIC_L2) {
        dist = (float)sqrt((double)t);
        cmp = t; // compare in d^2 space
    } else if (METRIC == METRIC_COS) {
        if (D == 0 || na == 0.0f || nbt == 0.0f) dist = 1.0f;
        else dist = 1.0f - __fdiv_rn(t, (float)sqrt((double)na * (double)nbt));
        cmp = dist;
    } else {
        dist = -t;
        cmp = dist;
    }
}

// one lane, one row, straight from memory in the reference's order (dimensions that are not multiples of 4, unaligned rows)
template <int METRIC, int ORDER>
__device__ __forceinline__ void exact_row_generic(const float *x, const float *q, int D, float na, float &dist, float &cmp)
{
    AccR<ORDER> acc, nb;
    acc.zero();
    nb.zero();
    const int dmain = D & ~3;
    for (int i = 0; i < dmain; i += 4) {
        const float x0 = x[i], x1 = x[i + 1], x2 = x[i + 2], x3 = x[i + 3];
        const float q0 = q[i], q1 = q[i + 1], q2 = q[i + 2], q3 = q[i + 3];
        if (METRIC == METRIC_COS) {
            nb.template add<0>(x0 * x0);
            nb.template add<1>(x1 * x1);
            nb.template add<2>(x2 * x2);
            nb.template add<3>(x3 * x3);
        }
        if (METRIC == METRIC_L2) {
            const float e0 = q0 - x0, e1 = q1 - x1, e2 = q2 - x2, e3 = q3 - x3;
            acc.template add<0>(e0 * e0);
            acc.template add<1>(e1 * e1);
            acc.template add<2>(e2 * e2);
            acc.template add<3>(e3 * e3);
        } else {
            acc.template add<0>(q0 * x0);
            acc.template add<1>(q1 * x1);
            acc.template add<2>(q2 * x2);
            acc.template add<3>(q3 * x3);
        }
    }
    for (int i = dmain; i < D; i++) {
        const float xv = x[i], qv = q[i];
        if (METRIC == METRIC_COS) nb.add_tail(xv * xv);
        if (METRIC == METRIC_L2) {
            const float e = qv - xv;
            acc.add_tail(e * e);
        } else {
            acc.add_tail(qv * xv);
        }
    }
    exact_values<METRIC>(acc.total(), nb.total(), na, D, dist, cmp);
}

// What separates the exact (computed) value of a row from the value its candidate key stands for, one-sided: the proof
// subtracts it from f(cut).  cutf: a key; dk (L2): the k-th member's distance (rows beyond |q| + d_k cannot enter the result).
// L2: in d^2 units; cosine: in distance units; dot: in key units.
template <int METRIC>
__device__ __forceinline__ float out_slack(float cutf, float dk, float nq2, float nqn, float xmax, float ga, float go)
{
    if (METRIC == METRIC_L2) {
        float xe = xmax;
        if (dk >= 0.0f && dk < FLT_MAX) xe = fminf(xmax, (nqn * 1.001f + dk * (1.001f + 2.0f * go)) * 1.001f);
        // (1.25: the f32 roundings of the key's fma and of cut + |q|^2, each below 2^-24 (xe^2 + 2 |q| xe))
        return fabsf(cutf + nq2) * 3.0f * go + 3.0f * go * nq2 + 1.25f * (go * xe * xe + 2.0f * ga * nqn * xe);
    }
    if (METRIC == METRIC_COS) { // (|q| from the f32 sum na: relative go / 2 on cut / |q|)
        const float cq = nqn > 0.f ? fabsf(cutf) / nqn : 0.f;
        return 1.1f * (ga + 2.8f * go) + go * fminf(cq, 2.0f) + 4e-6f;
    }
    return 1.1f * (ga + go) * nqn * xmax + 2e-6f * fabsf(cutf);
}

template <int METRIC, int ORDER, bool SPLIT, int PER>
__global__ __launch_bounds__(FN_THREADS) void finish_kernel(FinishArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t scal[8];
    __shared__ unsigned long long red[2];
    __shared__ float fred[4];
    __shared__ int s_last;

    const int qi = SPLIT ? (int)blockIdx.y : (int)blockIdx.x;
    const uint32_t g = SPLIT ? blockIdx.x : 0u, G = SPLIT ? gridDim.x : 1u; // (G divides 256)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, k = a.k;
    const uint32_t smax = a.smax;

    uint32_t *s_rows = reinterpret_cast<uint32_t *>(smem);     // [smax] rows of this workgroup's members
    uint64_t *skey = reinterpret_cast<uint64_t *>(s_rows + smax); // [smax] exact entries
    float *scmp = reinterpret_cast<float *>(skey + smax);      // [smax] compare values
    float *work = scmp + smax;                                 // the scoring scheme's tile

#ifdef LB_DIAG
    if (a.abl == 3) return;
#endif
    const uint32_t raw = a.cs.cnt[qi];
    const uint32_t n = raw < a.cs.cap ? raw : a.cs.cap;
    const uint64_t tau = a.cs.tau[qi];
    const uint64_t *list = a.cs.lists + (size_t)qi * a.cs.cap;
    uint32_t flags = raw > a.cs.cap ? 1u : 0u;

    auto emit = [&](const uint64_t *sorted, uint32_t nsorted, uint32_t fl) { // the sorted prefix -> the query's k results
        for (int r = tid; r < k; r += FN_THREADS) {
            float d = FLT_MAX;
            int64_t lab = -1;
            if ((uint32_t)r < nsorted) {
                const uint64_t en = sorted[r];
                d = entry_key(en);
                const uint32_t row = entry_row(en);
                lab = a.ids ? a.ids[row] : (int64_t)row;
            }
            a.out_dist[(int64_t)qi * k + r] = d;
            a.out_labels[(int64_t)qi * k + r] = lab;
        }
        if (tid == 0) {
            const uint32_t all = fl ? (atomicOr(&a.cs.flags[qi], fl) | fl) : atomicOr(&a.cs.flags[qi], 0u);
            if (a.flags_host) a.flags_host[qi] = all;
        }
    };
    if (n == 0) { // nothing was admitted: an empty view -- or a threshold that was too tight (the query is redone)
        if (g == 0) emit(skey, 0u, tau != kEntryMax ? 4u : 0u);
        return;
    }

    // ---- the list, in registers; |q| ---------------------------------------------------------------------------------
    uint64_t e[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint32_t idx = (uint32_t)tid + (uint32_t)FN_THREADS * j;
        e[j] = idx < n ? list[idx] : kEntryMax;
    }
    const float *q = a.Q + (int64_t)qi * D;
    float nq2 = 0.f;
    if (METRIC != METRIC_COS) { // (any order: only ever used as a bound, with slack)
        for (int i = tid; i < D; i += FN_THREADS) {
            const float v = (METRIC == METRIC_L2 && a.center) ? q[i] - a.center[i] : q[i];
            nq2 += v * v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nq2 += __shfl_xor(nq2, off);
        if (lane == 0) fred[wave] = nq2;
    }
    if (tid == 0) { scal[2] = 0; scal[3] = 0; }
    __syncthreads();
    const float na = METRIC == METRIC_COS ? a.qna[qi] : 0.f;
    if (METRIC != METRIC_COS) nq2 = (fred[0] + fred[1]) + (fred[2] + fred[3]);
    else nq2 = na;
    const float go = 1.05f * (float)(D + 8) * 5.9604645e-8f; // (D + 8) 2^-24: the rounding of an exact f32 sum of D terms
    const float nqn = sqrtf(nq2) * (1.000002f + go);         // >= the real |q| (nq2 is an f32 sum in some order)

    const uint32_t kk = n < (uint32_t)k ? n : (uint32_t)k;
    const bool dot_lb = METRIC == METRIC_DOT && a.lb_norm2 != nullptr;
    // (lower-bound dot keys: the cut comes from the k-th smallest UPPER bound below, a_k is not used -- one selection less)
#ifdef LB_DIAG
    const uint64_t pivot = (a.abl == 2 || dot_lb) ? (e[0] | 0xffffffffull) : radix_kth_regs<PER>(e, kk, hist, wsum, scal, red, tid);
#else
    const uint64_t pivot = dot_lb ? (e[0] | 0xffffffffull) : radix_kth_regs<PER>(e, kk, hist, wsum, scal, red, tid);
#endif

    // ---- the cut ------------------------------------------------------------------------------------------------------
    const float ga = a.gamma + (a.qrho ? a.qrho_k * a.qrho[qi] : 0.f);
    const float xmax = sqrtf(__builtin_bit_cast(float, a.maxnorm2[0])) * 1.000001f;
    const float ak = entry_key(pivot);
    // what the proof at the end takes off f(cut), evaluated at a_k: the cut lies (1 + beta) of it beyond a_k -- one part for the
    // proof's one-sided bound, beta of it for the k best members' own key errors (rigorous bound E, typical error far below)
    const float E = out_slack<METRIC>(ak, METRIC == METRIC_L2 ? sqrtf(fmaxf(ak + nq2, 0.f)) : 0.f, nq2, nqn, xmax, ga, go) *
                    (METRIC == METRIC_COS ? nqn : 1.0f); // (cosine: the slack is in distance units, keys are |q| times that)
    float cutk = ak + (1.0f + a.beta) * E;
    if (!(cutk >= ak)) cutk = ak; // (NaN / overflow: the proof below decides)
    float lbG = 0.f;
    if (dot_lb) {
        // Lower-bound keys: -q.x >= key' G for every row, and -q.x <= (key' + 2 |x|) G.  So the k-th smallest of the UPPER
        // bounds U = key' + 2 |x| over the list bounds the k-th best exact value from above, and every row that can still be
        // among the k best has key' <= that: the cut.  A very long row has a very low key' (it might be anything) and a very
        // high U: it becomes a member, is scored exactly, and widens nobody else's bound.
        lbG = a.lb_gsum * a.lb_qnrm[qi];
        const float pad = 1.0f + (a.lb_gsum > 0.f ? 6.0e-7f / a.lb_gsum : 0.f) + 1.0e-5f + 1.05f * (float)(D + 8) * 5.9604645e-8f;
        uint64_t u[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            u[j] = kEntryMax;
            if (e[j] != kEntryMax) {
                uint32_t row = entry_row(e[j]);
                if (a.posmap) row = a.posmap[row];
                const float up = entry_key(e[j]) + 2.0f * sqrtf(a.lb_norm2[row]) * pad * pad;
                u[j] = pack_entry(up, (uint32_t)tid + (uint32_t)FN_THREADS * j);
            }
        }
        const float uk = entry_key(radix_kth_regs<PER>(u, kk, hist, wsum, scal, red, tid));
        cutk = uk + fabsf(uk) * 1.0e-5f + 1.0e-30f;
        if (!(cutk >= uk)) cutk = FLT_MAX;
        if (tid == 0) { scal[2] = 0; scal[3] = 0; }
        __syncthreads();
    }
    if (kk < (uint32_t)k) cutk = FLT_MAX; // fewer entries than results wanted: all of them
    uint32_t cut_s = f32_sortable(cutk + 0.0f);
    const uint32_t tau_s = (uint32_t)(tau >> 32);
    if (tau != kEntryMax && !(cut_s < tau_s)) { // rows beyond the admission threshold could lie below the cut: not provable
        flags |= 2u;
        cut_s = tau_s > 0 ? tau_s - 1u : 0u;
    }

    // ---- this workgroup's members (SPLIT: list positions = g mod G, i.e. threads = g mod G) ---------------------------
    const bool mine = !SPLIT || ((uint32_t)tid % G) == g;
    uint32_t all_members = 0; // (counted by everybody: the list is complete iff every entry is a member)
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const bool mem = e[j] != kEntryMax && (uint32_t)(e[j] >> 32) <= cut_s;
        all_members += mem ? 1u : 0u;
        if (mem && mine) {
            const uint32_t slot = atomicAdd(&scal[2], 1u);
            if (slot < smax) s_rows[slot] = entry_row(e[j]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) all_members += __shfl_xor(all_members, off);
    if (lane == 0 && all_members) atomicAdd(&scal[3], all_members);
    __syncthreads();
    const uint32_t nm_raw = scal[2];
    const uint32_t nm = nm_raw < smax ? nm_raw : smax;
    const uint32_t ns_all = scal[3];
    // every row there is was admitted and is a member: nothing lies outside S
    const bool complete = tau == kEntryMax && ns_all == n && raw <= a.cs.cap && ns_all <= smax;
#ifdef LB_DIAG
    if (tid == 0 && g == 0) { // (tools/bench_filtered.py: how many rows a query's finish re-ranks)
        atomicAdd(&g_finish_probe[0], (unsigned long long)ns_all);
        atomicAdd(&g_finish_probe[1], 1ull);
        atomicAdd(&g_finish_probe[2], (unsigned long long)n);
        atomicMax(&g_finish_probe[3], (unsigned long long)ns_all);
    }
#endif
    __syncthreads();
    if (a.posmap) // a filtered view: positions -> corpus rows (ascending, so the (key, position) order is the (key, row) order)
        for (uint32_t c = tid; c < nm; c += FN_THREADS) s_rows[c] = a.posmap[s_rows[c]];
    if (a.posmap) __syncthreads();

    // ---- exact values of the members: skey[c], scmp[c], c < nm --------------------------------------------------------
#ifdef LB_DIAG
    if (a.abl == 1) {
        for (uint32_t c = tid; c < nm; c += FN_THREADS) {
            skey[c] = pack_entry(0.f, s_rows[c]);
            scmp[c] = 0.f;
        }
    } else
#endif
    if (!a.aligned) {
        for (uint32_t c = tid; c < nm; c += FN_THREADS) {
            float dist, cmp;
            exact_row_generic<METRIC, ORDER>(a.X + (int64_t)s_rows[c] * D, q, D, na, dist, cmp);
            skey[c] = pack_entry(dist, s_rows[c]);
            scmp[c] = cmp;
        }
    } else if (SPLIT && !a.split_ring) {
        // 16 rows at a time, whole (up to 1024 dims per stage): 256 lanes fetch, then 16 (SEQ) / 64 (UNROLL4: four lanes per
        // row, one per accumulator chain) walk the rows out of LDS in the reference's order
        float *tile = work;                  // [16][FN_LD16]
        float *qs = tile + FN_R16 * FN_LD16; // [FN_SD]
        constexpr bool QUAD = ORDER == ORDER_UNROLL4;
        const int myr = QUAD ? (tid >> 2) : tid, myt = QUAD ? (tid & 3) : 0;
        const bool worker = tid < (QUAD ? 4 * FN_R16 : FN_R16);
        for (uint32_t c0 = 0; c0 < nm; c0 += FN_R16) {
            uint32_t rows[FN_R16];
#pragma unroll
            for (int r = 0; r < FN_R16; r++) rows[r] = s_rows[c0 + r < nm ? c0 + r : c0];
            f32x4 stg[FN_R16], stq;
            auto load_stage = [&](int d0) {
                const int kx = d0 + tid * 4;
                if (kx < D) {
                    stq = *reinterpret_cast<const f32x4 *>(q + kx);
#pragma unroll
                    for (int r = 0; r < FN_R16; r++) stg[r] = *reinterpret_cast<const f32x4 *>(a.X + (int64_t)rows[r] * D + kx);
                }
            };
            float a0 = 0.f, b0 = 0.f;
            load_stage(0);
            for (int d0 = 0; d0 < D; d0 += FN_SD) {
                if (d0 + tid * 4 < D) {
                    *reinterpret_cast<f32x4 *>(&qs[tid * 4]) = stq;
#pragma unroll
                    for (int r = 0; r < FN_R16; r++) *reinterpret_cast<f32x4 *>(&tile[r * FN_LD16 + tid * 4]) = stg[r];
                }
                __syncthreads();
                if (d0 + FN_SD < D) load_stage(d0 + FN_SD);
                if (worker) {
                    // the chains are serial (one addition per element, in order), so what can be hidden is the LDS latency:
                    // 32 elements' reads are issued together, then their 32 chain steps run (one read + wait per group of
                    // four cost 17 us per 768-float row instead of 3)
                    const int nel = min(D, d0 + FN_SD) - d0;
                    const float *xr = &tile[myr * FN_LD16];
                    if (QUAD) {
                        const int mine_n = (nel - myt + 3) >> 2; // elements myt, myt + 4, ...
                        for (int j0 = 0; j0 < mine_n; j0 += 16) {
                            float xv[16], qv[16];
#pragma unroll
                            for (int u = 0; u < 16; u++) {
                                const int el = myt + 4 * (j0 + u < mine_n ? j0 + u : mine_n - 1);
                                xv[u] = xr[el];
                                qv[u] = qs[el];
                            }
#pragma unroll
                            for (int u = 0; u < 16; u++) {
                                if (j0 + u < mine_n) {
                                    if (METRIC == METRIC_COS) b0 = b0 + xv[u] * xv[u];
                                    if (METRIC == METRIC_L2) {
                                        const float d = qv[u] - xv[u];
                                        a0 = a0 + d * d;
                                    } else {
                                        a0 = a0 + qv[u] * xv[u];
                                    }
                                }
                            }
                        }
                    } else {
                        const int ng = nel >> 2;
                        for (int g0 = 0; g0 < ng; g0 += 8) {
                            f32x4 xv[8], qv[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) {
                                const int gq = g0 + u < ng ? g0 + u : ng - 1;
                                xv[u] = *reinterpret_cast<const f32x4 *>(&xr[gq * 4]);
                                qv[u] = *reinterpret_cast<const f32x4 *>(&qs[gq * 4]);
                            }
#pragma unroll
                            for (int u = 0; u < 8; u++) {
                                if (g0 + u < ng) {
                                    const float xe4[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, qe4[4] = {qv[u].x, qv[u].y, qv[u].z, qv[u].w};
#pragma unroll
                                    for (int c4 = 0; c4 < 4; c4++) {
                                        if (METRIC == METRIC_COS) b0 = b0 + xe4[c4] * xe4[c4];
                                        if (METRIC == METRIC_L2) {
                                            const float d = qe4[c4] - xe4[c4];
                                            a0 = a0 + d * d;
                                        } else {
                                            a0 = a0 + qe4[c4] * xe4[c4];
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                __syncthreads();
            }
            float t = a0, nbt = b0;
            if (QUAD) { // ((s0 + s1) + s2) + s3, as the reference's unrolled loops finish
                const int base = tid & ~3;
                const float a1 = __shfl(a0, base + 1), a2 = __shfl(a0, base + 2), a3 = __shfl(a0, base + 3);
                const float b1 = __shfl(b0, base + 1), b2 = __shfl(b0, base + 2), b3 = __shfl(b0, base + 3);
                t = a0 + a1;
                t = t + a2;
                t = t + a3;
                nbt = b0 + b1;
                nbt = nbt + b2;
                nbt = nbt + b3;
            }
            const uint32_t c = c0 + (uint32_t)myr;
            if (worker && myt == 0 && c < nm) {
                float dist, cmp;
                exact_values<METRIC>(t, nbt, na, D, dist, cmp);
                skey[c] = pack_entry(dist, s_rows[c]);
                scmp[c] = cmp;
            }
        }
    } else if (a.nst == 0) {
        // more than one workgroup per CU (beyond 256 queries): 256 members at a time, one per lane, their rows staged through
        // registers into a padded LDS tile, 32 dims per stage (the next stage's loads in flight under the current stage's
        // chains) -- 56 KB of LDS, two workgroups to a CU, whose gathers and selections overlap (1024 queries: 152 us against
        // 177 with the ring below, which takes a CU's LDS for one workgroup)
        float *tile = work;                    // [256][FN_LDT]
        float *sq = tile + FN_THREADS * FN_LDT; // [Dpad]
        const int Dpad = (D + 31) & ~31;
        for (int i = tid; i < Dpad; i += FN_THREADS) sq[i] = i < D ? q[i] : 0.f;
        const int nchunks = (D + FN_DK - 1) / FN_DK;
        for (uint32_t g0 = 0; g0 < nm; g0 += FN_THREADS) {
            __syncthreads(); // the tile is free (previous group done); sq is there
            AccR<ORDER> acc, nb;
            acc.zero();
            nb.zero();
            f32x4 stg[8];
            auto load_stage = [&](int ch) {
                const int d0 = ch * FN_DK;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int idx = tid + FN_THREADS * i;
                    const int r = idx >> 3, p = idx & 7;
                    int kx = d0 + p * 4;
                    if (kx > D - 4) kx = D - 4; // pieces past D are never consumed
                    const uint32_t c = g0 + (uint32_t)r;
                    stg[i] = *reinterpret_cast<const f32x4 *>(a.X + (int64_t)s_rows[c < nm ? c : g0] * D + kx);
                }
            };
            load_stage(0);
            for (int ch = 0; ch < nchunks; ch++) {
                __syncthreads(); // tile free
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int idx = tid + FN_THREADS * i;
                    *reinterpret_cast<f32x4 *>(&tile[(idx >> 3) * FN_LDT + (idx & 7) * 4]) = stg[i];
                }
                __syncthreads();
                if (ch + 1 < nchunks) load_stage(ch + 1);
                const int d0 = ch * FN_DK;
                const int n4 = (min(D, d0 + FN_DK) - d0) >> 2;
                const float *xr = &tile[tid * FN_LDT];
#pragma unroll
                for (int gq = 0; gq < FN_DK / 4; gq++) {
                    if (gq < n4) {
                        const f32x4 xv = *reinterpret_cast<const f32x4 *>(&xr[gq * 4]);
                        const f32x4 qv = *reinterpret_cast<const f32x4 *>(&sq[d0 + gq * 4]);
                        if (METRIC == METRIC_COS) {
                            nb.template add<0>(xv.x * xv.x);
                            nb.template add<1>(xv.y * xv.y);
                            nb.template add<2>(xv.z * xv.z);
                            nb.template add<3>(xv.w * xv.w);
                        }
                        if (METRIC == METRIC_L2) {
                            const float e0 = qv.x - xv.x, e1 = qv.y - xv.y, e2 = qv.z - xv.z, e3 = qv.w - xv.w;
                            acc.template add<0>(e0 * e0);
                            acc.template add<1>(e1 * e1);
                            acc.template add<2>(e2 * e2);
                            acc.template add<3>(e3 * e3);
                        } else {
                            acc.template add<0>(qv.x * xv.x);
                            acc.template add<1>(qv.y * xv.y);
                            acc.template add<2>(qv.z * xv.z);
                            acc.template add<3>(qv.w * xv.w);
                        }
                    }
                }
            }
            const uint32_t c = g0 + (uint32_t)tid;
            if (c < nm) {
                float dist, cmp;
                exact_values<METRIC>(acc.total(), nb.total(), na, D, dist, cmp);
                skey[c] = pack_entry(dist, s_rows[c]);
                scmp[c] = cmp;
            }
        }
    } else if (nm > 0) {
        // 256 members at a time, one per lane.  Their rows come in by LDS-DMA, 32 dims (one 128-B line per row) per stage, into a
        // ring of nst stages of [256 rows][128 B]: nst - 1 stages are in flight while one is walked, and the ring runs on across
        // the groups of 256 (the stages of all groups are one stream).  The gather is bound by memory latency (one workgroup per
        // query, a CU to itself up to 256 queries), and loads staged through registers were no way to deepen it: the compiler
        // waits for vmcnt(0) in front of the LDS writes at the loop header whatever the code says (measured: two register
        // stages in flight cost what one did).  A DMA request is not tracked by the compiler; the waits here are counted.
        // Layout: the eight 16-B pieces of row r's line sit at position p ^ (r & 7) -- a lane asks for the piece that belongs
        // where its request lands, and the walk (lane t = row t, piece after piece) reads conflict-free.
        // (Two members per lane for 257 .. 512 members -- half the stages, the ring as two stages of 512 rows -- measured slower:
        // 122 against 113 us on the config-5 share, 270 members per query.)
        // None of this changes a row's sum: every chain runs over its row's elements in the reference's order.
        const int Dpad = (D + 31) & ~31;
        float *sq = work;                                                  // [Dpad]
        unsigned char *ring = reinterpret_cast<unsigned char *>(sq + Dpad); // [nst][256][128 B]
        const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ring;
        const int nst = a.nst, dist = nst - 1;
        for (int i = tid; i < Dpad; i += FN_THREADS) sq[i] = i < D ? q[i] : 0.f;
        const int nchunks = (D + FN_DK - 1) / FN_DK;
        const int ngroups = (int)((nm + FN_THREADS - 1) / FN_THREADS);
        const int total = ngroups * nchunks; // stages
        // request cursor: chunk rc of group rg into slot rslot.  A wave asks for 64 rows of a stage in eight requests of 8 rows x
        // 128 B (lane l: row l / 8 of the eight, position l % 8).
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int lrow = lane >> 3, piece = (lane & 7) ^ (lrow & 7); // (the rows of a request start at a multiple of 8)
        const unsigned char *rsrc[8];
        int rg = 0, rc = 0, rslot = 0;
        auto set_group = [&](int grp) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t c = (uint32_t)grp * FN_THREADS + (uint32_t)((wv * 8 + j) * 8 + lrow);
                const uint32_t row = s_rows[c < nm ? c : nm - 1]; // (lanes beyond the last member: its row again, never walked)
                // (the request's instruction offset moves the LDS destination AND the global address: compensated here)
                rsrc[j] = reinterpret_cast<const unsigned char *>(a.X + (int64_t)row * D) - (j & 3) * 1024;
            }
        };
        auto request = [&]() { // the cursor's stage (beyond the last one: the last one again, into a slot nobody reads)
            int kx = rc * FN_DK + piece * 4;
            if (kx > D - 4) kx = D - 4; // pieces past D are never consumed
            const uint32_t dst = ring_base + (uint32_t)rslot * (FN_THREADS * 128u) + (uint32_t)wv * 8192u;
            fn_dma16x4(rsrc[0] + kx * 4, rsrc[1] + kx * 4, rsrc[2] + kx * 4, rsrc[3] + kx * 4, dst);
            fn_dma16x4(rsrc[4] + kx * 4, rsrc[5] + kx * 4, rsrc[6] + kx * 4, rsrc[7] + kx * 4, dst + 4096u);
            rslot = rslot + 1 == nst ? 0 : rslot + 1;
            if (rc + 1 < nchunks) rc++;
            else if (rg + 1 < ngroups) { rg++; rc = 0; set_group(rg); }
        };
        __syncthreads(); // s_rows (posmap applied), sq
        set_group(0);
        for (int st = 0; st < dist; st++) request();
        AccR<ORDER> acc, nb;
        acc.zero();
        nb.zero();
        int cslot = 0, cc = 0, cg = 0;
        for (int sidx = 0; sidx < total; sidx++) {
            // stage sidx has landed for this wave: at most the dist - 1 younger stages (8 requests each) are out; behind the
            // barrier it has landed for all, and everybody is done with the stage before -- whose slot the next request takes
            if (dist >= 3) fn_wait_vmcnt<16>();
            else if (dist == 2) fn_wait_vmcnt<8>();
            else fn_wait_vmcnt<0>();
            __syncthreads();
            request();
            const int d0 = cc * FN_DK;
            const int n4 = (min(D, d0 + FN_DK) - d0) >> 2;
            const unsigned char *xr = ring + cslot * (FN_THREADS * 128) + tid * 128;
            auto walk = [&](int gq) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + ((gq ^ (tid & 7)) << 4));
                const f32x4 qv = *reinterpret_cast<const f32x4 *>(&sq[d0 + gq * 4]);
                if (METRIC == METRIC_COS) {
                    nb.template add<0>(xv.x * xv.x);
                    nb.template add<1>(xv.y * xv.y);
                    nb.template add<2>(xv.z * xv.z);
                    nb.template add<3>(xv.w * xv.w);
                }
                if (METRIC == METRIC_L2) {
                    const float e0 = qv.x - xv.x, e1 = qv.y - xv.y, e2 = qv.z - xv.z, e3 = qv.w - xv.w;
                    acc.template add<0>(e0 * e0);
                    acc.template add<1>(e1 * e1);
                    acc.template add<2>(e2 * e2);
                    acc.template add<3>(e3 * e3);
                } else {
                    acc.template add<0>(qv.x * xv.x);
                    acc.template add<1>(qv.y * xv.y);
                    acc.template add<2>(qv.z * xv.z);
                    acc.template add<3>(qv.w * xv.w);
                }
            };
            if (n4 == FN_DK / 4) { // a whole chunk: no guards, so that the eight pieces' LDS reads go out together
#pragma unroll
                for (int gq = 0; gq < FN_DK / 4; gq++) walk(gq);
            } else {
                for (int gq = 0; gq < n4; gq++) walk(gq);
            }
            cslot = cslot + 1 == nst ? 0 : cslot + 1;
            if (++cc == nchunks) { // the group's rows are complete
                const uint32_t c = (uint32_t)cg * FN_THREADS + (uint32_t)tid;
                if (c < nm) {
                    float dist_, cmp;
                    exact_values<METRIC>(acc.total(), nb.total(), na, D, dist_, cmp);
                    skey[c] = pack_entry(dist_, s_rows[c]);
                    scmp[c] = cmp;
                }
                acc.zero();
                nb.zero();
                cc = 0;
                cg++;
            }
        }
        fn_wait_vmcnt<0>(); // the requests beyond the last stage: nothing may land in LDS once the ring is given up
    }
    __syncthreads();

    // ---- SPLIT: hand the block in; the last workgroup of the query to arrive goes on ------------------------------------
    uint32_t ns = nm;
    if (SPLIT) {
        if (tid == 0) scal[5] = nm ? atomicAdd(&a.xcnt[qi], nm) : 0u;
        __syncthreads();
        const uint32_t base = scal[5];
        uint64_t *xe = a.xent + (size_t)qi * smax;
        float *xc = a.xcmp + (size_t)qi * smax;
        for (uint32_t c = tid; c < nm; c += FN_THREADS)
            if (base + c < smax) {
                __hip_atomic_store(&xe[base + c], skey[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&xc[base + c], scmp[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        // device-scope (write-through) stores, acknowledged before the ticket; the finisher reads them at device scope
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const uint32_t t = atomicAdd(&a.done[qi], 1u);
            s_last = (t == G - 1u) ? 1 : 0;
        }
        __syncthreads();
        if (!s_last) return;
        uint32_t total = 0;
        if (tid == 0) {
            total = __hip_atomic_load(&a.xcnt[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            scal[6] = total;
            a.done[qi] = 0; // ready for the next launch on this workspace
            a.xcnt[qi] = 0;
        }
        __syncthreads();
        total = scal[6];
        if (total > smax) {
            flags |= 2u;
            total = smax;
        }
        ns = total;
        for (uint32_t c = tid; c < ns; c += FN_THREADS) {
            skey[c] = __hip_atomic_load(&xe[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            scmp[c] = __hip_atomic_load(&xc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (nm_raw > smax) {
        flags |= 2u;
    }
    // order the members by (distance, row).  Up to 512: every thread ranks two entries against all (broadcast LDS reads, no
    // barrier per stage: 2 us where the 45 stages of a bitonic sort take 4.5) and scatters them; beyond: bitonic sort.
    const uint64_t *sorted = skey;
    if (tid == 0) scal[3] = 0;
    if (ns <= 512u) {
        uint64_t *srt = reinterpret_cast<uint64_t *>(s_rows); // (the row list is not needed any more; smax >= 1024: 4 KB)
        __syncthreads();
        const uint64_t e0 = (uint32_t)tid < ns ? skey[tid] : kEntryMax, e1 = (uint32_t)tid + 256u < ns ? skey[tid + 256] : kEntryMax;
        uint32_t r0 = 0, r1 = 0;
        for (uint32_t j = 0; j < ns; j++) {
            const uint64_t ej = skey[j];
            r0 += ej < e0 ? 1u : 0u;
            r1 += ej < e1 ? 1u : 0u;
        }
        if ((uint32_t)tid < ns) srt[r0] = e0; // (entries are unique: the ranks are a permutation)
        if ((uint32_t)tid + 256u < ns) srt[r1] = e1;
        sorted = srt;
        __syncthreads();
    } else {
        const uint32_t P = next_pow2(ns);
        for (uint32_t c = ns + tid; c < P; c += FN_THREADS) skey[c] = kEntryMax;
        __syncthreads();
        bitonic_sort_u64(skey, P, tid, FN_THREADS);
    }

    // ---- proof --------------------------------------------------------------------------------------------------------
    if (!complete) {
        const float cutf = sortable_f32(cut_s); // the cut that was applied (a key)
        // a row y outside S has key(y) > cut, so its exact value exceeds f(cut) - out_slack(cut):
        //   L2   d2c(y) >= (key(y) + |q|^2 - go |y|^2 - 2 ga |q||y|)(1 - go) for |y| <= xe; a row of a larger norm than
        //        xe = |q| + d_k is farther than the k-th member by the triangle inequality, whatever its key says
        //   cos  dist(y) = 1 - cos >= 1 + key(y) / |q| - (ga + 2.8 go)
        //   dot  dist(y) = -q.y >= key(y) - (ga + go) |q||y|
        float T, dk = 0.f;
        const bool skip = METRIC == METRIC_COS && na == 0.0f; // (every distance is exactly 1.0: selection by row is exact)
        if (METRIC == METRIC_L2) {
            const uint32_t kx = (ns < (uint32_t)k ? ns : (uint32_t)k);
            dk = kx > 0 ? entry_key(sorted[kx - 1]) : FLT_MAX;
        }
        const float slack = out_slack<METRIC>(cutf, dk, nq2, nqn, xmax, ga, go);
        if (METRIC == METRIC_L2) T = (cutf + nq2) - slack;
        else if (METRIC == METRIC_COS) T = 1.0f + (na > 0.f ? __fdiv_rn(cutf, sqrtf(na)) : 0.f) - slack;
        else if (dot_lb) T = cutf * lbG; // (-q.y >= key'(y) G > cut G; the roundings of the key are inside its padded |x|)
        else T = cutf - slack;
        T = T - fabsf(T) * 1e-6f;
        unsigned int local = 0;
        for (uint32_t c = tid; c < ns; c += FN_THREADS) local += (scmp[c] < T) ? 1u : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
        if (lane == 0 && local) atomicAdd(&scal[3], local);
        __syncthreads();
        if (!skip && scal[3] < (unsigned int)k) flags |= 2u;
    }
    emit(sorted, ns, flags);
}

} // namespace

#ifdef LB_DIAG
void read_finish_probe(unsigned long long out[8], bool reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_finish_probe), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_finish_probe), z, sizeof(z));
    }
}
#endif

size_t finish_scratch_bytes(int nq_split_max, uint32_t smax) { return (size_t)nq_split_max * smax * 12; }

// smax: members a query may have; the split form serves up to `nq_split_max` queries (scratch xent / xcmp sized for them)
void launch_finish(int metric, int order, const float *X, int D, const float *Q, int nq, const float *qna, CandState cs, int k,
                   const uint32_t *d_maxnorm2, float gamma, float beta, const int64_t *ids, const uint32_t *posmap, float *out_dist,
                   int64_t *out_labels, hipStream_t s, uint32_t *flags_host, uint32_t *done, uint32_t *xcnt, void *xscratch,
                   int nq_split_max, uint32_t smax, const float *center, const float *lb_norm2, const float *lb_qnrm, float lb_gsum,
                   const float *qrho, float qrho_k)
{
    if (nq <= 0) return;
    FinishArgs a;
    a.qrho = qrho;
    a.qrho_k = qrho_k;
    a.center = metric == METRIC_L2 ? center : nullptr;
    a.lb_norm2 = metric == METRIC_DOT ? lb_norm2 : nullptr;
    a.lb_qnrm = lb_qnrm;
    a.lb_gsum = lb_gsum;
    a.X = X; a.D = D; a.Q = Q; a.qna = qna; a.cs = cs; a.k = k; a.maxnorm2 = d_maxnorm2; a.gamma = gamma; a.beta = beta;
    a.ids = ids; a.posmap = posmap; a.out_dist = out_dist; a.out_labels = out_labels; a.flags_host = flags_host;
    a.smax = smax; a.done = done; a.xcnt = xcnt;
    a.abl = lb_tunable("LB_FINISH_ABL", 0);
    a.xent = reinterpret_cast<uint64_t *>(xscratch);
    a.xcmp = reinterpret_cast<float *>(a.xent + (size_t)nq_split_max * smax);
    a.aligned = (D % 4 == 0) && D >= 4 && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && ((reinterpret_cast<uintptr_t>(Q) & 15) == 0);
    const bool big = cs.cap > 8192u; // list entries per thread: 32 (cap 8192) or 64 (k > 512: cap 16384)
    const size_t common = (size_t)smax * 16;
    const size_t sh_split = common + ((size_t)FN_R16 * FN_LD16 + FN_SD) * 4;
    // tiled form: the query + a ring of up to four stages of 256 rows x 128 B beside the member arrays (the kernel's static
    // arrays take ~1.2 KB of the 160 KB); fewer than two stages fit only for very long queries with k > 512 -- those walk
    // their members straight from memory (the generic form)
    const size_t lds_max = 160 * 1024 - 2048, tiled_fixed = common + (size_t)((D + 31) & ~31) * 4;
    static const int nst_max = lb_tunable("LB_FINISH_NST", 4);
    int nst = tiled_fixed < lds_max ? (int)((lds_max - tiled_fixed) / ((size_t)FN_THREADS * 128)) : 0;
    if (nst > nst_max) nst = nst_max;
    if (nst > 4) nst = 4;
    // few queries: G workgroups per query (G divides 256)
    static const int g_force = lb_tunable("LB_FINISH_G", 0);
    int G = nq <= 4 ? 32 : (nq <= 16 ? 16 : 1);
    if (g_force > 0) G = g_force;
    const bool can_split = nq <= nq_split_max && done && xcnt && xscratch;
    if (!can_split) G = 1;
    // 17 .. 128 queries with many bytes to gather per query (k x D from 128 Ki: k = 200 at 1536 dimensions is 1.6 MB of rows):
    // one workgroup per query leaves CUs idle and each query's gather to one CU's share of the bandwidth -- 8 / 4 / 2
    // workgroups per query share its members (list position mod G), each through the row ring, and hand their exact values
    // to the last one to arrive as the split form does.  Config-5 share: finish 99 -> 67 us at 32 queries, 104 -> 74 at 128
    // (135 members a workgroup: one group of the ring instead of 256 + 14); at k = 100 x 768 dimensions (0.5 MB per query) it
    // is level, and stays one workgroup per query.
    static const int split_ring_on = lb_tunable("LB_FINISH_SPLIT_RING", 1);
    static const int split_ring_maxq = lb_tunable("LB_FINISH_SPLIT_RING_MAXQ", 128);
    const bool split_ring = G == 1 && split_ring_on && nq > 16 && nq <= split_ring_maxq && (int64_t)k * D >= 131072 && can_split &&
                            a.aligned && nst >= 2;
    if (split_ring) G = nq <= 32 ? 8 : (nq <= 64 ? 4 : 2);
    const bool split = G > 1;
    static const int ring_maxq = lb_tunable("LB_FINISH_RING_MAXQ", 256);
    const size_t sh_regtile = common + ((size_t)FN_THREADS * FN_LDT + (size_t)((D + 31) & ~31)) * 4;
    // beyond 256 queries several workgroups share a CU: the register-staged tile (56 KB) lets two of them overlap
    if (nq > ring_maxq && sh_regtile <= lds_max) nst = 0;
    else if (!split && nst < 2) a.aligned = 0;
    a.split_ring = split_ring ? 1 : 0;
    a.nst = nst;
    const size_t sh_tiled = !a.aligned ? common : (nst == 0 ? sh_regtile : tiled_fixed + (size_t)nst * FN_THREADS * 128);
    const size_t shmem = (split && !split_ring) ? sh_split : sh_tiled;
    dim3 grid = split ? dim3((unsigned)G, (unsigned)nq) : dim3((unsigned)nq);
#define LB_FN(M, O, S, P)                                                                                  \
    do {                                                                                                   \
        if (shmem > 64 * 1024)                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(finish_kernel<M, O, S, P>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);             \
        hipLaunchKernelGGL((finish_kernel<M, O, S, P>), grid, dim3(FN_THREADS), shmem, s, a);              \
    } while (0)
#define LB_FN_O(M, O)                              \
    do {                                           \
        if (split) {                               \
            if (big) LB_FN(M, O, true, 64);        \
            else LB_FN(M, O, true, 32);            \
        } else {                                   \
            if (big) LB_FN(M, O, false, 64);       \
            else LB_FN(M, O, false, 32);           \
        }                                          \
    } while (0)
#define LB_FN_M(M)                                                 \
    do {                                                           \
        if (order == ORDER_UNROLL4) LB_FN_O(M, ORDER_UNROLL4);     \
        else LB_FN_O(M, ORDER_SEQ);                                \
    } while (0)
    if (metric == METRIC_L2) LB_FN_M(METRIC_L2);
    else if (metric == METRIC_COS) LB_FN_M(METRIC_COS);
    else LB_FN_M(METRIC_DOT);
#undef LB_FN_M
#undef LB_FN_O
#undef LB_FN
}

} // namespace lb
