// kernels_select.hip -- batched top-k selection, exact re-rank and cross-shard merge.
//
// select:  per query, bitonic-sort the admitted (key,row) entries in LDS, keep the best kc,
//          publish the kc-th entry as the next admission threshold.  The ordering it realises
//          is the canonical form of BruteForceIndex.SearchVectors' bounded heap
//          (internal/store/adaptive_index.go:176-222): ascending (distance, row position).
// rerank:  recompute the kept candidates' distances in the reference's exact f32 order
//          (internal/simd/simd_test.go:13-33, simd.go:138-163,365-479), order by
//          (distance, row), verify that no row outside the candidate set can beat the k-th
//          result (rigorous rounding-error bound), write results.
// merge:   store.MergeSortedStreams (internal/store/result_merger.go:34-101) for S shards.
#include "lb_device.h"
#include "lb_select.h"

#include <float.h>
#include <algorithm>

#pragma clang fp contract(off)

namespace lb {

__global__ void init_cand_kernel(CandState cs, const int *qsel, int nsel)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nsel) {
        const int i = qsel ? qsel[j] : j;
        cs.cnt[i] = 0;
        cs.tau[i] = kEntryMax;
        cs.flags[i] = 0;
    }
}

void launch_init_cand(CandState cs, const int *qsel, int nsel, hipStream_t s)
{
    if (nsel <= 0) return;
    hipLaunchKernelGGL(init_cand_kernel, dim3((nsel + 255) / 256), dim3(256), 0, s, cs, qsel, nsel);
}

// Kernels that carve more than 64 KB of dynamic LDS must opt in once.
template <typename K>
static void allow_big_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// ---------------------------------------------------------------------------
// select: keep the kc smallest of n unique u64 entries, sorted.
//   n <= 2*kc : bitonic sort of everything.
//   otherwise : MSB radix select (8 passes of 8 bits over the LDS copy) finds the kc-th smallest
//               entry exactly (entries are unique: the row is part of the key), the <= pivot
//               entries are compacted and only those kc are sorted.
// LDS: entries u64[P] | hist u32[256] | wave sums u32[4] | scalars
// lane permutes on the DPP path (no LDS round trip): OR / AND of a u64 over the 64 lanes, result in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v)
{
    const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return ((uint64_t)(uint32_t)ohi << 32) | (uint64_t)(uint32_t)olo;
}
__device__ __forceinline__ void wave_or_and_u64(uint64_t &o, uint64_t &a)
{
#define LB_STEP(CTRL, RM)                 \
    o |= dpp_u64<CTRL, RM>(o);            \
    a &= dpp_u64<CTRL, RM>(a);
    LB_STEP(0xB1, 0xf)  // quad_perm [1,0,3,2]
    LB_STEP(0x4E, 0xf)  // quad_perm [2,3,0,1]
    LB_STEP(0x141, 0xf) // row_half_mirror
    LB_STEP(0x140, 0xf) // row_mirror
    LB_STEP(0x142, 0xa) // row_bcast15 -> rows 1,3 (disabled rows keep their own value: x|x, x&x)
    LB_STEP(0x143, 0xc) // row_bcast31 -> rows 2,3
#undef LB_STEP
}

template <int NT> // threads per workgroup: 256 (throughput, many queries) or 1024 (latency, few queries)
__global__ __launch_bounds__(NT) void select_kernel(CandState cs, const int *qsel, int kc,
                                                             uint32_t boot_rows, uint32_t tau_only,
                                                             uint32_t need_at_least, uint32_t sort_max,
                                                             EmitArgs em, uint32_t striped, uint32_t unsorted)
{
    // tau_only: the list holds a *sample* of the rows; publish its kc-th entry (row bits saturated) as
    // the admission threshold and leave the list empty.  need_at_least: a list shorter than this means a
    // sampled threshold admitted too few rows -> flag bit 2, the query is redone without sampling.
    extern __shared__ __attribute__((aligned(16))) uint64_t sh[];
    const int q = qsel ? qsel[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // boot_rows > 0: the bootstrap chunk stored one entry per row without atomics
    // striped: positions s, s+16, ... of the list were handed out by 16 counters (slot = blockIdx.x);
    // the list then has holes, which are read as kEntryMax exactly like masked rows of a bootstrap chunk
    uint32_t my_stripe_cnt = 0;
    if (striped) {
        __shared__ uint32_t s_stripe[LB_STRIPES];
        if (tid < LB_STRIPES) s_stripe[tid] = cs.stripes[(blockIdx.x * LB_STRIPES + tid) * LB_STRIPE_PAD];
        __syncthreads();
        uint32_t mx = 0;
#pragma unroll
        for (int st = 0; st < LB_STRIPES; st++) mx = s_stripe[st] > mx ? s_stripe[st] : mx;
        my_stripe_cnt = s_stripe[tid & (LB_STRIPES - 1)];
        boot_rows = mx * LB_STRIPES; // (0 admissions -> the plain path below sees cnt[q] == 0)
    }
    const uint32_t raw = boot_rows ? boot_rows : cs.cnt[q];
    uint32_t n = raw < cs.cap ? raw : cs.cap;
    uint64_t *list = cs.lists + (size_t)q * cs.cap;
    if (raw > cs.cap && tid == 0) atomicOr(&cs.flags[q], 1u);
    // last select of a scan-path search: write the k results (and the slot's status word, to pinned host
    // memory) from the sorted prefix sorted[0..nsorted) instead of launching emit_lists_kernel
    auto emit = [&](const uint64_t *sorted, uint32_t nsorted) {
        if (em.out_dist == nullptr) return;
        for (int r = tid; r < em.k; r += NT) {
            float d = FLT_MAX;
            int64_t lab = -1;
            if ((uint32_t)r < nsorted) {
                const uint64_t e = sorted[r];
                d = entry_key(e);
                const uint32_t row = entry_row(e);
                lab = em.ids ? em.ids[row] : (int64_t)row;
            }
            em.out_dist[(int64_t)q * em.k + r] = d;
            em.out_labels[(int64_t)q * em.k + r] = lab;
        }
        if (em.flags_host && tid == 0) em.flags_host[q] = atomicOr(&cs.flags[q], 0u);
    };
    if (n == 0) {
        if (tid == 0) {
            cs.tau[q] = kEntryMax;
            cs.cnt[q] = 0;
            if (need_at_least) atomicOr(&cs.flags[q], 4u);
        }
        emit(nullptr, 0);
        return;
    }
    const uint32_t P = next_pow2(n);
    uint32_t *hist = reinterpret_cast<uint32_t *>(sh + next_pow2(cs.cap));
    uint32_t *wsum = hist + 256;
    uint32_t *scal = wsum + 4; // [0]=bucket [1]=need [2]=out counter [3]=valid count [4]=take the whole bucket

    if (tid == 0) { scal[2] = 0; scal[3] = 0; }
    uint32_t myvalid = 0;
    for (uint32_t i = tid; i < P; i += NT) { // (NT is a multiple of 16: a thread stays in one stripe)
        const bool there = i < n && (!striped || (i / LB_STRIPES) < my_stripe_cnt);
        const uint64_t e = there ? list[i] : kEntryMax;
        sh[i] = e;
        myvalid += (e != kEntryMax) ? 1u : 0u;
    }
    __syncthreads();
    if (boot_rows) { // masked-out rows of the bootstrap chunk hold kEntryMax: count the real ones
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) myvalid += __shfl_xor(myvalid, off); // one LDS atomic per wave
        if (lane == 0 && myvalid) atomicAdd(&scal[3], myvalid);
        __syncthreads();
        n = scal[3];
        __syncthreads();
        if (n == 0) {
            if (tid == 0) {
                cs.tau[q] = kEntryMax;
                cs.cnt[q] = 0;
                if (need_at_least) atomicOr(&cs.flags[q], 4u);
            }
            emit(nullptr, 0);
            return;
        }
    }
    const uint32_t keep = n < (uint32_t)kc ? n : (uint32_t)kc;
    if (n < need_at_least && tid == 0) atomicOr(&cs.flags[q], 4u);

    if (P <= 2u * next_pow2((uint32_t)kc) || n <= (uint32_t)kc || P <= sort_max) {
        bitonic_sort_u64(sh, P, tid, NT); // kEntryMax padding sorts last
        if (tau_only) {
            if (tid == 0) {
                cs.cnt[q] = 0;
                cs.tau[q] = n >= (uint32_t)kc ? (sh[kc - 1] | 0xffffffffull) : kEntryMax;
            }
            return;
        }
        for (uint32_t i = tid; i < keep; i += NT) list[i] = sh[i];
        if (tid == 0) {
            cs.cnt[q] = keep;
            cs.tau[q] = n >= (uint32_t)kc ? sh[kc - 1] : kEntryMax;
        }
        emit(sh, keep);
        return;
    }

    // ---- radix select of the kc-th smallest (1-based rank `need`) ----
    // Leading bytes shared by every real entry (distances live in a narrow range) need no pass.
    unsigned long long *red = reinterpret_cast<unsigned long long *>(hist); // [0] = OR, [1] = AND of real entries
    if (tid == 0) { red[0] = 0ull; red[1] = ~0ull; }
    __syncthreads();
    {
        uint64_t o = 0, an = ~0ull;
        for (uint32_t i = tid; i < P; i += NT) {
            const uint64_t e = sh[i];
            if (e != kEntryMax) { o |= e; an &= e; }
        }
        wave_or_and_u64(o, an); // one pair of LDS atomics per wave, not per thread
        if (lane == 63) {
            atomicOr(&red[0], (unsigned long long)o);
            atomicAnd(&red[1], (unsigned long long)an);
        }
    }
    __syncthreads();
    const uint64_t diff = red[0] ^ red[1]; // bit positions that differ among real entries
    const uint64_t common = red[1];
    __syncthreads();
    int first_shift = 56;
    uint64_t prefix = 0, mask = 0;
    if (diff != 0ull) {
        const int same_bytes = __builtin_clzll(diff) >> 3; // whole leading bytes identical
        first_shift = 56 - 8 * same_bytes;
        if (same_bytes > 0) {
            mask = ~0ull << (64 - 8 * same_bytes);
            prefix = common & mask;
        }
    }
    uint32_t need = (uint32_t)kc;
    for (int shift = first_shift; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0; // 256 bins
        __syncthreads();
        for (uint32_t i = tid; i < P; i += NT) {
            const uint64_t e = sh[i];
            if ((e & mask) == prefix) atomicAdd(&hist[(uint32_t)(e >> shift) & 0xffu], 1u);
        }
        __syncthreads();
        const uint32_t h = tid < 256 ? hist[tid] : 0u;
        uint32_t incl = wave_incl_scan(h, lane);
        if (tid < 256 && lane == 63) wsum[wave] = incl;
        __syncthreads();
        if (tid < 256) {
            uint32_t base = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) base += (w < wave) ? wsum[w] : 0u;
            incl += base;
            const uint32_t excl = incl - h;
            if (excl < need && need <= incl) {
                scal[0] = (uint32_t)tid;
                scal[1] = need - excl;
                scal[4] = (need == incl) ? 1u : 0u; // the whole bucket is wanted: no need to split it further
            }
        }
        __syncthreads();
        prefix |= (uint64_t)scal[0] << shift;
        mask |= 0xffull << shift;
        need = scal[1];
        const bool whole_bucket = scal[4] != 0u;
        __syncthreads();
        if (whole_bucket) { // (entries are unique, so the row bytes are rarely ever walked)
            prefix |= ~mask; // every entry sharing the resolved bytes is kept; this bound is >= the kc-th entry
            break;
        }
    }
    const uint64_t pivot = prefix; // >= the kc-th smallest entry and < the (kc+1)-th: exactly kc entries are <= pivot
    if (tau_only) {
        if (tid == 0) {
            cs.cnt[q] = 0;
            cs.tau[q] = pivot | 0xffffffffull;
        }
        return;
    }
    // compact the kc entries <= pivot (unordered), then sort them.  The staging area is the unused tail of
    // the LDS entry array when there is room, else the front of the global list.
    const uint32_t Pk = next_pow2((uint32_t)kc);
    uint64_t *stage = (P + Pk <= next_pow2(cs.cap)) ? sh + P : list;
    __syncthreads();
    for (uint32_t i = tid; i < P; i += NT) {
        const uint64_t e = sh[i];
        if (e <= pivot) {
            const uint32_t pos = atomicAdd(&scal[2], 1u);
            stage[pos] = e; // pos < kc by construction
        }
    }
    __syncthreads();
    if (unsorted) {
        // the caller re-ranks the kept entries anyway and only needs to know which one has the worst key: it goes to
        // position kc - 1 (one swap), the other kc - 1 stay as the compaction left them -- no sort (-4 us)
        unsigned long long *mx = reinterpret_cast<unsigned long long *>(hist);
        if (tid == 0) mx[0] = 0ull;
        __syncthreads();
        unsigned long long m = 0ull;
        for (uint32_t i = tid; i < (uint32_t)kc; i += NT) m = stage[i] > m ? stage[i] : m;
        if (m) atomicMax(&mx[0], m);
        __syncthreads();
        const uint64_t worst = mx[0];
        __syncthreads();
        for (uint32_t i = tid; i < (uint32_t)kc; i += NT) {
            const uint64_t e = stage[i];
            if (e == worst && i != (uint32_t)kc - 1) { // entries are unique: one thread swaps
                stage[i] = stage[kc - 1];
                stage[kc - 1] = e;
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < (uint32_t)kc; i += NT) list[i] = stage[i];
        if (tid == 0) {
            cs.cnt[q] = (uint32_t)kc;
            cs.tau[q] = pivot;
        }
        return; // (no emit on this route: the re-rank writes the results)
    }
    // (this path has P > 2 * Pk, so sh[0..Pk) and the staging tail sh[P..P+Pk) do not overlap)
    for (uint32_t i = tid; i < Pk; i += NT) sh[i] = i < (uint32_t)kc ? stage[i] : kEntryMax;
    __syncthreads();
    bitonic_sort_u64(sh, Pk, tid, NT);
    for (uint32_t i = tid; i < (uint32_t)kc; i += NT) list[i] = sh[i];
    if (tid == 0) {
        cs.cnt[q] = (uint32_t)kc;
        cs.tau[q] = pivot;
    }
    emit(sh, (uint32_t)kc);
}

// smap[i] = corpus row behind the i-th of `count` evenly spaced positions of [0, span)
__global__ __launch_bounds__(256) void sample_map_kernel(const uint32_t *rowmap, int64_t span, uint32_t count,
                                                         uint32_t *smap)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const int64_t pos = (int64_t)(((uint64_t)i * (uint64_t)span) / count); // i, span < 2^32
    smap[i] = rowmap ? rowmap[pos] : (uint32_t)pos;
}

void launch_sample_map(const uint32_t *rowmap, int64_t span, uint32_t count, uint32_t *smap, hipStream_t s)
{
    if (count == 0) return;
    hipLaunchKernelGGL(sample_map_kernel, dim3((count + 255) / 256), dim3(256), 0, s, rowmap, span, count, smap);
}

void launch_select(CandState cs, const int *qsel, int nsel, int kc, uint32_t boot_rows, hipStream_t s,
                   bool tau_only, uint32_t need_at_least, const EmitArgs *emit, bool striped, bool unsorted)
{
    EmitArgs em{};
    if (emit) em = *emit;
    if (nsel <= 0) return;
    const size_t shmem = (size_t)next_pow2_host(cs.cap) * sizeof(uint64_t) + (256 + 4 + 8) * sizeof(uint32_t);
    // lists up to this size are simply sorted (one 1024-thread workgroup); larger ones take the radix select
    static const uint32_t sort_max_big = (uint32_t)lb_tunable("LB_SELECT_SORT_MAX", 0);
    static const int big_max = lb_tunable("LB_SELECT_BIG_MAXQ", 512); // (1024-thread selects: 10 us less than 256-thread ones at 128-384 queries)
    const uint32_t sort_max = nsel <= big_max ? sort_max_big : 0u;
    if (nsel <= big_max) { // few queries: one big workgroup each, latency matters
        allow_big_lds(select_kernel<1024>, shmem);
        hipLaunchKernelGGL(select_kernel<1024>, dim3(nsel), dim3(1024), shmem, s, cs, qsel, kc, boot_rows,
                           tau_only ? 1u : 0u, need_at_least, sort_max, em,
                           (striped && cs.stripes) ? 1u : 0u, (unsorted && emit == nullptr) ? 1u : 0u);
    } else {
        allow_big_lds(select_kernel<256>, shmem);
        hipLaunchKernelGGL(select_kernel<256>, dim3(nsel), dim3(256), shmem, s, cs, qsel, kc, boot_rows,
                           tau_only ? 1u : 0u, need_at_least, sort_max, em,
                           (striped && cs.stripes) ? 1u : 0u, (unsorted && emit == nullptr) ? 1u : 0u);
    }
}

// ---------------------------------------------------------------------------
struct RerankArgs {
    const float *X;
    int D;
    const float *Q;
    const float *qna;
    CandState cs;
    int kc, k;
    const uint32_t *maxnorm2;
    float gamma;
    const int64_t *ids;
    float *out_dist;
    int64_t *out_labels;
    int aligned;
    uint32_t *flags_host; // pinned host copy of the slots' status words (no D2H copy after the batch), or null
    uint32_t *done;       // [nq] arrival tickets of rerank_split_kernel (zero between launches), or null
};

// Shared tail of the re-rank kernels: containment check, final (distance,row) ordering, output.
template <int METRIC>
__device__ __forceinline__ void rerank_finish(const RerankArgs &a, int qi, int tid, uint32_t nc, uint32_t P,
                                              const float *sq, uint64_t *skey, const float *scmp,
                                              unsigned int &s_count, float &s_w, float na)
{
    const int D = a.D;
    // ---- containment check (only meaningful when the list is full: rows were left out) ----
    // Any row y outside the list has approx_key(y) >= approx_key(c_last); with rounding-error
    // bound E on both evaluations, exact_cmp(y) >= w - E.  If at least k candidates satisfy
    // cmp < w - E (strictly), the true top-k lies inside the list.
    if (nc >= (uint32_t)a.kc && nc > (uint32_t)a.k) {
        // ga: error bound of the candidate inner product per unit of |q||x| (host-provided, depends
        // on the contraction: f32 fma chain or split-bf16); go: the same for the exact f32 re-rank sums
        const float ga = a.gamma;
        const float go = 1.05f * (float)(D + 8) * 5.9604645e-8f; // (D+8) * 2^-24
        const float xmax2 = __builtin_bit_cast(float, *a.maxnorm2);
        const float xmax = sqrtf(xmax2) * 1.000001f;
        const float w = s_w;
        float T;
        bool skip = false;
        if (METRIC == METRIC_L2) {
            float nq2 = 0.f;
            for (int i = 0; i < D; i++) nq2 += sq[i] * sq[i];
            const float nqn = sqrtf(nq2) * 1.000001f;
            // d^2 space: key error go*|x|^2 + 2*ga*|q||x| for each of c_last and the outsider;
            // exact side relative go on each of the two d^2 values.
            // The key errors only have to cover rows up to the norm R = |q| + d(c_last): a row beyond it is farther than
            // c_last whatever its key says (d(y) >= |y| - |q| > sqrt(w), and its computed d^2 stays above w >= T through the
            // relative slack below), and c_last itself lies inside (|c| <= |q| + d(c)).  With a few rows of a much larger norm
            // than the rest -- unnormalised data -- the corpus maximum would inflate the bound for every query.
            float xe = xmax;
            if (w >= 0.0f && w < FLT_MAX) xe = fminf(xmax, (nqn * 1.001f + sqrtf(w) * (1.001f + 2.0f * go)) * 1.001f);
            T = w * (1.0f - 3.0f * go) - 2.2f * (go * xe * xe + 2.0f * ga * nqn * xe);
        } else if (METRIC == METRIC_COS) {
            T = w - 2.2f * (ga + 2.8f * go);
            skip = (na == 0.0f); // all distances are exactly 1.0; selection by row is exact
        } else {
            float nq2 = 0.f;
            for (int i = 0; i < D; i++) nq2 += sq[i] * sq[i];
            const float nqn = sqrtf(nq2) * 1.000001f;
            T = w - 2.2f * (ga + go) * nqn * xmax;
        }
        T = T - fabsf(T) * 1e-6f;
        unsigned int local = 0;
        for (uint32_t c = tid; c < nc; c += SEL_THREADS) local += (scmp[c] < T) ? 1u : 0u;
        if (local) atomicAdd(&s_count, local);
        __syncthreads();
        if (tid == 0 && !skip && s_count < (unsigned int)a.k) atomicOr(&a.cs.flags[qi], 2u);
    }

    bitonic_sort_u64(skey, P, tid, SEL_THREADS);

    for (int r = tid; r < a.k; r += SEL_THREADS) {
        float d = FLT_MAX;
        int64_t lab = -1;
        if ((uint32_t)r < nc) {
            const uint64_t e = skey[r];
            d = entry_key(e);
            const uint32_t row = entry_row(e);
            lab = a.ids ? a.ids[row] : (int64_t)row;
        }
        a.out_dist[(int64_t)qi * a.k + r] = d;
        a.out_labels[(int64_t)qi * a.k + r] = lab;
    }
    if (a.flags_host && tid == 0) a.flags_host[qi] = atomicOr(&a.cs.flags[qi], 0u);
}

// One workgroup per query.  LDS: q[D] | sort keys u64[P] | cmp values f32[P]
template <int METRIC, int ORDER>
__global__ __launch_bounds__(SEL_THREADS) void rerank_kernel(RerankArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int qi = blockIdx.x;
    const int tid = threadIdx.x;
    const int D = a.D;
    const uint32_t nc = a.cs.cnt[qi];
    const uint32_t P = next_pow2(nc > 0 ? nc : 1);
    const int Dpad = (D + 3) & ~3;
    float *sq = reinterpret_cast<float *>(smem);
    // all LDS in the one dynamic region (16-B aligned carve offsets, no static __shared__)
    uint64_t *skey = reinterpret_cast<uint64_t *>(smem + (size_t)Dpad * 4);
    const uint32_t Pmax = next_pow2((uint32_t)a.kc);
    float *scmp = reinterpret_cast<float *>(skey + Pmax);
    unsigned int &s_count = *reinterpret_cast<unsigned int *>(scmp + Pmax);
    float &s_w = *reinterpret_cast<float *>(scmp + Pmax + 1);

    const float *q = a.Q + (int64_t)qi * D;
    for (int i = tid; i < D; i += SEL_THREADS) sq[i] = q[i];
    if (tid == 0) { s_count = 0; s_w = 0.f; }
    __syncthreads();

    const uint64_t *list = a.cs.lists + (size_t)qi * a.cs.cap;
    const int dmain = D & ~3;
    const float na = (METRIC == METRIC_COS) ? a.qna[qi] : 0.f;

    for (uint32_t c = tid; c < P; c += SEL_THREADS) {
        if (c >= nc) {
            skey[c] = kEntryMax;
            scmp[c] = FLT_MAX;
            continue;
        }
        const uint32_t row = entry_row(list[c]);
        const float *x = a.X + (int64_t)row * D;
        AccR<ORDER> acc, nb;
        acc.zero();
        nb.zero();
        // each lane walks its own row: keep 16 independent 16-B loads (two 128-B lines) in flight
#pragma unroll 16
        for (int i = 0; i < dmain; i += 4) {
            float x0, x1, x2, x3;
            if (a.aligned) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(x + i);
                x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w;
            } else {
                x0 = x[i]; x1 = x[i + 1]; x2 = x[i + 2]; x3 = x[i + 3];
            }
            const f32x4 qv = *reinterpret_cast<const f32x4 *>(&sq[i]);
            if (METRIC == METRIC_COS) {
                nb.template add<0>(x0 * x0);
                nb.template add<1>(x1 * x1);
                nb.template add<2>(x2 * x2);
                nb.template add<3>(x3 * x3);
            }
            if (METRIC == METRIC_L2) {
                const float e0 = qv.x - x0, e1 = qv.y - x1, e2 = qv.z - x2, e3 = qv.w - x3;
                acc.template add<0>(e0 * e0);
                acc.template add<1>(e1 * e1);
                acc.template add<2>(e2 * e2);
                acc.template add<3>(e3 * e3);
            } else {
                acc.template add<0>(qv.x * x0);
                acc.template add<1>(qv.y * x1);
                acc.template add<2>(qv.z * x2);
                acc.template add<3>(qv.w * x3);
            }
        }
        for (int i = dmain; i < D; i++) {
            const float xv = x[i], qv = sq[i];
            if (METRIC == METRIC_COS) nb.add_tail(xv * xv);
            if (METRIC == METRIC_L2) {
                const float e = qv - xv;
                acc.add_tail(e * e);
            } else {
                acc.add_tail(qv * xv);
            }
        }
        const float t = acc.total();
        float dist, cmp;
        if (METRIC == METRIC_L2) {
            dist = (float)sqrt((double)t);
            cmp = t; // compare in d^2 space
        } else if (METRIC == METRIC_COS) {
            const float nbt = nb.total();
            if (D == 0 || na == 0.0f || nbt == 0.0f) dist = 1.0f;
            else dist = 1.0f - __fdiv_rn(t, (float)sqrt((double)na * (double)nbt));
            cmp = dist;
        } else {
            dist = -t;
            cmp = dist;
        }
        skey[c] = pack_entry(dist, row);
        scmp[c] = cmp;
        if (c == (uint32_t)a.kc - 1) s_w = cmp; // the candidate with the worst approximate key
    }
    __syncthreads();

    rerank_finish<METRIC>(a, qi, tid, nc, P, sq, skey, scmp, s_count, s_w, na);
}

// ---------------------------------------------------------------------------
// rerank_tiled_kernel: same arithmetic as rerank_kernel, but the candidate rows are gathered through
// LDS in coalesced 256-B pieces (lane = candidate walks its own row out of LDS), instead of every lane
// striding through HBM on its own: 16 independent 16-B loads per lane in flight, full lines per row.
// One workgroup (256 lanes) per query, candidates processed in groups of 256.
// LDS: q[Dpad] | row ids u32[256] | tile f32[256][68] | keys u64[P] | cmp f32[P] | scalars
constexpr int RR_DK = 64, RR_LD = RR_DK + 4;

template <int METRIC, int ORDER>
__global__ __launch_bounds__(SEL_THREADS) void rerank_tiled_kernel(RerankArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int qi = blockIdx.x;
    const int tid = threadIdx.x;
    const int D = a.D; // D % 4 == 0 and 16-B aligned rows (checked by the launcher)
    const uint32_t nc = a.cs.cnt[qi];
    const uint32_t P = next_pow2(nc > 0 ? nc : 1);
    const uint32_t Pmax = next_pow2((uint32_t)a.kc);
    const int Dpad = (D + 63) & ~63;
    float *sq = reinterpret_cast<float *>(smem);
    uint32_t *srow = reinterpret_cast<uint32_t *>(sq + Dpad);
    float *tile = reinterpret_cast<float *>(srow + SEL_THREADS);
    uint64_t *skey = reinterpret_cast<uint64_t *>(tile + SEL_THREADS * RR_LD);
    float *scmp = reinterpret_cast<float *>(skey + Pmax);
    unsigned int &s_count = *reinterpret_cast<unsigned int *>(scmp + Pmax);
    float &s_w = *reinterpret_cast<float *>(scmp + Pmax + 1);

    const float *q = a.Q + (int64_t)qi * D;
    for (int i = tid; i < Dpad; i += SEL_THREADS) sq[i] = i < D ? q[i] : 0.f;
    if (tid == 0) { s_count = 0; s_w = 0.f; }
    const uint64_t *list = a.cs.lists + (size_t)qi * a.cs.cap;
    const float na = (METRIC == METRIC_COS) ? a.qna[qi] : 0.f;
    const int nchunks = (D + RR_DK - 1) / RR_DK;

    for (uint32_t g0 = 0; g0 < P; g0 += SEL_THREADS) {
        const uint32_t c = g0 + tid;
        const uint32_t myrow = c < nc ? entry_row(list[c]) : (nc ? entry_row(list[0]) : 0u);
        __syncthreads(); // previous group's tile reads are done; sq / s_* initialised
        srow[tid] = myrow;
        __syncthreads();
        AccR<ORDER> acc, nb;
        acc.zero();
        nb.zero();
        f32x4 stg[16];
        auto load_stage = [&](int ch) {
            const int d0 = ch * RR_DK;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int idx = tid + SEL_THREADS * i;
                const int r = idx >> 4, p = idx & 15;
                int k = d0 + p * 4;
                if (k > D - 4) k = D - 4; // pieces past D are never consumed
                stg[i] = *reinterpret_cast<const f32x4 *>(a.X + (int64_t)srow[r] * D + k);
            }
        };
        load_stage(0);
        for (int ch = 0; ch < nchunks; ch++) {
            __syncthreads(); // tile free
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int idx = tid + SEL_THREADS * i;
                *reinterpret_cast<f32x4 *>(&tile[(idx >> 4) * RR_LD + (idx & 15) * 4]) = stg[i];
            }
            __syncthreads();
            if (ch + 1 < nchunks) load_stage(ch + 1); // in flight under the compute below
            const int d0 = ch * RR_DK;
            const int n4 = (min(D, d0 + RR_DK) - d0) >> 2;
            const float *xr = &tile[tid * RR_LD];
#pragma unroll 4
            for (int g = 0; g < n4; g++) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(&xr[g * 4]);
                const f32x4 qv = *reinterpret_cast<const f32x4 *>(&sq[d0 + g * 4]);
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
        if (c < P) {
            if (c >= nc) {
                skey[c] = kEntryMax;
                scmp[c] = FLT_MAX;
            } else {
                const float t = acc.total();
                float dist, cmp;
                if (METRIC == METRIC_L2) {
                    dist = (float)sqrt((double)t);
                    cmp = t;
                } else if (METRIC == METRIC_COS) {
                    const float nbt = nb.total();
                    if (D == 0 || na == 0.0f || nbt == 0.0f) dist = 1.0f;
                    else dist = 1.0f - __fdiv_rn(t, (float)sqrt((double)na * (double)nbt));
                    cmp = dist;
                } else {
                    dist = -t;
                    cmp = dist;
                }
                skey[c] = pack_entry(dist, myrow);
                scmp[c] = cmp;
                if (c == (uint32_t)a.kc - 1) s_w = cmp;
            }
        }
    }
    __syncthreads();
    rerank_finish<METRIC>(a, qi, tid, nc, P, sq, skey, scmp, s_count, s_w, na);
}

// ---------------------------------------------------------------------------
// rerank_split_kernel: the same arithmetic again, spread over kc/16 workgroups per query.  A workgroup owns 16
// candidates: its 256 lanes fetch the 16 rows whole (up to 1024 dims per stage, every 16-B piece in flight at once:
// ONE memory round trip for D <= 1024 instead of D/64 dependent stages), 16 lanes then walk one row each out of LDS
// in the reference's order and leave (exact entry, compare value) behind the list; the last workgroup of a query to
// arrive (ticket counter, reset for the next launch) does the containment check, the final ordering and the output.
// Measured at 1M x 768, kc = 256: 39 us -> see DESIGN.md for the whole-search effect.
// LDS: tile f32[16][1028] | q stage f32[1024] | rows u32[16]   (finish: q[Dpad] | keys u64[P] | cmp f32[P] | scalars)
constexpr int RS_ROWS = 16, RS_SD = 1024, RS_LD = RS_SD + 4;

template <int METRIC, int ORDER>
__global__ __launch_bounds__(SEL_THREADS) void rerank_split_kernel(RerankArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_last;
    const int qi = blockIdx.y;
    const int tid = threadIdx.x;
    const int D = a.D; // D % 4 == 0 and 16-B aligned rows (checked by the launcher)
    const uint32_t Pmax = next_pow2((uint32_t)a.kc);
    uint64_t *list = a.cs.lists + (size_t)qi * a.cs.cap;
    // this workgroup's candidates, fetched beside the count (positions < Pmax <= cap are always readable)
    const uint64_t my_entry = tid < RS_ROWS ? list[blockIdx.x * RS_ROWS + tid] : 0ull;
    const uint32_t nc = min(a.cs.cnt[qi], (uint32_t)a.kc);
    uint64_t *exact = list + Pmax;                                   // [Pmax] exact entries (cap >= 4 kc)
    float *cmpv = reinterpret_cast<float *>(list + 2 * (size_t)Pmax); // [Pmax] compare values
    const float na = (METRIC == METRIC_COS) ? a.qna[qi] : 0.f;
    const uint32_t c0 = blockIdx.x * RS_ROWS;

    if (c0 < nc) { // workgroup-uniform
        float *tile = reinterpret_cast<float *>(smem);
        float *qs = tile + RS_ROWS * RS_LD;
        uint32_t *srow = reinterpret_cast<uint32_t *>(qs + RS_SD);
        const uint64_t first_entry = __shfl(my_entry, 0); // c0 < nc: lane 0 holds a real candidate
        if (tid < RS_ROWS) srow[tid] = entry_row(c0 + tid < nc ? my_entry : first_entry);
        __syncthreads();
        const float *q = a.Q + (int64_t)qi * D;
        f32x4 stg[RS_ROWS], stq;
        auto load_stage = [&](int d0) {
            const int k = d0 + tid * 4;
            if (k < D) {
                stq = *reinterpret_cast<const f32x4 *>(q + k);
#pragma unroll
                for (int r = 0; r < RS_ROWS; r++) stg[r] = *reinterpret_cast<const f32x4 *>(a.X + (int64_t)srow[r] * D + k);
            }
        };
        // UNROLL4: the four accumulator chains of a row are independent, so four lanes share a row (lane t owns
        // chain t: elements 4g + t) and the quad's first lane adds them up in the reference's order; SEQ: one lane per row.
        constexpr bool QUAD = ORDER == ORDER_UNROLL4;
        const int myr = QUAD ? (tid >> 2) : tid, myt = QUAD ? (tid & 3) : 0;
        const bool worker = tid < (QUAD ? 4 * RS_ROWS : RS_ROWS);
        float a0 = 0.f, b0 = 0.f; // this lane's chain of acc / nb (SEQ: the only chain)
        load_stage(0);
        for (int d0 = 0; d0 < D; d0 += RS_SD) {
            if (d0 + tid * 4 < D) {
                *reinterpret_cast<f32x4 *>(&qs[tid * 4]) = stq;
#pragma unroll
                for (int r = 0; r < RS_ROWS; r++) *reinterpret_cast<f32x4 *>(&tile[r * RS_LD + tid * 4]) = stg[r];
            }
            __syncthreads();
            if (d0 + RS_SD < D) load_stage(d0 + RS_SD); // in flight under the compute below
            if (worker) {
                const int nel = min(D, d0 + RS_SD) - d0;
                const float *xr = &tile[myr * RS_LD];
                if (QUAD) {
#pragma unroll 8
                    for (int e = myt; e < nel; e += 4) {
                        const float xv = xr[e], qv = qs[e];
                        if (METRIC == METRIC_COS) b0 = b0 + xv * xv;
                        if (METRIC == METRIC_L2) {
                            const float d = qv - xv;
                            a0 = a0 + d * d;
                        } else {
                            a0 = a0 + qv * xv;
                        }
                    }
                } else {
#pragma unroll 4
                    for (int g = 0; g < (nel >> 2); g++) {
                        const f32x4 xv = *reinterpret_cast<const f32x4 *>(&xr[g * 4]);
                        const f32x4 qv = *reinterpret_cast<const f32x4 *>(&qs[g * 4]);
                        const float xe[4] = {xv.x, xv.y, xv.z, xv.w}, qe[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (METRIC == METRIC_COS) b0 = b0 + xe[u] * xe[u];
                            if (METRIC == METRIC_L2) {
                                const float d = qe[u] - xe[u];
                                a0 = a0 + d * d;
                            } else {
                                a0 = a0 + qe[u] * xe[u];
                            }
                        }
                    }
                }
            }
            __syncthreads(); // the stage is free again
        }
        float t = a0, nbt = b0;
        if (QUAD) { // (s0 + s1) + s2 + s3, as the reference's unrolled loops finish
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
        const uint32_t c = c0 + myr;
        if (worker && myt == 0 && c < nc) {
            float dist, cmp;
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
            __hip_atomic_store(&exact[c], pack_entry(dist, srow[myr]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&cmpv[c], cmp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- the last workgroup of the query to get here finishes it ----
    // The results above went out as device-scope (write-through) stores and are read back below with device-scope
    // loads, so waiting for their acknowledgement orders them before the ticket; a __threadfence() here would write
    // back the whole per-XCD L2 once per workgroup (measured: 3.3 ms instead of 0.1 at 1024 queries).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = atomicAdd(&a.done[qi], 1u);
        s_last = (t == gridDim.x - 1u) ? 1 : 0;
        if (s_last) a.done[qi] = 0; // ready for the next launch on this workspace
    }
    __syncthreads();
    if (!s_last) return;

    const uint32_t P = next_pow2(nc > 0 ? nc : 1);
    const int Dpad = (D + 3) & ~3;
    float *sq = reinterpret_cast<float *>(smem);
    uint64_t *skey = reinterpret_cast<uint64_t *>(smem + (size_t)Dpad * 4);
    float *scmp = reinterpret_cast<float *>(skey + Pmax);
    unsigned int &s_count = *reinterpret_cast<unsigned int *>(scmp + Pmax);
    float &s_w = *reinterpret_cast<float *>(scmp + Pmax + 1);
    if (METRIC != METRIC_COS) { // the containment bound of L2 / dot needs |q|
        const float *q = a.Q + (int64_t)qi * D;
        for (int i = tid; i < D; i += SEL_THREADS) sq[i] = q[i];
    }
    if (tid == 0) { s_count = 0; s_w = 0.f; }
    __syncthreads();
    for (uint32_t c = tid; c < P; c += SEL_THREADS) {
        uint64_t e = kEntryMax;
        float cmp = FLT_MAX;
        if (c < nc) { // written by other workgroups of this launch: read at device scope, not through this CU's L1
            e = __hip_atomic_load(&exact[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cmp = __hip_atomic_load(&cmpv[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        skey[c] = e;
        scmp[c] = cmp;
        if (c == (uint32_t)a.kc - 1 && c < nc) s_w = cmp; // the candidate with the worst approximate key
    }
    __syncthreads();
    rerank_finish<METRIC>(a, qi, tid, nc, P, sq, skey, scmp, s_count, s_w, na);
}

void launch_rerank(int metric, int order, const float *X, int D, const float *Q, int nq,
                   const float *qna, CandState cs, int kc, int k, const uint32_t *d_maxnorm2, float gamma,
                   const int64_t *ids, float *out_dist, int64_t *out_labels, hipStream_t s, uint32_t *flags_host, uint32_t *done)
{
    if (nq <= 0) return;
    RerankArgs a;
    a.flags_host = flags_host;
    a.done = done;
    a.X = X; a.D = D; a.Q = Q; a.qna = qna; a.cs = cs; a.kc = kc; a.k = k;
    a.maxnorm2 = d_maxnorm2; a.gamma = gamma; a.ids = ids; a.out_dist = out_dist; a.out_labels = out_labels;
    a.aligned = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const int Dpad = (D + 3) & ~3;
    const size_t P = next_pow2_host((uint32_t)kc);
    const size_t shmem = (size_t)Dpad * 4 + P * 8 + P * 4 + 16;
    dim3 grid(nq), block(SEL_THREADS);
    static const int split_on = lb_tunable("LB_RERANK_SPLIT", 1);
    static const int split_max_wg = lb_tunable("LB_RERANK_SPLIT_MAXWG", 512);
    // Worth it while the whole launch is resident at once (a workgroup's life is a chain of ~8 memory round trips, two
    // resident per CU): 30 vs 38-40 us at 8 queries, 33 vs 40 at 16, 35 vs 39 at 32, slower from 64 on (66 vs 42 us).
    if (a.aligned && D >= 4 && done != nullptr && split_on && cs.cap >= 3u * (uint32_t)P && (size_t)nq * (P / RS_ROWS) <= (size_t)split_max_wg) {
        const size_t sh_rows = (size_t)RS_ROWS * RS_LD * 4 + RS_SD * 4 + RS_ROWS * 4;
        const size_t sh3 = std::max(sh_rows, shmem);
        dim3 grid3((unsigned)(P / RS_ROWS), (unsigned)nq);
#define LB_RRS(M, O)                                                                \
    do {                                                                            \
        allow_big_lds(rerank_split_kernel<M, O>, sh3);                              \
        hipLaunchKernelGGL((rerank_split_kernel<M, O>), grid3, block, sh3, s, a);   \
    } while (0)
        if (metric == METRIC_L2) { if (order == ORDER_UNROLL4) LB_RRS(METRIC_L2, ORDER_UNROLL4); else LB_RRS(METRIC_L2, ORDER_SEQ); }
        else if (metric == METRIC_COS) { if (order == ORDER_UNROLL4) LB_RRS(METRIC_COS, ORDER_UNROLL4); else LB_RRS(METRIC_COS, ORDER_SEQ); }
        else { if (order == ORDER_UNROLL4) LB_RRS(METRIC_DOT, ORDER_UNROLL4); else LB_RRS(METRIC_DOT, ORDER_SEQ); }
#undef LB_RRS
        return;
    }
    if (a.aligned && D >= 4) {
        const int Dq = (D + 63) & ~63;
        const size_t sh2 = (size_t)Dq * 4 + SEL_THREADS * 4 + (size_t)SEL_THREADS * RR_LD * 4 + P * 8 + P * 4 + 16;
#define LB_RRT(M, O)                                                                \
    do {                                                                            \
        allow_big_lds(rerank_tiled_kernel<M, O>, sh2);                              \
        hipLaunchKernelGGL((rerank_tiled_kernel<M, O>), grid, block, sh2, s, a);    \
    } while (0)
        if (metric == METRIC_L2) { if (order == ORDER_UNROLL4) LB_RRT(METRIC_L2, ORDER_UNROLL4); else LB_RRT(METRIC_L2, ORDER_SEQ); }
        else if (metric == METRIC_COS) { if (order == ORDER_UNROLL4) LB_RRT(METRIC_COS, ORDER_UNROLL4); else LB_RRT(METRIC_COS, ORDER_SEQ); }
        else { if (order == ORDER_UNROLL4) LB_RRT(METRIC_DOT, ORDER_UNROLL4); else LB_RRT(METRIC_DOT, ORDER_SEQ); }
#undef LB_RRT
        return;
    }
    allow_big_lds(rerank_kernel<METRIC_L2, ORDER_SEQ>, shmem);
    allow_big_lds(rerank_kernel<METRIC_L2, ORDER_UNROLL4>, shmem);
    allow_big_lds(rerank_kernel<METRIC_COS, ORDER_SEQ>, shmem);
    allow_big_lds(rerank_kernel<METRIC_COS, ORDER_UNROLL4>, shmem);
    allow_big_lds(rerank_kernel<METRIC_DOT, ORDER_SEQ>, shmem);
    allow_big_lds(rerank_kernel<METRIC_DOT, ORDER_UNROLL4>, shmem);
#define LB_RR(M)                                                                                   \
    do {                                                                                           \
        if (order == ORDER_UNROLL4) hipLaunchKernelGGL((rerank_kernel<M, ORDER_UNROLL4>), grid, block, shmem, s, a); \
        else hipLaunchKernelGGL((rerank_kernel<M, ORDER_SEQ>), grid, block, shmem, s, a);           \
    } while (0)
    if (metric == METRIC_L2) LB_RR(METRIC_L2);
    else if (metric == METRIC_COS) LB_RR(METRIC_COS);
    else LB_RR(METRIC_DOT);
#undef LB_RR
}

// ---------------------------------------------------------------------------
// scan path: lists already hold exact distances, sorted by the last select.
__global__ void emit_lists_kernel(CandState cs, const int *qsel, int nsel, int k, const int64_t *ids,
                                  float *out_dist, int64_t *out_labels, uint32_t *flags_host)
{
    const int j = blockIdx.x;
    if (j >= nsel) return;
    const int q = qsel ? qsel[j] : j;
    // the slot's status word goes straight to pinned host memory: no separate D2H copy after the search
    if (flags_host && threadIdx.x == 0) flags_host[q] = cs.flags[q];
    const uint32_t n = cs.cnt[q];
    const uint64_t *list = cs.lists + (size_t)q * cs.cap;
    for (int r = threadIdx.x; r < k; r += blockDim.x) {
        float d = FLT_MAX;
        int64_t lab = -1;
        if ((uint32_t)r < n) {
            const uint64_t e = list[r];
            d = entry_key(e);
            const uint32_t row = entry_row(e);
            lab = ids ? ids[row] : (int64_t)row;
        }
        out_dist[(int64_t)q * k + r] = d;
        out_labels[(int64_t)q * k + r] = lab;
    }
}

void launch_emit_lists(CandState cs, const int *qsel, int nsel, int k, const int64_t *ids,
                       float *out_dist, int64_t *out_labels, uint32_t *flags_host, hipStream_t s)
{
    if (nsel <= 0) return;
    hipLaunchKernelGGL(emit_lists_kernel, dim3(nsel), dim3(128), 0, s, cs, qsel, nsel, k, ids, out_dist,
                       out_labels, flags_host);
}

// ---------------------------------------------------------------------------
// Cross-shard merge: per query S*k (dist,label) pairs -> k smallest by (dist, label).
// The LDS sort runs on ONE u64 per entry, (canonical sortable distance << 32) | input index, so 16384
// entries (k = 2048 on 8 shards) fit in 128 KB; labels stay in HBM and are fetched for the k winners.
// Canonical order as in the single-shard lists: every NaN after +inf, padding (label < 0) last of all.
// The index tie-break is (shard, position); entries with EQUAL distances are then re-ranked by label
// (unsigned, so that padding stays last) in a fix-up pass that only touches runs of equal distance.
__global__ __launch_bounds__(SEL_THREADS) void merge_topk_kernel(int nshards, int64_t nq, int k,
                                                                 const float *dist_in,
                                                                 const int64_t *lab_in,
                                                                 int64_t dist_stride, int64_t lab_stride,
                                                                 float *dist_out, int64_t *lab_out)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t mkey[];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t n = (uint32_t)(nshards * k);
    const uint32_t P = next_pow2(n);
    auto label_of = [&](uint32_t idx) -> uint64_t {
        const uint32_t s = idx / (uint32_t)k, r = idx % (uint32_t)k;
        return (uint64_t)lab_in[(int64_t)s * lab_stride + q * k + r];
    };
    for (uint32_t i = tid; i < P; i += SEL_THREADS) {
        uint64_t e = kEntryMax;
        if (i < n) {
            const uint32_t s = i / (uint32_t)k, r = i % (uint32_t)k;
            const int64_t off = q * k + r; // within one shard's [nq][k] block
            const float d = dist_in[(int64_t)s * dist_stride + off] + 0.0f; // -0 -> +0
            const int64_t lab = lab_in[(int64_t)s * lab_stride + off];
            uint32_t sk = (d != d) ? 0xffc00000u : f32_sortable(d);
            if (lab < 0) sk = 0xffffffffu; // padding sorts after every real row, non-finite ones included
            e = ((uint64_t)sk << 32) | i;
        }
        mkey[i] = e;
    }
    __syncthreads();
    bitonic_sort_u64(mkey, P, tid, SEL_THREADS);
    // fix-up + output: an entry inside a run of equal distances moves to its label rank within the run
    for (uint32_t i = tid; i < n; i += SEL_THREADS) {
        const uint64_t e = mkey[i];
        const uint32_t sk = (uint32_t)(e >> 32);
        uint32_t lo = i, hi = i + 1;
        while (lo > 0 && (uint32_t)(mkey[lo - 1] >> 32) == sk) lo--;
        if (lo >= (uint32_t)k) continue; // the whole run lies beyond the output
        while (hi < n && (uint32_t)(mkey[hi] >> 32) == sk) hi++;
        uint32_t pos = i;
        const uint64_t mylab = label_of((uint32_t)e);
        if (hi - lo > 1) {
            pos = lo;
            for (uint32_t j = lo; j < hi; j++) {
                if (j == i) continue;
                const uint64_t lj = label_of((uint32_t)mkey[j]);
                pos += (lj < mylab || (lj == mylab && j < i)) ? 1u : 0u;
            }
        }
        if (pos < (uint32_t)k) {
            const bool pad = sk == 0xffffffffu;
            dist_out[q * k + pos] = pad ? FLT_MAX : sortable_f32(sk);
            lab_out[q * k + pos] = pad ? (int64_t)-1 : (int64_t)mylab;
        }
    }
    // fewer than k inputs in total (never with nshards >= 1, kept for safety)
    for (uint32_t r = n + tid; r < (uint32_t)k; r += SEL_THREADS) {
        dist_out[q * k + r] = FLT_MAX;
        lab_out[q * k + r] = -1;
    }
}

void launch_merge_topk(int nshards, int64_t nq, int k, const float *dist_in, const int64_t *lab_in,
                       int64_t dist_stride, int64_t lab_stride, float *dist_out, int64_t *lab_out, hipStream_t s)
{
    if (nq <= 0 || k <= 0) return;
    const size_t P = next_pow2_host((uint32_t)(nshards * k));
    const size_t shmem = P * sizeof(uint64_t);
    allow_big_lds(merge_topk_kernel, shmem);
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)nq), dim3(SEL_THREADS), shmem, s, nshards, nq,
                       k, dist_in, lab_in, dist_stride, lab_stride, dist_out, lab_out);
}

} // namespace lb
