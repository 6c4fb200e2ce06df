// kernels_select.hip -- batched top-k selection and cross-shard merge (the exact re-rank + proof: kernels_finish.hip).
//
// select:  per query, bitonic-sort the admitted (key,row) entries in LDS, keep the best kc,
//          publish the kc-th entry as the next admission threshold.  The ordering it realises
//          is the canonical form of BruteForceIndex.SearchVectors' bounded heap
//          (internal/store/adaptive_index.go:176-222): ascending (distance, row position).
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
