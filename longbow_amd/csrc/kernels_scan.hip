// kernels_scan.hip -- exact-order distance scan (HBM-bound path), row norms, data fill.
//
// scan_kernel restates, bit for bit, the reference's per-pair arithmetic for a few
// queries against every corpus row:
//   SEQ     referenceEuclidean/referenceCosine (internal/simd/simd_test.go:13-33),
//           cosineGeneric/dotGeneric (internal/simd/simd.go:138-163)
//   UNROLL4 euclideanUnrolled4x/cosineUnrolled4x/dotUnrolled4x (internal/simd/simd.go:365-479)
// i.e. the hot loop of simd.EuclideanDistanceBatchFlat (simd.go:203-229) and of
// BruteForceIndex.SearchVectors (internal/store/adaptive_index.go:180-211).
// Each lane owns one corpus row and carries that row's f32 accumulator chain(s)
// across the D dimension in the reference's order; rows are staged through LDS in
// coalesced 256-B pieces (128 rows x 64 floats per stage; one LDS stage plus a
// register-held prefetch, non-temporal loads) so HBM sees full lines while every lane
// walks its own row.  No FMA contraction here.
//
// Also here: the sampled admission threshold (sample_scores_kernel, sample_tau_kernel,
// sample_topm_kernel; see index.hip: sample_plan) that lets a search walk the corpus once.
#include "lb_device.h"

#pragma clang fp contract(off)

namespace lb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SC_ROWS = 128;   // rows per tile == threads per workgroup
constexpr int SC_DK = 64;      // floats per row per stage
constexpr int SC_LD = SC_DK + 4; // padded LDS row stride (floats): conflict-free ds_read_b128

struct ScanArgs {
    const float *X;
    int64_t row_begin, row_end;
    int D;
    const float *Q;
    const int *qsel;
    int nsel;
    const float *qna; // per selected slot: ||q||^2 in the requested order (cosine)
    const uint8_t *mask;
    const uint32_t *rowmap; // filtered search over a compacted row list: position -> corpus row (or null)
    CandState cs;
    float *all_out;
    int64_t ld;
    int raw_dot;
    int aligned;
    int boot;
    int striped; // admissions go through cs.stripes (slot = position in the launch's slot list)
};

template <int ORDER>
struct Acc {
    float s[ORDER == ORDER_UNROLL4 ? 4 : 1];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int i = 0; i < (ORDER == ORDER_UNROLL4 ? 4 : 1); i++) s[i] = 0.f;
    }
    // t = position within a group of 4 (compile-time); tail elements always use slot 0
    template <int T>
    __device__ __forceinline__ void add(float v)
    {
        if (ORDER == ORDER_UNROLL4) s[T] = s[T] + v;
        else s[0] = s[0] + v;
    }
    __device__ __forceinline__ void add_tail(float v) { s[0] = s[0] + v; }
    __device__ __forceinline__ float total() const
    {
        if (ORDER == ORDER_UNROLL4) {
            float t = s[0] + s[1];
            t = t + s[2];
            t = t + s[3];
            return t;
        }
        return s[0];
    }
};

// MAPPED: positions [row_begin,row_end) index a.rowmap (the visible rows under a filter) instead of the corpus.
// NBUF: LDS stages.  2 = write the next stage while the current one is read (69.6 KB, 2 workgroups/CU);
// 1 = one stage + an extra barrier (34.8 KB, 4 workgroups/CU: twice the bytes in flight per CU).
// The (tile, chunk) sequence of a workgroup is one flat pipeline: the first chunk of the next tile
// is already in flight while the last chunk of the current tile is consumed.
template <int METRIC, int ORDER, int NQ, bool MAPPED, int NBUF>
__global__ __launch_bounds__(SC_ROWS) void scan_kernel(ScanArgs a)
{
    // one __shared__ object: [stage][rows | query chunk]
    constexpr int STAGE_F = SC_ROWS * SC_LD + NQ * SC_DK;
    __shared__ __attribute__((aligned(16))) float lds[NBUF][STAGE_F];
    const int tid = threadIdx.x;
    const int D = a.D;
    const int nchunks = (D + SC_DK - 1) / SC_DK;
    const int dmain = D & ~3; // elements covered by the 4-wide main loop of UNROLL4
    const int64_t nrows = a.row_end - a.row_begin;
    const int64_t ntiles = (nrows + SC_ROWS - 1) / SC_ROWS;
    if ((int64_t)blockIdx.x >= ntiles) return;

    // query slots (wave-uniform)
    int qidx[NQ];
    uint64_t tau[NQ];
#pragma unroll
    for (int j = 0; j < NQ; j++) {
        int jj = j < a.nsel ? j : a.nsel - 1;
        qidx[j] = __builtin_amdgcn_readfirstlane(a.qsel ? a.qsel[jj] : jj);
        tau[j] = (a.all_out == nullptr && j < a.nsel) ? a.cs.tau[qidx[j]] : 0ull;
    }

    // the query chunk rides in the same LDS stage (read back as a broadcast ds_read_b128):
    // keeping Q off the vector-memory queue lets the row prefetch stay in flight under compute.
    const bool q_loader = tid < NQ * 16;
    const float *q_src = nullptr;
    if (q_loader) {
        int j = tid >> 4;
        if (j >= a.nsel) j = a.nsel - 1;
        q_src = a.Q + (int64_t)(a.qsel ? a.qsel[j] : j) * D;
    }

    f32x4 stg[16];
    f32x4 stq = {0.f, 0.f, 0.f, 0.f};
    uint32_t rid[MAPPED ? 16 : 1]; // MAPPED: corpus rows behind this thread's 16 staging slots of the tile being loaded
    auto load_stage = [&](int64_t tile, int c) {
        const int64_t trow0 = a.row_begin + tile * SC_ROWS;
        const int d0 = c * SC_DK;
        if (q_loader) {
            int k = d0 + (tid & 15) * 4;
            if (k > D - 4) k = D - 4;
            stq = *reinterpret_cast<const f32x4 *>(q_src + k);
        }
        if (MAPPED && c == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                int64_t pos = trow0 + ((tid + SC_ROWS * i) >> 4);
                if (pos >= a.row_end) pos = a.row_end - 1;
                rid[MAPPED ? i : 0] = a.rowmap[pos];
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int ch = tid + SC_ROWS * i;
            const int r = ch >> 4, p = ch & 15;
            int64_t row = trow0 + r;
            if (row >= a.row_end) row = a.row_end - 1;
            if (MAPPED) row = rid[MAPPED ? i : 0];
            int k = d0 + p * 4;
            if (k > D - 4) k = D - 4; // D % 4 == 0 here; chunks past D are never consumed
            stg[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.X + row * (int64_t)D + k)); // streamed once
        }
    };
    auto write_stage = [&](int st) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int ch = tid + SC_ROWS * i;
            const int r = ch >> 4, p = ch & 15;
            *reinterpret_cast<f32x4 *>(&lds[st][r * SC_LD + p * 4]) = stg[i];
        }
        if (q_loader)
            *reinterpret_cast<f32x4 *>(&lds[st][SC_ROWS * SC_LD + (tid >> 4) * SC_DK + (tid & 15) * 4]) = stq;
    };

    Acc<ORDER> acc[NQ]; // L2: sum (q-x)^2 ; cos/dot: sum q*x
    Acc<ORDER> nb;      // cos: sum x*x

    // finished tile waiting for admission (done right after the next stage's loads are issued, so the
    // returning atomic's latency overlaps the stream instead of draining it)
    float pend_dist[NQ];
    int64_t pend_tile = -1;
    auto flush = [&]() {
        const int64_t myrow = a.row_begin + pend_tile * SC_ROWS + tid;
        const bool valid = myrow < a.row_end;
        int64_t arow = myrow; // corpus row behind position myrow
        if (MAPPED) arow = a.rowmap[valid ? myrow : a.row_end - 1];
        const bool masked_out = valid && a.all_out == nullptr && a.mask != nullptr && !a.mask[arow];
        if (valid && a.boot && masked_out) {
#pragma unroll
            for (int j = 0; j < NQ; j++)
                if (j < a.nsel) a.cs.lists[(size_t)qidx[j] * a.cs.cap + (myrow - a.row_begin)] = kEntryMax;
        }
        if (valid && !masked_out) {
#pragma unroll
            for (int j = 0; j < NQ; j++) {
                if (j >= a.nsel) break;
                const float dist = pend_dist[j];
                if (a.all_out) {
                    a.all_out[(int64_t)j * a.ld + myrow] = dist;
                } else {
                    const uint64_t ent = pack_entry(dist, (uint32_t)arow);
                    const int qj = qidx[j];
                    if (a.boot) {
                        a.cs.lists[(size_t)qj * a.cs.cap + (myrow - a.row_begin)] = ent;
                    } else if (ent < tau[j]) {
                        // (a returning atomic on one hot counter: measured 6-20 ns apiece in the 1-query
                        // scan, which is why the sampled threshold aims at ~1.2k admissions, not cap/2)
                        uint32_t pos;
                        if (a.striped) {
                            const uint32_t st = blockIdx.x & (LB_STRIPES - 1);
                            pos = st + LB_STRIPES * atomicAdd(&a.cs.stripes[(j * LB_STRIPES + st) * LB_STRIPE_PAD], 1u);
                        } else {
                            pos = atomicAdd(&a.cs.cnt[qj], 1u);
                        }
                        if (pos < a.cs.cap) a.cs.lists[(size_t)qj * a.cs.cap + pos] = ent;
                    }
                }
            }
        }
        pend_tile = -1;
    };

    int64_t tile = blockIdx.x;
    int c = 0, cur = 0;
    load_stage(tile, 0);
    write_stage(0);
    __syncthreads();

    while (true) {
        int64_t ntile = tile;
        int nc = c + 1;
        if (nc == nchunks) {
            nc = 0;
            ntile = tile + gridDim.x;
        }
        const bool has_next = ntile < ntiles;
        if (has_next) load_stage(ntile, nc);
        if (pend_tile >= 0) flush();
        if (c == 0) {
#pragma unroll
            for (int j = 0; j < NQ; j++) acc[j].zero();
            nb.zero();
        }
        {
            const float *xr = &lds[cur][tid * SC_LD];
            const float *lq = &lds[cur][SC_ROWS * SC_LD];
            const int d0 = c * SC_DK;
            const int nfull4 = (min(dmain, d0 + SC_DK) - d0) >> 2; // groups of 4 in the main loop
            // main loop: groups of 4 elements, positions 0..3 -> accumulators 0..3 (UNROLL4)
#pragma unroll 4
            for (int g = 0; g < nfull4; g++) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(&xr[g * 4]);
                if (METRIC == METRIC_COS) {
                    nb.template add<0>(xv.x * xv.x);
                    nb.template add<1>(xv.y * xv.y);
                    nb.template add<2>(xv.z * xv.z);
                    nb.template add<3>(xv.w * xv.w);
                }
#pragma unroll
                for (int j = 0; j < NQ; j++) {
                    const f32x4 qv = *reinterpret_cast<const f32x4 *>(&lq[j * SC_DK + g * 4]);
                    const float q0 = qv.x, q1 = qv.y, q2 = qv.z, q3 = qv.w;
                    if (METRIC == METRIC_L2) {
                        const float e0 = q0 - xv.x, e1 = q1 - xv.y, e2 = q2 - xv.z, e3 = q3 - xv.w;
                        acc[j].template add<0>(e0 * e0);
                        acc[j].template add<1>(e1 * e1);
                        acc[j].template add<2>(e2 * e2);
                        acc[j].template add<3>(e3 * e3);
                    } else {
                        acc[j].template add<0>(q0 * xv.x);
                        acc[j].template add<1>(q1 * xv.y);
                        acc[j].template add<2>(q2 * xv.z);
                        acc[j].template add<3>(q3 * xv.w);
                    }
                }
            }
        }
        if (NBUF == 1) {
            __syncthreads(); // every wave is done reading the single stage
            if (has_next) write_stage(0);
        } else if (has_next) {
            write_stage(cur ^ 1);
        }

        if (c == nchunks - 1) { // the tile's distances are complete: park them, admission runs under the next loads
            const float nbt = nb.total();
#pragma unroll
            for (int j = 0; j < NQ; j++) {
                const float t = acc[j].total();
                float dist;
                if (METRIC == METRIC_L2) {
                    dist = (float)sqrt((double)t);
                } else if (METRIC == METRIC_COS) {
                    const float na = a.qna[j < a.nsel ? j : 0];
                    if (D == 0 || na == 0.0f || nbt == 0.0f) dist = 1.0f;
                    else {
                        const float den = (float)sqrt((double)na * (double)nbt);
                        dist = 1.0f - __fdiv_rn(t, den);
                    }
                } else {
                    dist = a.raw_dot ? t : -t;
                }
                pend_dist[j] = dist;
            }
            pend_tile = tile;
        }
        __syncthreads();
        if (!has_next) break;
        tile = ntile;
        c = nc;
        if (NBUF == 2) cur ^= 1;
    }
    if (pend_tile >= 0) flush();
}


// Generic fallback for D % 4 != 0 (or a misaligned base): one lane per row walks its row
// straight from global memory.  Same arithmetic, no staging; correctness path only.
template <int METRIC, int ORDER>
__global__ __launch_bounds__(256) void scan_generic_kernel(ScanArgs a)
{
    const int D = a.D;
    const int dmain = D & ~3;
    for (int64_t pos = a.row_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pos < a.row_end;
         pos += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = a.rowmap ? (int64_t)a.rowmap[pos] : pos;
        if (a.all_out == nullptr && a.mask != nullptr && !a.mask[row]) {
            if (a.boot)
                for (int j = 0; j < a.nsel; j++)
                    a.cs.lists[(size_t)(a.qsel ? a.qsel[j] : j) * a.cs.cap + (pos - a.row_begin)] = kEntryMax;
            continue;
        }
        const float *x = a.X + row * (int64_t)D;
        for (int j = 0; j < a.nsel; j++) {
            const int qj = a.qsel ? a.qsel[j] : j;
            const float *q = a.Q + (int64_t)qj * D;
            Acc<ORDER> acc, nb;
            acc.zero();
            nb.zero();
            for (int i = 0; i < dmain; i += 4) {
                const float x0 = x[i], x1 = x[i + 1], x2 = x[i + 2], x3 = x[i + 3];
                if (METRIC == METRIC_COS) {
                    nb.template add<0>(x0 * x0);
                    nb.template add<1>(x1 * x1);
                    nb.template add<2>(x2 * x2);
                    nb.template add<3>(x3 * x3);
                }
                if (METRIC == METRIC_L2) {
                    const float e0 = q[i] - x0, e1 = q[i + 1] - x1, e2 = q[i + 2] - x2, e3 = q[i + 3] - x3;
                    acc.template add<0>(e0 * e0);
                    acc.template add<1>(e1 * e1);
                    acc.template add<2>(e2 * e2);
                    acc.template add<3>(e3 * e3);
                } else {
                    acc.template add<0>(q[i] * x0);
                    acc.template add<1>(q[i + 1] * x1);
                    acc.template add<2>(q[i + 2] * x2);
                    acc.template add<3>(q[i + 3] * x3);
                }
            }
            for (int i = dmain; i < D; i++) {
                const float xv = x[i];
                if (METRIC == METRIC_COS) nb.add_tail(xv * xv);
                if (METRIC == METRIC_L2) {
                    const float e = q[i] - xv;
                    acc.add_tail(e * e);
                } else {
                    acc.add_tail(q[i] * xv);
                }
            }
            const float t = acc.total();
            float dist;
            if (METRIC == METRIC_L2) {
                dist = (float)sqrt((double)t);
            } else if (METRIC == METRIC_COS) {
                const float na = a.qna[j], nbt = nb.total();
                if (D == 0 || na == 0.0f || nbt == 0.0f) dist = 1.0f;
                else dist = 1.0f - __fdiv_rn(t, (float)sqrt((double)na * (double)nbt));
            } else {
                dist = a.raw_dot ? t : -t;
            }
            if (a.all_out) {
                a.all_out[(int64_t)j * a.ld + pos] = dist; // indexed by position (== row without a row map)
            } else {
                const uint64_t ent = pack_entry(dist, (uint32_t)row);
                if (a.boot) {
                    a.cs.lists[(size_t)qj * a.cs.cap + (pos - a.row_begin)] = ent;
                } else if (ent < a.cs.tau[qj]) {
                    uint32_t pos;
                    if (a.striped) {
                        const uint32_t st = blockIdx.x & (LB_STRIPES - 1);
                        pos = st + LB_STRIPES * atomicAdd(&a.cs.stripes[(j * LB_STRIPES + st) * LB_STRIPE_PAD], 1u);
                    } else {
                        pos = atomicAdd(&a.cs.cnt[qj], 1u);
                    }
                    if (pos < a.cs.cap) a.cs.lists[(size_t)qj * a.cs.cap + pos] = ent;
                }
            }
        }
    }
}

// ||q||^2 per selected slot in the requested order (cosine only).  One 64-lane block per slot:
// the query is staged in LDS with coalesced loads, then lane 0 runs the reference's f32 chain
// (batched ds_read_b128, no dependent global loads).
template <int ORDER>
__global__ __launch_bounds__(64) void query_norms_kernel(const float *Q, const int *qsel, int nsel, int D, float *qna)
{
    extern __shared__ __attribute__((aligned(16))) float sq[];
    const int j = blockIdx.x;
    if (j >= nsel) return;
    const float *q = Q + (int64_t)(qsel ? qsel[j] : j) * D;
    const int Dpad = (D + 3) & ~3;
    for (int i = threadIdx.x; i < Dpad; i += 64) sq[i] = i < D ? q[i] : 0.f;
    __syncthreads();
    if (threadIdx.x != 0) return;
    Acc<ORDER> a;
    a.zero();
    const int dmain = D & ~3;
#pragma unroll 8
    for (int i = 0; i < dmain; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(&sq[i]);
        a.template add<0>(v.x * v.x);
        a.template add<1>(v.y * v.y);
        a.template add<2>(v.z * v.z);
        a.template add<3>(v.w * v.w);
    }
    for (int i = dmain; i < D; i++) a.add_tail(sq[i] * sq[i]);
    qna[j] = a.total();
}

void launch_query_norms(int order, const float *Q, const int *qsel, int nsel, int D, float *qna,
                        hipStream_t s)
{
    if (nsel <= 0) return;
    dim3 grid(nsel), block(64);
    const size_t shmem = (size_t)((D + 3) & ~3) * sizeof(float);
    if (order == ORDER_UNROLL4)
        hipLaunchKernelGGL(query_norms_kernel<ORDER_UNROLL4>, grid, block, shmem, s, Q, qsel, nsel, D, qna);
    else
        hipLaunchKernelGGL(query_norms_kernel<ORDER_SEQ>, grid, block, shmem, s, Q, qsel, nsel, D, qna);
}

// ---------------------------------------------------------------------------
// Sampled admission threshold (index.hip: sample_plan).  `count` evenly spaced positions of
// [0, span) are scored approximately -- one wave per sampled row, every load of the row in flight
// at once, wave-shuffle reduction: the whole sample costs one HBM round trip instead of the exact
// kernel's D/64 dependent stages -- and the m-th best of them becomes tau.  tau is only a filter
// (every row below it is collected and scored exactly afterwards), so the summation order here
// does not matter; a threshold that admits too few or too many rows is detected and the query is
// redone by the classic schedule.
// ---------------------------------------------------------------------------
constexpr int SS_MAXQ = 8;        // query slots scored together (register accumulators)
constexpr int SS_MAX_SLOTS = 64;  // query slots per launch (groups of SS_MAXQ)

struct SampleArgs {
    const float *X;
    int D;
    int64_t span;
    uint32_t count;
    const uint32_t *rowmap;
    const uint8_t *mask;
    const float *Q;
    const int *qsel;
    int nsel;
    CandState cs;
    int aligned;
    float *qna;       // cosine: exact ||q||^2 per slot, computed by `nsel` extra workgroups of this launch
    int order;
    uint32_t nblocks; // workgroups that score sample rows
    // keys mode (batched path): entries carry the MFMA pipeline's candidate key instead of the distance
    int keys;
    const float *norm2, *rnorm;
    const float *center; // keys mode, L2 over the centred image: (q - c).(x - c) and norm2 = |x - c|^2 (or null)
    // riding along (fp16 route, Qh != null): `nsel` more workgroups prepare a query each (query_prep_body) -- nothing in this
    // launch reads what they write, the candidate pass two launches on does
    _Float16 *Qh = nullptr;
    float *qinv = nullptr, *qnrm = nullptr;
    const float *pcenter = nullptr;
    int tau_zero = 0; // the riding preparation leaves tau = 0 ("not out yet") for a candidate launch that computes its own
    float *qrho = nullptr;
    float rho_gain = 0.f;
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <int ORDER>
__device__ __forceinline__ float exact_sq_norm_lds(const float *sq, int D)
{
    Acc<ORDER> a;
    a.zero();
    const int dmain = D & ~3;
#pragma unroll 8
    for (int i = 0; i < dmain; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(&sq[i]);
        a.template add<0>(v.x * v.x);
        a.template add<1>(v.y * v.y);
        a.template add<2>(v.z * v.z);
        a.template add<3>(v.w * v.w);
    }
    for (int i = dmain; i < D; i++) a.add_tail(sq[i] * sq[i]);
    return a.total();
}

// Everything a batched search over the fp16 route needs from its queries, in one launch (one wave per query): the fp16
// image [D / 32][nq][32] (each query scaled by the power of two that brings its norm into [1, 2); qinv[q] = 1 / scale, exact),
// the exact ||q||^2 in the reference's order (cosine: the re-rank divides by it, internal/simd/simd.go:138-152), and the
// reset of the query's candidate state.  (A zero or non-finite query keeps scale 1: the exact scan answers it anyway.)
// center (or null; L2 over the centred image): the image holds q - center (the exact norm, cosine only, is never asked for then)
// reset: 1 = the query's candidate state (count, threshold, status bits), 2 = the status bits only (the caller's launch sets
// the other two itself, from another workgroup), 0 = nothing.  active: this thread belongs to the ONE wave that does the work
// (every thread of the workgroup makes the call: there is a barrier inside).
__device__ __forceinline__ void query_prep_body(const float *Q, int nq, int D, _Float16 *Qh, float *qinv, float *qna, int order, CandState cs,
                                                const float *center, float *qnrm, int reset, int q, int lane, bool active, float *sq,
                                                float *qrho = nullptr, float rho_gain = 0.f)
{
    const float *src = Q + (int64_t)q * D;
    const int Dpad = (D + 3) & ~3;
    float s = 0.f;
    if (active) {
        // (eight pieces of the row asked for before the first is used: one memory round trip per 512 dimensions instead of
        // one per 64; the sums run in the same order as before)
        constexpr int PF = 8;
        for (int i0 = lane; i0 < Dpad; i0 += 64 * PF) {
            float v[PF], c[PF];
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int i = i0 + 64 * u;
                v[u] = i < D ? src[i] : 0.f;
                c[u] = (center && i < D) ? center[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int i = i0 + 64 * u;
                if (i < Dpad) {
                    const float x = v[u] - c[u]; // (x - 0 = x exactly)
                    sq[i] = x;
                    s += x * x;
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    }
    float scale = 1.f, inv = 1.f;
    if (s > 0.f && s < 3.0e38f) {
        const float nrm = sqrtf(s);
        int ex;
        (void)frexpf(nrm, &ex); // nrm = m * 2^ex, 0.5 <= m < 1  ->  nrm * 2^(1 - ex) in [1, 2)
        int sh = 1 - ex;
        sh = sh < -120 ? -120 : (sh > 120 ? 120 : sh);
        scale = ldexpf(1.f, sh);
        inv = ldexpf(1.f, -sh);
    }
    __syncthreads();
    if (!active) return;
    const int Dp = (D + 31) & ~31; // (dimensions beyond D: zero -- they add nothing to a product)
    float r2 = 0.f; // |scaled q - its fp16 image|^2 (the scaling is a power of two: the ratio to |q|^2 is the unscaled one)
    for (int i = lane; i < Dp; i += 64) {
        const float v = i < D ? sq[i] * scale : 0.f;
        const _Float16 hv = (_Float16)v;
        Qh[((int64_t)(i >> 5) * nq + q) * 32 + (i & 31)] = hv;
        const float d = v - (float)hv;
        r2 = fmaf(d, d, r2);
    }
    if (qrho) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) r2 += __shfl_xor(r2, off);
    }
    if (lane == 0) {
        qinv[q] = inv;
        // (f32 sums of D non-negative terms: relative error below (D + 8) 2^-24 each)
        const float s2 = s * scale * scale;
        const float rho = (s2 > 0.f && s2 < 3.0e38f) ? sqrtf((r2 / s2) * (1.0f + 4.0f * (float)(D + 8) * 5.9604645e-8f)) * 1.000001f : 1.0f;
        if (qrho) qrho[q] = rho;
        // (s is an f32 sum in some order: relative error below (D + 8) 2^-24)
        if (qnrm) qnrm[q] = ((s < 3.0e38f) ? sqrtf(s) * (1.000002f + 1.05f * (float)(D + 8) * 5.9604645e-8f) : s) * (1.0f + rho_gain * rho);
        if (reset == 1) {
            cs.cnt[q] = 0;
            cs.tau[q] = kEntryMax;
        }
        if (reset == 3) cs.tau[q] = 0ull; // "not out yet": the candidate launch computes the thresholds itself (TAUIN)
        if (reset == 1 || reset == 2) cs.flags[q] = 0;
        if (qna) qna[q] = order == ORDER_UNROLL4 ? exact_sq_norm_lds<ORDER_UNROLL4>(sq, D) : exact_sq_norm_lds<ORDER_SEQ>(sq, D);
    }
}

__global__ __launch_bounds__(64) void query_prep_kernel(const float *Q, int nq, int D, _Float16 *Qh, float *qinv, float *qna, int order,
                                                        CandState cs, const float *center, float *qnrm, int tau_zero, float *qrho, float rho_gain)
{
    extern __shared__ __attribute__((aligned(16))) float sq[];
    query_prep_body(Q, nq, D, Qh, qinv, qna, order, cs, center, qnrm, 1, (int)blockIdx.x, (int)threadIdx.x, true, sq, qrho, rho_gain);
    if (tau_zero && threadIdx.x == 0) cs.tau[blockIdx.x] = 0ull; // "not out yet" (TAUIN)
}

void launch_query_prep(const float *Q, int nq, int D, void *Qh, float *qinv, float *qna, int order, CandState cs, hipStream_t s,
                       const float *center, float *qnrm, bool tau_zero, float *qrho, float rho_gain)
{
    if (nq <= 0) return;
    hipLaunchKernelGGL(query_prep_kernel, dim3((unsigned)nq), dim3(64), (size_t)((D + 3) & ~3) * sizeof(float), s, Q, nq, D,
                       reinterpret_cast<_Float16 *>(Qh), qinv, center ? nullptr : qna, order, cs, center, qnrm, tau_zero ? 1 : 0, qrho, rho_gain);
}

// R = sampled rows per wave: every query chunk fetched from L2 is used for R rows (with 32 query
// slots and R = 1 the launch is bound by 8192 x 96 KB of L2 reads: 97 us; R = 4: a quarter of that)
// one wave scores sample rows i0 .. i0 + R against every query slot
template <int METRIC, int R>
__device__ __forceinline__ void sample_wave(const SampleArgs &a, uint32_t i0, int lane)
{
    const int D = a.D;
    int64_t row[R];
    bool live[R], hidden[R];
    const float *x[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t i = i0 + r;
        live[r] = i < a.count;
        if (!live[r]) i = a.count - 1;
        const int64_t pos = (int64_t)(((uint64_t)i * (uint64_t)a.span) / a.count); // i, span < 2^32
        row[r] = a.rowmap ? (int64_t)a.rowmap[pos] : pos;
        hidden[r] = a.mask != nullptr && !a.mask[row[r]];
        x[r] = a.X + row[r] * (int64_t)D;
    }
    float aux[R]; // keys mode: the row's norm term, asked for before the products (one round trip less at the end)
#pragma unroll
    for (int r = 0; r < R; r++) aux[r] = !a.keys ? 0.f : (METRIC == METRIC_L2 ? a.norm2[row[r]] : (METRIC == METRIC_COS ? a.rnorm[row[r]] : 0.f));
    const bool plain_dot = METRIC != METRIC_L2 || a.keys; // L2 keys come from the inner product too
    const bool want_norms = METRIC == METRIC_COS && !a.keys;
    for (int g0 = 0; g0 < a.nsel; g0 += SS_MAXQ) { // groups of 8 query slots; the rows stay hot in L1/L2
        const int gn = a.nsel - g0 < SS_MAXQ ? a.nsel - g0 : SS_MAXQ;
        float acc[SS_MAXQ][R], qq[SS_MAXQ], xx[R];
#pragma unroll
        for (int j = 0; j < SS_MAXQ; j++) {
            qq[j] = 0.f;
#pragma unroll
            for (int r = 0; r < R; r++) acc[j][r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; r++) xx[r] = 0.f;
        if (a.aligned) {
            // the pieces of up to 1024 dimensions are asked for up front: the loop as first written waited for memory once per
            // 256 dimensions (three round trips at 768, ~2 us each on rows nobody has touched)
            constexpr int PF = R == 1 ? 4 : 1; // (several rows a wave: their pieces are the loads in flight, as before)
            for (int k0 = lane * 4; k0 < D; k0 += 256 * PF) {
                f32x4 xw[PF][R];
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    const int k = k0 + 256 * u;
#pragma unroll
                    for (int r = 0; r < R; r++)
                        if (k < D) xw[u][r] = *reinterpret_cast<const f32x4 *>(x[r] + k);
                }
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    const int k = k0 + 256 * u;
                    if (k >= D) break;
                    f32x4 xv[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        xv[r] = xw[u][r];
                        if (a.center) xv[r] = xv[r] - *reinterpret_cast<const f32x4 *>(a.center + k);
                        if (want_norms) xx[r] += xv[r].x * xv[r].x + xv[r].y * xv[r].y + xv[r].z * xv[r].z + xv[r].w * xv[r].w;
                    }
#pragma unroll
                    for (int j = 0; j < SS_MAXQ; j++) {
                        if (j >= gn) break;
                        const int qj = a.qsel ? a.qsel[g0 + j] : g0 + j;
                        f32x4 qv = *reinterpret_cast<const f32x4 *>(a.Q + (int64_t)qj * D + k);
                        if (a.center) qv = qv - *reinterpret_cast<const f32x4 *>(a.center + k);
                        if (want_norms) qq[j] += qv.x * qv.x + qv.y * qv.y + qv.z * qv.z + qv.w * qv.w;
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            if (!plain_dot) {
                                const float e0 = qv.x - xv[r].x, e1 = qv.y - xv[r].y, e2 = qv.z - xv[r].z, e3 = qv.w - xv[r].w;
                                acc[j][r] += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
                            } else {
                                acc[j][r] += qv.x * xv[r].x + qv.y * xv[r].y + qv.z * xv[r].z + qv.w * xv[r].w;
                            }
                        }
                    }
                }
            }
        } else {
            for (int k = lane; k < D; k += 64) {
                float xv[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    xv[r] = x[r][k];
                    if (a.center) xv[r] -= a.center[k];
                    if (want_norms) xx[r] += xv[r] * xv[r];
                }
#pragma unroll
                for (int j = 0; j < SS_MAXQ; j++) {
                    if (j >= gn) break;
                    const int qj = a.qsel ? a.qsel[g0 + j] : g0 + j;
                    float qv = a.Q[(int64_t)qj * D + k];
                    if (a.center) qv -= a.center[k];
                    if (want_norms) qq[j] += qv * qv;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (!plain_dot) {
                            const float e = qv - xv[r];
                            acc[j][r] += e * e;
                        } else {
                            acc[j][r] += qv * xv[r];
                        }
                    }
                }
            }
        }
        if (want_norms) {
#pragma unroll
            for (int r = 0; r < R; r++) xx[r] = wave_sum(xx[r]);
        }
#pragma unroll
        for (int j = 0; j < SS_MAXQ; j++) {
            if (j >= gn) break;
            const int qj = a.qsel ? a.qsel[g0 + j] : g0 + j;
            const float na = want_norms ? wave_sum(qq[j]) : 0.f;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float t = wave_sum(acc[j][r]);
                float v;
                if (a.keys) { // as gemm_filter_kernel's key_of
                    if (METRIC == METRIC_L2) v = fmaf(-2.0f, t, aux[r]);
                    else if (METRIC == METRIC_COS) v = -t * aux[r];
                    else v = -t;
                } else if (METRIC == METRIC_L2) {
                    v = sqrtf(t);
                } else if (METRIC == METRIC_COS) {
                    v = (na == 0.f || xx[r] == 0.f) ? 1.0f : 1.0f - t * rsqrtf(na * xx[r]);
                } else {
                    v = -t;
                }
                if (lane == 0 && live[r]) {
                    const uint64_t ent = hidden[r] ? kEntryMax : pack_entry(v, (uint32_t)row[r]);
                    a.cs.lists[(size_t)qj * a.cs.cap + i0 + r] = ent;
                }
            }
        }
    }
}

template <int METRIC, int R>
__global__ __launch_bounds__(256) void sample_scores_kernel(SampleArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sq[];
    const uint32_t nnorm = a.qna ? (uint32_t)a.nsel : 0u; // the serial norm chains are dispatched first
    if (blockIdx.x < nnorm) { // exact ||q||^2 in the reference's order, off the critical path
        const int j = (int)blockIdx.x;
        const float *q = a.Q + (int64_t)(a.qsel ? a.qsel[j] : j) * a.D;
        const int Dpad = (a.D + 3) & ~3;
        for (int i = threadIdx.x; i < Dpad; i += 256) sq[i] = i < a.D ? q[i] : 0.f;
        __syncthreads();
        if (threadIdx.x == 0)
            a.qna[j] = a.order == ORDER_UNROLL4 ? exact_sq_norm_lds<ORDER_UNROLL4>(sq, a.D) : exact_sq_norm_lds<ORDER_SEQ>(sq, a.D);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nprep = a.Qh ? (uint32_t)a.nsel : 0u;
    if (blockIdx.x < nnorm + nprep) { // (no query subset on this route: slot j is query j)
        query_prep_body(a.Q, a.nsel, a.D, a.Qh, a.qinv, nullptr, a.order, a.cs, a.pcenter, a.qnrm, /*reset=*/a.tau_zero ? 3 : 0,
                        (int)(blockIdx.x - nnorm), lane, wave == 0, sq, a.qrho, a.rho_gain);
        return;
    }
    const uint32_t blk = blockIdx.x - nnorm - nprep;
    if (blk == 0 && (int)threadIdx.x < a.nsel) a.cs.flags[a.qsel ? a.qsel[threadIdx.x] : threadIdx.x] = 0;
    const uint32_t i0 = (blk * 4u + (uint32_t)wave) * R; // this wave's first sample index
    if (i0 >= a.count) return;
    sample_wave<METRIC, R>(a, i0, lane);
}

void launch_sample_scores(int metric, int order, const float *X, int D, int64_t span, uint32_t count,
                          const uint32_t *rowmap, const uint8_t *mask, const float *Q, const int *qsel, int nsel,
                          CandState cs, float *qna, hipStream_t s, const float *norm2, const float *rnorm, const float *center,
                          const SamplePrep *prep)
{
    if (count == 0 || nsel <= 0) return;
    SampleArgs a;
    if (prep && qsel == nullptr && nsel <= SS_MAX_SLOTS) {
        a.Qh = reinterpret_cast<_Float16 *>(prep->Qh);
        a.qinv = prep->qinv;
        a.qnrm = prep->qnrm;
        a.pcenter = prep->center;
        a.tau_zero = prep->tau_zero ? 1 : 0;
        a.qrho = prep->qrho;
        a.rho_gain = prep->rho_gain;
    }
    a.center = (norm2 != nullptr && metric == METRIC_L2) ? center : nullptr;
    a.keys = norm2 != nullptr ? 1 : 0;
    a.norm2 = norm2;
    a.rnorm = rnorm;
    a.qna = (metric == METRIC_COS && !a.keys) ? qna : nullptr;
    a.order = order;
    a.X = X; a.D = D; a.span = span; a.count = count; a.rowmap = rowmap; a.mask = mask;
    a.Q = Q; a.qsel = qsel; a.nsel = nsel < SS_MAX_SLOTS ? nsel : SS_MAX_SLOTS; a.cs = cs;
    a.aligned = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && ((reinterpret_cast<uintptr_t>(Q) & 15) == 0);
    static const int env_r = lb_tunable("LB_SAMPLE_ROWS_PER_WAVE", 0);
    const int R = (env_r == 1 || env_r == 2 || env_r == 4) ? env_r : (a.nsel > SS_MAXQ ? 4 : a.nsel > 4 ? 2 : 1); // measured at 8 slots: 61 -> 53 us for sample + threshold + select
    a.nblocks = (count + 4 * R - 1) / (4 * R);
    dim3 grid(a.nblocks + (a.qna ? (unsigned)a.nsel : 0u) + (a.Qh ? (unsigned)a.nsel : 0u)), block(256);
    const size_t shmem = (a.qna || a.Qh) ? (size_t)((D + 3) & ~3) * sizeof(float) : 0;
#define LB_SS(M)                                                                                      \
    do {                                                                                              \
        if (R == 4) hipLaunchKernelGGL((sample_scores_kernel<M, 4>), grid, block, shmem, s, a);       \
        else if (R == 2) hipLaunchKernelGGL((sample_scores_kernel<M, 2>), grid, block, shmem, s, a);  \
        else hipLaunchKernelGGL((sample_scores_kernel<M, 1>), grid, block, shmem, s, a);              \
    } while (0)
    if (metric == METRIC_L2) LB_SS(METRIC_L2);
    else if (metric == METRIC_COS) LB_SS(METRIC_COS);
    else LB_SS(METRIC_DOT);
#undef LB_SS
}

// tau[q] = m-th smallest of the first `count` entries of list q (row bits saturated), cnt[q] = 0.
// m is small (8..64): pops of a minimum over register-resident, pre-sorted entries beat a radix select here.
// The wave-level minimum runs on DPP lane permutes (no LDS round trips).
constexpr int ST_THREADS = 1024;
constexpr int ST_PER = 8;

// the m-th smallest of list[0 .. count) (kEntryMax if there are fewer visible entries), by a workgroup of ST_THREADS threads;
// the value is returned in wave 0
__device__ __forceinline__ uint64_t sample_tau_body(const uint64_t *list, uint32_t count, int m, uint32_t (*wl)[20], int tid)
{
    // The threshold is a KEY with the row bits saturated: the selection runs on the entries' upper words alone (one DPP
    // minimum per step instead of a 64-bit compare-and-select; equal keys are popped one a round, the first lane that holds
    // one) -- the value is the one the 64-bit selection gave.
    constexpr uint32_t NONE = 0xffffffffu; // (the upper word of kEntryMax: a hidden row, or beyond the sample)
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t e[ST_PER];
#pragma unroll
    for (int i = 0; i < ST_PER; i++) {
        const uint32_t idx = (uint32_t)tid + (uint32_t)ST_THREADS * i;
        const uint64_t ent = idx >= count ? kEntryMax : list[idx];
        e[i] = (uint32_t)(ent >> 32);
    }
    // each thread sorts its 8 keys once (19 compare-exchanges); a round then only looks at the heads
    static_assert(ST_PER == 8, "the sorting network below is for 8 entries");
#define LB_CE(i, j)                         \
    {                                       \
        const uint32_t x = e[i], y = e[j];  \
        e[i] = x < y ? x : y;               \
        e[j] = x < y ? y : x;               \
    }
    LB_CE(0, 1) LB_CE(2, 3) LB_CE(4, 5) LB_CE(6, 7) LB_CE(0, 2) LB_CE(1, 3) LB_CE(4, 6) LB_CE(5, 7) LB_CE(1, 2)
    LB_CE(5, 6) LB_CE(0, 4) LB_CE(3, 7) LB_CE(1, 5) LB_CE(2, 6) LB_CE(1, 4) LB_CE(3, 6) LB_CE(2, 4) LB_CE(3, 5)
    LB_CE(3, 4)
#undef LB_CE
    // Two levels, one barrier.  Level 1: every wave pops its own r smallest keys (wave-wide minimum on DPP lane permutes,
    // no LDS round trip, no barrier) into wl[wave][0 .. r) -- ascending.  Level 2: one wave merges the 16 sorted runs, m pops.
    // r = min(m, 4 + m / 4) < m for m > 5: should one wave hold more than r of the m smallest keys, the merge runs out
    // of that wave's run and returns a LARGER value than the m-th smallest -- a looser threshold, which admits a few more
    // rows and is as valid as the exact one (tau is only a filter; P ~ 1e-4 per search at m = 19).
    const int r1 = m < 4 + m / 4 ? m : 4 + m / 4;
    for (int r = 0; r < r1; r++) {
        const uint32_t v = wave_min_u32(e[0]);
        if (lane == 0) wl[wave][r] = v;
        const uint64_t holders = __builtin_amdgcn_ballot_w64(e[0] == v);
        if (v != NONE && lane == (int)__builtin_ctzll(holders)) { // exactly one lane pops its head
#pragma unroll
            for (int i = 0; i + 1 < ST_PER; i++) e[i] = e[i + 1];
            e[ST_PER - 1] = NONE;
        }
    }
    __syncthreads();
    uint32_t kth = NONE;
    if (wave == 0) {
        static_assert(ST_THREADS / 64 == 16, "one row of 16 lanes merges the per-wave runs");
        int ptr = 0; // lanes 0 .. 15: the head of wave `lane`'s run
        for (int r = 0; r < m; r++) {
            const uint32_t head = (lane < 16 && ptr < r1) ? wl[lane][ptr] : NONE;
            kth = (uint32_t)__builtin_amdgcn_readfirstlane((int)row16_min_u32(head)); // (rows 1 .. 3 of the wave hold NONE)
            if (kth == NONE) break; // fewer than m visible sample rows, or every run used up: no threshold
            const uint64_t holders = __builtin_amdgcn_ballot_w64(head == kth);
            if (lane == (int)__builtin_ctzll(holders)) ptr++;
        }
    }
    return kth == NONE ? kEntryMax : (((uint64_t)kth << 32) | 0xffffffffull);
}

__global__ __launch_bounds__(ST_THREADS) void sample_tau_kernel(CandState cs, const int *qsel, int nsel, uint32_t count, int m,
                                                                int zero_stripes, const float *Q, int D, float *qna, int order)
{
    extern __shared__ __attribute__((aligned(16))) float sq[];
    __shared__ uint32_t wl[ST_THREADS / 64][20]; // (m <= 64: r1 <= 20)
    const int tid = threadIdx.x;
    const int q = qsel ? qsel[blockIdx.x] : blockIdx.x;
    // qna != null (cosine): the exact ||q||^2 of this query rides along -- its D dependent additions (3.5 us at 768) are done by
    // one lane of wave 1 while wave 0 merges the runs; the row is staged here, the selection's barrier publishes it.  (Rider
    // workgroups of their own doubled the launch's workgroups: 1024 queries 25 -> 15 us.)
    if (qna) {
        const float *qr = Q + (int64_t)q * D;
        const int Dpad = (D + 3) & ~3;
        for (int i = tid; i < Dpad; i += ST_THREADS) sq[i] = i < D ? qr[i] : 0.f;
    }
    const uint64_t kth = sample_tau_body(cs.lists + (size_t)q * cs.cap, count, m, wl, tid);
    if (tid == 0) {
        cs.tau[q] = kth;
        cs.cnt[q] = 0;
    }
    if (qna && tid == 64) qna[blockIdx.x] = order == ORDER_UNROLL4 ? exact_sq_norm_lds<ORDER_UNROLL4>(sq, D) : exact_sq_norm_lds<ORDER_SEQ>(sq, D);
    if (zero_stripes && tid < LB_STRIPES) cs.stripes[(blockIdx.x * LB_STRIPES + tid) * LB_STRIPE_PAD] = 0;
}

// Level 1 of a two-level threshold for samples larger than one list (PQ at 100M rows samples 390k
// codes): workgroup g leaves the m smallest of in[g*8192 .. +8192) at lists[slot][g*m .. +m) (padded
// with kEntryMax); sample_tau_kernel over those G*m entries then yields the exact m-th smallest of
// the whole sample.
__global__ __launch_bounds__(ST_THREADS) void sample_topm_kernel(const uint64_t *in, uint32_t count_total, int m,
                                                                 CandState cs, int slot)
{
    __shared__ uint64_t wmin[2][ST_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t base = blockIdx.x * (uint32_t)(ST_THREADS * ST_PER);
    uint64_t e[ST_PER];
#pragma unroll
    for (int i = 0; i < ST_PER; i++) {
        const uint32_t idx = base + (uint32_t)tid + (uint32_t)ST_THREADS * i;
        e[i] = idx < count_total ? in[idx] : kEntryMax;
    }
#define LB_CE(i, j)                         \
    {                                       \
        const uint64_t x = e[i], y = e[j];  \
        e[i] = x < y ? x : y;               \
        e[j] = x < y ? y : x;               \
    }
    LB_CE(0, 1) LB_CE(2, 3) LB_CE(4, 5) LB_CE(6, 7) LB_CE(0, 2) LB_CE(1, 3) LB_CE(4, 6) LB_CE(5, 7) LB_CE(1, 2)
    LB_CE(5, 6) LB_CE(0, 4) LB_CE(3, 7) LB_CE(1, 5) LB_CE(2, 6) LB_CE(1, 4) LB_CE(3, 6) LB_CE(2, 4) LB_CE(3, 5)
    LB_CE(3, 4)
#undef LB_CE
    uint64_t *out = cs.lists + (size_t)slot * cs.cap + (size_t)blockIdx.x * m;
    for (int r = 0; r < m; r++) {
        uint64_t v = wave_min_u64(e[0]);
        if (lane == 0) wmin[r & 1][wave] = v;
        __syncthreads();
        v = row16_min_u64(wmin[r & 1][lane & 15]);
        if (tid == 0) out[r] = v; // kEntryMax once the group is exhausted
        if (v != kEntryMax && e[0] == v) {
#pragma unroll
            for (int i = 0; i + 1 < ST_PER; i++) e[i] = e[i + 1];
            e[ST_PER - 1] = kEntryMax;
        }
    }
}

// returns the number of level-1 groups (entries for level 2 = groups * m), 0 if it does not fit the list
uint32_t launch_sample_topm(const uint64_t *in, uint32_t count_total, int m, CandState cs, int slot, hipStream_t s)
{
    const uint32_t per = (uint32_t)(ST_THREADS * ST_PER);
    const uint32_t groups = (count_total + per - 1) / per;
    if (groups == 0 || (uint64_t)groups * (uint64_t)m > (uint64_t)per || (uint64_t)groups * (uint64_t)m > cs.cap) return 0;
    hipLaunchKernelGGL(sample_topm_kernel, dim3(groups), dim3(ST_THREADS), 0, s, in, count_total, m, cs, slot);
    return groups;
}

bool sample_tau_supported(uint32_t count, int m) { return count <= (uint32_t)(ST_THREADS * ST_PER) && m >= 1 && m <= 64; }

void launch_sample_tau(CandState cs, const int *qsel, int nsel, uint32_t count, int m, bool zero_stripes, hipStream_t s,
                       const float *Q, int D, float *qna, int order)
{
    if (nsel <= 0) return;
    const size_t shmem = qna ? (size_t)((D + 3) & ~3) * sizeof(float) : 0;
    hipLaunchKernelGGL(sample_tau_kernel, dim3(nsel), dim3(ST_THREADS), shmem, s, cs, qsel, nsel, count,
                       m, (zero_stripes && cs.stripes != nullptr) ? 1 : 0, Q, D, qna, order);
}

int g_scan_nbuf = lb_tunable("LB_SCAN_NBUF", 1);

template <int METRIC, int ORDER, int NQ>
static void launch_scan_variant(dim3 grid, hipStream_t s, const ScanArgs &a)
{
    if (g_scan_nbuf == 1) {
        if (a.rowmap) hipLaunchKernelGGL((scan_kernel<METRIC, ORDER, NQ, true, 1>), grid, dim3(SC_ROWS), 0, s, a);
        else hipLaunchKernelGGL((scan_kernel<METRIC, ORDER, NQ, false, 1>), grid, dim3(SC_ROWS), 0, s, a);
    } else {
        if (a.rowmap) hipLaunchKernelGGL((scan_kernel<METRIC, ORDER, NQ, true, 2>), grid, dim3(SC_ROWS), 0, s, a);
        else hipLaunchKernelGGL((scan_kernel<METRIC, ORDER, NQ, false, 2>), grid, dim3(SC_ROWS), 0, s, a);
    }
}

template <int METRIC, int ORDER>
static void launch_scan_nq(int nq_t, dim3 grid, hipStream_t s, const ScanArgs &a)
{
    switch (nq_t) {
    case 1: launch_scan_variant<METRIC, ORDER, 1>(grid, s, a); break;
    case 2: launch_scan_variant<METRIC, ORDER, 2>(grid, s, a); break;
    case 4: launch_scan_variant<METRIC, ORDER, 4>(grid, s, a); break;
    default: launch_scan_variant<METRIC, ORDER, 8>(grid, s, a); break;
    }
}

void launch_scan(int metric, int order, bool raw_dot, const float *X, int64_t row_begin,
                      int64_t row_end, int D, const float *Q, const int *qsel, int nsel,
                      const float *qna, const uint8_t *mask, const uint32_t *rowmap, CandState cs, bool boot,
                      float *all_out, int64_t ld, hipStream_t s, bool striped)
{
    if (row_end <= row_begin || nsel <= 0) return;
    ScanArgs a;
    a.striped = (striped && cs.stripes != nullptr && !boot) ? 1 : 0;
    a.rowmap = rowmap;
    a.boot = boot ? 1 : 0;
    a.X = X; a.row_begin = row_begin; a.row_end = row_end; a.D = D;
    a.Q = Q; a.qsel = qsel; a.nsel = nsel; a.qna = qna; a.mask = mask; a.cs = cs;
    a.all_out = all_out; a.ld = ld; a.raw_dot = raw_dot ? 1 : 0;
    a.aligned = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    if (!a.aligned) {
        int64_t blocks = (row_end - row_begin + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        dim3 g((unsigned)blocks), b(256);
#define LB_GEN(M)                                                                          \
    do {                                                                                   \
        if (order == ORDER_UNROLL4) hipLaunchKernelGGL((scan_generic_kernel<M, ORDER_UNROLL4>), g, b, 0, s, a); \
        else hipLaunchKernelGGL((scan_generic_kernel<M, ORDER_SEQ>), g, b, 0, s, a);        \
    } while (0)
        if (metric == METRIC_L2) LB_GEN(METRIC_L2);
        else if (metric == METRIC_COS) LB_GEN(METRIC_COS);
        else LB_GEN(METRIC_DOT);
#undef LB_GEN
        return;
    }
    const int64_t ntiles = (row_end - row_begin + SC_ROWS - 1) / SC_ROWS;
    // 69.6 KB (2 stages) / 34.8 KB (1 stage) LDS per workgroup -> 2 / 4 workgroups per CU; 256 CUs.
    static const int waves_mult = lb_tunable("LB_SCAN_GRIDMULT", 4);
    const int64_t maxgrid = 256 * (g_scan_nbuf == 1 ? 4 : 2) * waves_mult;
    dim3 grid((unsigned)(ntiles < maxgrid ? ntiles : maxgrid));
    int nq_t = nsel <= 1 ? 1 : nsel <= 2 ? 2 : nsel <= 4 ? 4 : 8;
#define LB_SCAN(M)                                                                       \
    do {                                                                                 \
        if (order == ORDER_UNROLL4) launch_scan_nq<M, ORDER_UNROLL4>(nq_t, grid, s, a);   \
        else launch_scan_nq<M, ORDER_SEQ>(nq_t, grid, s, a);                              \
    } while (0)
    if (metric == METRIC_L2) LB_SCAN(METRIC_L2);
    else if (metric == METRIC_COS) LB_SCAN(METRIC_COS);
    else LB_SCAN(METRIC_DOT);
#undef LB_SCAN
}

// ---------------------------------------------------------------------------
// Row norms at Add time: one wave per row, coalesced float4 reads, wave reduction.
// Used only for candidate keys and error bounds (never for reported distances).
// ---------------------------------------------------------------------------
// center (or null): the norms of x - center (the L2 keys over the centred fp16 image, index.hip: sync_f16_image); rnorm may be null
__global__ __launch_bounds__(256) void row_norms_kernel(const float *X, int64_t n, int D,
                                                        float *norm2, float *rnorm,
                                                        uint32_t *maxnorm2, int aligned, const float *center)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *x = X + row * (int64_t)D;
    float s = 0.f;
    if (center) {
        for (int i = lane; i < D; i += 64) {
            const float v = x[i] - center[i];
            s += v * v;
        }
    } else if (aligned) {
        for (int i = lane * 4; i < D; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + i);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    } else {
        for (int i = lane; i < D; i += 64) s += x[i] * x[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        norm2[row] = s;
        if (rnorm) rnorm[row] = s > 0.f ? (float)(1.0 / sqrt((double)s)) : 0.f;
        // s >= 0: uint order == float order.  Read first: one contended atomic per row would
        // serialise the whole kernel on a single address.
        const uint32_t bits = __builtin_bit_cast(uint32_t, s);
        if (bits > __hip_atomic_load(maxnorm2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxnorm2, bits);
        // [1]: the smallest non-zero norm (zero rows are exact under every contraction: they do not count)
        if (s > 0.f && bits < __hip_atomic_load(maxnorm2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(maxnorm2 + 1, bits);
    }
}

void launch_row_norms(const float *X, int64_t n, int D, float *norm2, float *rnorm,
                      uint32_t *d_maxnorm2, hipStream_t s, const float *center)
{
    if (n <= 0) return;
    const int aligned = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    dim3 grid((unsigned)((n + 3) / 4));
    hipLaunchKernelGGL(row_norms_kernel, grid, dim3(256), 0, s, X, n, D, norm2, rnorm, d_maxnorm2,
                       aligned, center);
}

// Column means of X[0 .. n) in a fixed summation order (two stages: 256 partial sums per column, then their sum): the centre
// the L2 keys of the fp16 image are taken about.  Any fixed vector would do -- L2 distances do not move when both sides are
// shifted -- the mean makes the shifted norms, and with them the key errors, as small as a shift can.
__global__ __launch_bounds__(256) void column_sums_kernel(const float *X, int64_t n, int D, float *partial)
{
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
    for (int c = threadIdx.x; c < D; c += 256) {
        float acc = 0.f;
        for (int64_t r = r0; r < r1; r++) acc += X[r * (int64_t)D + c];
        partial[(int64_t)blockIdx.x * D + c] = acc;
    }
}
__global__ __launch_bounds__(256) void column_mean_kernel(const float *partial, int nparts, int D, int64_t n, float *center, int Dpad)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Dpad) return;
    float acc = 0.f;
    if (c < D)
        for (int p = 0; p < nparts; p++) acc += partial[(int64_t)p * D + c];
    center[c] = c < D ? acc / (float)n : 0.f;
}
void launch_column_means(const float *X, int64_t n, int D, float *partial /* [256][D] */, float *center /* [Dpad] */, int Dpad,
                         hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(column_sums_kernel, dim3(256), dim3(256), 0, s, X, n, D, partial);
    hipLaunchKernelGGL(column_mean_kernel, dim3((unsigned)((Dpad + 255) / 256)), dim3(256), 0, s, partial, 256, D, n, center, Dpad);
}

// ---------------------------------------------------------------------------
// Synthetic data: counter-based splitmix64 generator (the CPU checker restates the same formula).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed + (idx + 1) * 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

__global__ void fill_uniform_kernel(float *dst, int64_t n, uint64_t seed, int64_t offset)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (float)(splitmix64_at(seed, (uint64_t)(offset + i)) >> 40) * (1.0f / 16777216.0f);
}

__global__ void fill_codes_kernel(uint8_t *dst, int64_t n, uint64_t seed, int64_t offset)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (uint8_t)(splitmix64_at(seed, (uint64_t)(offset + i)) >> 56);
}

__global__ void fill_uniform_rows_kernel(float *dst, const int64_t *ids, int64_t nrows, int dim, uint64_t seed)
{
    const int64_t total = nrows * (int64_t)dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dim;
        const int j = (int)(i - r * dim);
        dst[i] = (float)(splitmix64_at(seed, (uint64_t)(ids[r] * (int64_t)dim + j)) >> 40) * (1.0f / 16777216.0f);
    }
}

void launch_fill_uniform_rows(float *dst, const int64_t *ids, int64_t nrows, int dim, uint64_t seed, hipStream_t s)
{
    if (nrows <= 0 || dim <= 0) return;
    int64_t blocks = (nrows * (int64_t)dim + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fill_uniform_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, ids, nrows, dim, seed);
}

void launch_fill_uniform(float *dst, int64_t n, uint64_t seed, int64_t offset, hipStream_t s)
{
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, n, seed, offset);
}

void launch_fill_codes(uint8_t *dst, int64_t n, uint64_t seed, int64_t offset, hipStream_t s)
{
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fill_codes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, n, seed, offset);
}

} // namespace lb
