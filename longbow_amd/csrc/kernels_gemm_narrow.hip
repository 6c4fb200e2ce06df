// kernels_gemm_narrow.hip -- candidate generation for SMALL and MID-SIZE query batches (5..384 queries).
//
// Same contract as gemm_filter_kernel (kernels_gemm.hip) with a tile shaped for the HBM-bound
// regime: 256 corpus rows x 32 queries per workgroup.  At B <= 32 the f32 contraction needs
// 2*32*N*D flop = 0.31 ms of MFMA time at 1M x 768 while the corpus read needs ~0.5 ms of HBM time,
// so one pass over the corpus serves the whole batch at the HBM rate; the 128-query tile of the wide
// kernel would spend 4x the MFMA time on padding.  4 waves, each 64 rows x 32 queries (2 MFMA
// 32x32 tiles, 32 accumulator VGPRs); each wave stages exactly the 64 corpus rows it consumes
// (8 direct-to-LDS DMA instructions of 1 KiB per K-step, nt policy) plus a quarter of the 4 KiB query
// tile; LDS 2 x (32 + 4) KiB, two workgroups per CU = 64 KiB of corpus bytes in flight per CU.
// LDS image, swizzle, k permutation and the admission test (key, branch-free) are those of the wide
// kernel; admitted entries go through a workgroup-local LDS list and out with one returning global
// atomic per query of the tile (the wide kernel's one-atomic-per-lane epilogue stalls a wave once per
// lane with admissions -- fine under 41 us of MFMAs per tile, not under a 66 us HBM-bound one).
//
// The kernel is a template over the tile: <256 rows, 32 queries> (above) and <128 rows, 64 queries>
// for batches of 33..~200 queries (each wave 32 rows x 64 queries, again 2 MFMA tiles; LDS
// 2 x (16 + 8) KiB, three workgroups per CU), which would otherwise pay a second corpus pass per
// extra 32 queries; over SPLIT (the contraction on 3 x bf16 MFMA with both operands split in
// registers: the default) and over FUSED (5..32 queries: the sampled threshold of the search rides
// inside the launch -- sample tiles, per-query threshold workgroups, corpus workgroups of two row
// tiles that pick the thresholds up; see FusedSample in lb_device.h and LABNOTES.md 3.3).
#include "lb_device.h"

namespace lb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x (8 consecutive f32 of one row) -> hi = bf16(x) (round to nearest even), lo = bf16(x - hi):
// x = hi + lo + O(2^-18 |x|).  Same split as split_bf16_kernel (kernels_gemm.hip), done in registers.
__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, bf16x8 &hi, bf16x8 &lo)
{
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = h;
        lo[i] = (__bf16)(x[i] - (float)h);
    }
}

constexpr int NBK = 32;
constexpr int NTHREADS = 256;

struct NarrowArgs {
    const float *X;
    const float *norm2;
    const float *rnorm;
    int64_t row_begin, row_end;
    int D;
    const float *Q;
    int nq;
    const uint8_t *mask;
    const uint32_t *rowmap; // position -> corpus row for filtered search over a compacted list (or null)
    CandState cs;
    int n_row_tiles, n_q_tiles;
    int boot;
    FusedSample fs; // FUSED only
};

__device__ __forceinline__ int nswz(int row, int chunk) { return row * NBK + ((chunk ^ ((row >> 1) & 7)) << 2); }

#ifdef LB_DIAG
// timing probe of the fused launch (100 MHz real-time ticks): [0] first entry stamp (min), [1] last threshold published
// (max), [2] sum of the corpus workgroups' waits for thresholds, [3] corpus workgroups, [4] those that had to wait,
// [5] last sample workgroup's keys out (max), [6] last threshold workgroup has its keys loaded, [7] ... its rounds done
__device__ unsigned long long g_fused_probe[8];
void read_fused_probe(unsigned long long out[8], bool reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_probe), 8 * sizeof(unsigned long long));
    if (reset) {
        unsigned long long z[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fused_probe), z, sizeof z);
    }
}
#endif
// ---- helpers of the fused sample (FUSED) ---------------------------------------------------------------------
constexpr uint32_t kSpinLimit = 1500; // x ~0.6 us (s_sleep 8 + one L2 round trip) ~ 1 ms: a wait that long means the launch's workgroups are
                                      // not co-resident; the batch is then redone on the exact path and the give-up is counted
                                      // (lb_gpu_index_fused_giveups)

// ||q||^2 in the reference's accumulation order (as kernels_scan.hip: exact_sq_norm_lds), q in LDS, one lane
__device__ __forceinline__ float narrow_exact_sq_norm(const float *sq, int D, int order)
{
#pragma clang fp contract(off)
    if (order == ORDER_UNROLL4) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const int dmain = D & ~3;
        for (int i = 0; i < dmain; i += 4) {
            s0 = s0 + sq[i] * sq[i];
            s1 = s1 + sq[i + 1] * sq[i + 1];
            s2 = s2 + sq[i + 2] * sq[i + 2];
            s3 = s3 + sq[i + 3] * sq[i + 3];
        }
        for (int i = dmain; i < D; i++) s0 = s0 + sq[i] * sq[i];
        float t = s0 + s1;
        t = t + s2;
        t = t + s3;
        return t;
    }
    float t = 0.f;
    for (int i = 0; i < D; i++) t = t + sq[i] * sq[i];
    return t;
}

// SPLIT: the inner products are computed as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with both
// operands split into bf16 pairs IN REGISTERS after the (unchanged) f32 LDS staging -- no second copy of
// the corpus.  3/16 of the f32 MFMA cycles: at 32-64 queries the f32 contraction keeps the MFMA pipe 70 %
// busy under the corpus stream (0.31 of 0.45 ms per pass at 1M x 768), which is what held this kernel at
// 5.1 TB/s; split, the pipe is ~13 % busy and the kernel is a pure HBM stream.  Candidate keys carry the
// split contraction's error bound (index.hip: gamma); reported results come from the exact re-rank.
// FUSED: the launch starts with a.fs.n_blocks workgroups that score the sampled rows and publish the thresholds (see
// FusedSample in lb_device.h); the corpus tiles fetch their thresholds in front of the epilogue instead of at entry.
// The sample workgroups have the lowest block ids, so they are resident before any workgroup that waits for them.
template <int METRIC, int NBM, int NBN, bool SPLIT, bool FUSED = false> // NBM corpus rows x NBN queries per tile: <256, 32> or <128, 64>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_filter_narrow_kernel(NarrowArgs a)
{
    constexpr int WROWS = NBM / 4;   // corpus rows per wave
    constexpr int TM = WROWS / 32;   // MFMA row tiles per wave
    constexpr int TN = NBN / 32;     // MFMA query tiles per wave
    constexpr int NA = WROWS / 8;    // A DMA instructions per wave and K-step (8 rows x 128 B each)
    constexpr int NB = NBN / 32;     // B DMA instructions per wave and K-step
    // FUSED: a corpus workgroup contracts TWO row tiles (a second set of 32 accumulator registers) before its epilogue:
    // the thresholds of the launch's own sample are then long published when the first epilogues need them, and a
    // launch has half as many pipeline fills and drains.
    constexpr int TPW = FUSED ? 2 : 1;
#ifdef LB_DIAG
    if (FUSED && threadIdx.x == 0) atomicMin(&g_fused_probe[0], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    static_assert(TM * TN == 2, "two MFMA tiles (32 accumulator VGPRs) per wave");
    int b = blockIdx.x;
    const bool is_sample = FUSED && b < (int)a.fs.n_blocks;
    if (FUSED && !is_sample) b -= (int)a.fs.n_blocks;
    int qt, rt_first, ntiles_here;
    if (is_sample) { // sample role: positions [rt * NBM, +NBM) of the sample list, bootstrap semantics
        qt = b % a.n_q_tiles;
        rt_first = b / a.n_q_tiles;
        ntiles_here = 1;
    } else {
        const int xcd = b & 7;
        const int in_xcd = b >> 3;
        qt = in_xcd % a.n_q_tiles;
        rt_first = ((in_xcd / a.n_q_tiles) * 8 + xcd) * TPW;
        if (rt_first >= a.n_row_tiles) return;
        ntiles_here = a.n_row_tiles - rt_first < TPW ? a.n_row_tiles - rt_first : TPW;
    }
    const uint32_t *const rowmap = is_sample ? a.fs.smap : a.rowmap;
    const int64_t row_begin = is_sample ? 0 : a.row_begin;
    const int64_t row_end = is_sample ? (int64_t)a.fs.count : a.row_end;
    const bool boot = is_sample || a.boot != 0;

    constexpr int STAGE_F = (NBM + NBN) * NBK;
    __shared__ __attribute__((aligned(16))) float lds_all[2 * STAGE_F + TPW * (NBM + NBM + NBM / 4)];
    float *s_aux = lds_all + 2 * STAGE_F;                                   // [TPW][NBM]
    uint32_t *s_rowid = reinterpret_cast<uint32_t *>(s_aux + TPW * NBM);    // [TPW][NBM]
    uint8_t *s_vis = reinterpret_cast<uint8_t *>(s_rowid + TPW * NBM);      // [TPW][NBM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int q0 = qt * NBN;
    const int64_t last_row = row_end - 1;
    const int last_q = a.nq - 1;
    const int nk = a.D / NBK; // D % 32 == 0 (launcher)

    auto corpus_row = [&](int64_t pos) -> int64_t {
        if (pos > last_row) pos = last_row;
        return rowmap ? (int64_t)rowmap[pos] : pos;
    };
    // B DMA sources (the query tile is the same for every row tile of this workgroup): instruction i of this wave fills
    // query rows 8*NB*wave + 8i .. +7; lane l lands at (row l/8, chunk position l%8)
    const float *srcB[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int row = (wave * NB + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int qr = q0 + row;
        if (qr > last_q) qr = last_q;
        srcB[i] = a.Q + (int64_t)qr * a.D + 4 * c;
    }
    float tk[TN];
    uint32_t tr[TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int qj = q0 + tn * 32 + l31;
        uint64_t tau = (boot || FUSED) ? 0ull : a.cs.tau[qj < a.nq ? qj : a.nq - 1]; // (FUSED: fetched after the loops)
        if (qj >= a.nq) tau = 0ull;
        tk[tn] = tau_key_of(tau);
        tr[tn] = entry_row(tau);
    }

    // A DMA sources of every row tile of this workgroup: instruction i of this wave fills rows WROWS*wave + 8i .. +7 (the
    // rows the wave consumes).  The tiles form ONE flat pipeline: the last K-step of a tile already requests the first
    // stage of the next one, and the side inputs of all tiles are fetched up front behind the first request.
    const float *srcA[TPW][NA];
#pragma unroll
    for (int tp = 0; tp < TPW; tp++)
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int row = wave * WROWS + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int64_t row0 = row_begin + (int64_t)(rt_first + (tp < ntiles_here ? tp : 0)) * NBM;
            srcA[tp][i] = a.X + corpus_row(row0 + row) * (int64_t)a.D + 4 * c;
        }
    auto stage_in = [&](const float *const (&src)[NA], int stage, int k0) {
        float *A = lds_all + stage * STAGE_F;
        float *B = A + NBM * NBK;
#pragma unroll
        for (int i = 0; i < NA; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i] + k0),
                                             (__attribute__((address_space(3))) void *)(A + (wave * WROWS + i * 8) * NBK),
                                             16, 0, 2 /* nt: the corpus streams through once */);
#pragma unroll
        for (int i = 0; i < NB; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcB[i] + k0),
                                             (__attribute__((address_space(3))) void *)(B + (wave * NB + i) * 8 * NBK), 16, 0, 0);
    };
    stage_in(srcA[0], 0, 0);
    // one burst behind the DMA (not needed before the epilogue): the tiles' side inputs
#pragma unroll
    for (int tp = 0; tp < TPW; tp++) {
        if (tp >= ntiles_here) continue; // (workgroup-uniform)
        const int64_t row0 = row_begin + (int64_t)(rt_first + tp) * NBM;
        const int64_t side_ri = corpus_row(row0 + (tid & (NBM - 1)));
        const float side_aux = METRIC == METRIC_L2 ? a.norm2[side_ri] : (METRIC == METRIC_COS ? a.rnorm[side_ri] : 0.f);
        uint8_t side_vis = 1;
        if (a.mask) side_vis = a.mask[side_ri];
        if (tid < NBM) {
            s_aux[tp * NBM + tid] = side_aux;
            s_vis[tp * NBM + tid] = (row0 + tid <= last_row && side_vis) ? (uint8_t)1 : (uint8_t)0;
            s_rowid[tp * NBM + tid] = (uint32_t)side_ri;
        }
    }
    __syncthreads();

    uint32_t spec_ready = 0;
    uint64_t spec_tau = 0ull;
    f32x16 acc[TPW][TM][TN];
#pragma unroll
    for (int tp = 0; tp < TPW; tp++) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[tp][i][j][r] = 0.f;
        if (tp >= ntiles_here) continue; // (workgroup-uniform)
        for (int kt = 0; kt < nk; kt++) {
            const int cur = (tp * nk + kt) & 1;
            if (FUSED && !is_sample && tp == ntiles_here - 1 && tid < NBN) {
                // early look at this workgroup's thresholds (a K-loop ahead of their use): published long ago for all but
                // the first workgroups of a launch, and the two dependent device-scope loads then cost nothing at the end
                const int qj = q0 + tid < a.nq ? q0 + tid : a.nq - 1;
                // (the flag is read with ACQUIRE: pairs with the release store of the threshold workgroup)
#ifdef LB_DIAG
                if (kt == 0 && a.fs.relaxed) spec_ready = __hip_atomic_load(&a.fs.ready[qj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
#endif
                if (kt == 0) spec_ready = __hip_atomic_load(&a.fs.ready[qj], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (kt == 3 && spec_ready == a.fs.epoch)
                    spec_tau = __hip_atomic_load(&a.cs.tau[qj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (kt + 1 < nk) stage_in(srcA[tp], cur ^ 1, (kt + 1) * NBK);
            else if (tp + 1 < ntiles_here) stage_in(srcA[tp + 1 < TPW ? tp + 1 : tp], cur ^ 1, 0);
            const float *As = lds_all + cur * STAGE_F;
            const float *Bs = As + NBM * NBK;
            if (SPLIT) {
                // MFMA k-step ks covers floats [16 ks, 16 ks + 16) of the 32-float K-step; lane half h supplies
                // k = 8h .. 8h+7 of it: two 16-B chunks (4 ks + 2h, 4 ks + 2h + 1) of the row's 128-B piece
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const int ch = 4 * ks + 2 * h;
                    bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                    for (int t = 0; t < TM; t++) {
                        const int r = wave * WROWS + t * 32 + l31;
                        split8(*reinterpret_cast<const f32x4 *>(&As[nswz(r, ch)]), *reinterpret_cast<const f32x4 *>(&As[nswz(r, ch + 1)]),
                               ah[t], al[t]);
                    }
#pragma unroll
                    for (int t = 0; t < TN; t++) {
                        const int r = t * 32 + l31;
                        split8(*reinterpret_cast<const f32x4 *>(&Bs[nswz(r, ch)]), *reinterpret_cast<const f32x4 *>(&Bs[nswz(r, ch + 1)]),
                               bh[t], bl[t]);
                    }
#pragma unroll
                    for (int tm = 0; tm < TM; tm++)
#pragma unroll
                        for (int tn = 0; tn < TN; tn++) {
                            acc[tp][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tp][tm][tn], 0, 0, 0);
                            acc[tp][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tp][tm][tn], 0, 0, 0);
                            acc[tp][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tp][tm][tn], 0, 0, 0);
                        }
                }
            } else {
                f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
                for (int t = 0; t < TM; t++) fa[0][t] = *reinterpret_cast<const f32x4 *>(&As[nswz(wave * WROWS + t * 32 + l31, h)]);
#pragma unroll
                for (int t = 0; t < TN; t++) fb[0][t] = *reinterpret_cast<const f32x4 *>(&Bs[nswz(t * 32 + l31, h)]);
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int cb = s & 1, nb = cb ^ 1;
                    if (s < 3) {
                        const int ch = 2 * (s + 1) + h;
#pragma unroll
                        for (int t = 0; t < TM; t++)
                            fa[nb][t] = *reinterpret_cast<const f32x4 *>(&As[nswz(wave * WROWS + t * 32 + l31, ch)]);
#pragma unroll
                        for (int t = 0; t < TN; t++) fb[nb][t] = *reinterpret_cast<const f32x4 *>(&Bs[nswz(t * 32 + l31, ch)]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++)
#pragma unroll
                        for (int tm = 0; tm < TM; tm++)
#pragma unroll
                            for (int tn = 0; tn < TN; tn++)
                                acc[tp][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][tm][e], fb[cb][tn][e], acc[tp][tm][tn], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }

    // workgroup-local admission list (every non-bootstrap launch), carved from the (now free) stages behind s_tau
    constexpr int FL_CAP = 2048;
    uint32_t *s_lcnt = reinterpret_cast<uint32_t *>(lds_all) + 256;     // entries in the list
    uint32_t *s_qcnt = s_lcnt + 1;                                      // [NBN] of them per query of the tile ...
    uint32_t *s_qbase = s_qcnt + NBN;                                   // [NBN] ... and where they start in the query's list
    uint64_t *s_lent = reinterpret_cast<uint64_t *>(lds_all + 512);
    uint16_t *s_lq = reinterpret_cast<uint16_t *>(lds_all + 512 + 2 * FL_CAP);
    uint16_t *s_lr = s_lq + FL_CAP;                                     // rank of the entry among its query's
    if (FUSED && !is_sample) { // thresholds published by the sample workgroups of this launch
        uint64_t *s_tau = reinterpret_cast<uint64_t *>(lds_all); // (the stages are free: the loops ended with a barrier)
        if (tid == 0) *s_lcnt = 0;
        if (tid < NBN) s_qcnt[tid] = 0;
#ifdef LB_DIAG
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            atomicAdd(&g_fused_probe[3], 1ull);
            if (__hip_atomic_load(&a.fs.ready[q0 < a.nq ? q0 : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.fs.epoch)
                atomicAdd(&g_fused_probe[4], 1ull);
        }
#endif
        if (tid < NBN) {
            const int qj = q0 + tid;
            uint64_t tau = 0ull; // nothing passes
            if (qj < a.nq && spec_ready == a.fs.epoch && nk > 3) {
                tau = spec_tau;
            } else if (qj < a.nq) {
                bool ok = false;
                for (uint32_t it = 0; it < kSpinLimit; it++) { // relaxed polls (acquire polls cost 2-3x per hop) ...
                    if (__hip_atomic_load(&a.fs.ready[qj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.fs.epoch) {
                        ok = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                // ... and ONE acquire once the flag has matched, in front of the payload load
                if (ok) ok = __hip_atomic_load(&a.fs.ready[qj], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == a.fs.epoch;
                if (ok) tau = __hip_atomic_load(&a.cs.tau[qj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *a.fs.fail_host = a.fs.epoch; // no threshold in time: the host redoes the batch on the exact path
            }
            s_tau[tid] = tau;
        }
        __syncthreads();
#ifdef LB_DIAG
        if (tid == 0) atomicAdd(&g_fused_probe[2], (unsigned long long)__builtin_amdgcn_s_memrealtime() - w0);
#endif
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
            const uint64_t tau = s_tau[tn * 32 + l31];
            tk[tn] = tau_key_of(tau);
            tr[tn] = entry_row(tau);
        }
    } else if (!boot) {
        if (tid == 0) *s_lcnt = 0;
        if (tid < NBN) s_qcnt[tid] = 0;
        __syncthreads();
    }

    // ---- epilogue (as in gemm_filter_kernel), one row tile after the other --------------------------
    auto key_of = [&](float dot, float ax) -> float {
        if (METRIC == METRIC_L2) return fmaf(-2.0f, dot, ax);
        if (METRIC == METRIC_COS) return -dot * ax;
        return -dot;
    };
#pragma unroll
    for (int tp = 0; tp < TPW; tp++) {
        if (tp >= ntiles_here) continue;
        const int64_t row0 = row_begin + (int64_t)(rt_first + tp) * NBM;
        float aux[TM][4][4];
        uint32_t rid[TM][4][4];
        uint32_t vbits = 0; // bit (tm*16 + g*4 + e)
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int lr = tp * NBM + wave * WROWS + tm * 32 + 8 * g + 4 * h;
                const f32x4 av = *reinterpret_cast<const f32x4 *>(&s_aux[lr]);
                const uint4 rv = *reinterpret_cast<const uint4 *>(&s_rowid[lr]);
                rid[tm][g][0] = rv.x; rid[tm][g][1] = rv.y; rid[tm][g][2] = rv.z; rid[tm][g][3] = rv.w;
                const uint32_t vv = *reinterpret_cast<const uint32_t *>(&s_vis[lr]);
                aux[tm][g][0] = av.x; aux[tm][g][1] = av.y; aux[tm][g][2] = av.z; aux[tm][g][3] = av.w;
                const uint32_t nib = (vv & 1u) | ((vv >> 7) & 2u) | ((vv >> 14) & 4u) | ((vv >> 21) & 8u);
                vbits |= nib << (tm * 16 + g * 4);
            }
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
            const int qj = q0 + tn * 32 + l31;
            const bool qok = qj < a.nq;
            uint64_t *list = a.cs.lists + (size_t)(qok ? qj : 0) * a.cs.cap;
            if (boot) {
                if (qok) {
#pragma unroll
                    for (int tm = 0; tm < TM; tm++)
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const int64_t rbase = row0 + wave * WROWS + tm * 32 + 8 * g + 4 * h;
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                if (rbase + e < row_end) {
                                    const uint64_t ent = ((vbits >> (tm * 16 + g * 4 + e)) & 1u)
                                                             ? pack_entry(key_of(acc[tp][tm][tn][4 * g + e], aux[tm][g][e]), rid[tm][g][e])
                                                             : kEntryMax;
                                    if (FUSED) // read by another workgroup of this launch: device-scope store
                                        __hip_atomic_store(&list[rbase + e - row_begin], ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    else
                                        list[rbase + e - row_begin] = ent;
                                }
                        }
                }
                continue;
            }
            uint32_t bits = 0;
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float key = key_of(acc[tp][tm][tn][4 * g + e], aux[tm][g][e]);
                        const uint32_t ri = rid[tm][g][e];
                        const uint32_t lt = (uint32_t)(key < tk[tn]) | ((uint32_t)(key == tk[tn]) & (uint32_t)(ri < tr[tn]));
                        bits |= lt << (tm * 16 + g * 4 + e);
                    }
            bits &= vbits; // out-of-range rows and queries never pass (tau of a padded query decodes to NaN)
            if (bits) {
                // admissions go to a workgroup-local list first (one LDS atomic per lane) and out to the per-query lists
                // at the very end, every entry's returning global atomic in flight at once: at 32 queries a wave otherwise
                // sits through ~7 of those round trips, one after the other, per pair of tiles
                const uint32_t n = (uint32_t)__builtin_popcount(bits);
                uint32_t lp = atomicAdd(s_lcnt, n);
                if (lp + n <= (uint32_t)FL_CAP) {
                    uint32_t lr = atomicAdd(&s_qcnt[qj - q0], n);
#pragma unroll
                    for (int tm = 0; tm < TM; tm++)
#pragma unroll
                        for (int g = 0; g < 4; g++)
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                if (bits & (1u << (tm * 16 + g * 4 + e))) {
                                    s_lent[lp] = pack_entry(key_of(acc[tp][tm][tn][4 * g + e], aux[tm][g][e]), rid[tm][g][e]);
                                    s_lq[lp] = (uint16_t)(qj - q0);
                                    s_lr[lp] = (uint16_t)lr;
                                    lp++;
                                    lr++;
                                }
                    bits = 0; // done
                } else {
                    for (uint32_t i = lp; i < lp + n && i < (uint32_t)FL_CAP; i++) s_lent[i] = kEntryMax; // reserved, unused
                }
            }
            if (bits) {
                uint32_t pos = atomicAdd(&a.cs.cnt[qj], (uint32_t)__builtin_popcount(bits));
#pragma unroll
                for (int tm = 0; tm < TM; tm++)
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (bits & (1u << (tm * 16 + g * 4 + e))) {
                                const uint32_t ri = rid[tm][g][e];
                                if (pos < a.cs.cap) list[pos] = pack_entry(key_of(acc[tp][tm][tn][4 * g + e], aux[tm][g][e]), ri);
                                pos++;
                            }
            }
        }
    }
    if (!boot) { // flush the workgroup-local admissions: ONE returning global atomic per query of the tile, all in flight
        __syncthreads();
        if (tid < NBN) {
            const uint32_t n = s_qcnt[tid];
            s_qbase[tid] = n ? atomicAdd(&a.cs.cnt[q0 + tid], n) : 0u; // (n != 0 implies a real query)
        }
        __syncthreads();
        const uint32_t total = *s_lcnt < (uint32_t)FL_CAP ? *s_lcnt : (uint32_t)FL_CAP;
        for (uint32_t i = tid; i < total; i += NTHREADS) {
            const uint64_t ent = s_lent[i];
            if (ent == kEntryMax) continue;
            const int ql = (int)s_lq[i];
            const uint32_t pos = s_qbase[ql] + (uint32_t)s_lr[i];
            if (pos < a.cs.cap) a.cs.lists[(size_t)(q0 + ql) * a.cs.cap + pos] = ent;
        }
        return;
    }
    if (!is_sample) return;

    // ---- sample role, second act: thresholds --------------------------------------------------------------------
    // every sample workgroup reports in (its keys are out: device-scope stores, acknowledged); workgroup j < nq then
    // waits for all of them and turns query j's keys into tau[j]
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) atomicAdd(a.fs.ticket, 1u);
#ifdef LB_DIAG
    if (tid == 0) atomicMax(&g_fused_probe[5], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    const int j = (int)blockIdx.x;
    if (j >= a.nq) return;
    uint32_t *s_flag = reinterpret_cast<uint32_t *>(lds_all);
    if (tid == 0) a.cs.flags[j] = 0; // (nothing in this launch sets status bits; select / re-rank run behind it)
    if (a.fs.qna) { // cosine: exact ||q_j||^2 for the re-rank -- while the slower sample workgroups finish
        float *sq = lds_all + 64;
        const float *q = a.Q + (int64_t)j * a.D;
        for (int i = tid; i < a.D; i += NTHREADS) sq[i] = q[i];
        __syncthreads();
        if (tid == 0) a.fs.qna[j] = narrow_exact_sq_norm(sq, a.D, a.fs.order);
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t ok = 0;
        for (uint32_t it = 0; it < kSpinLimit; it++) {
            if (__hip_atomic_load(a.fs.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.fs.ticket_base >= a.fs.n_blocks) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        s_flag[0] = ok;
    }
    __syncthreads();
    const bool all_in = s_flag[0] != 0;
    __syncthreads();
    uint64_t kth = 0ull; // (not all in: nothing passes, the batch is redone)
    if (all_in) {
        // m-th smallest of the `count` keys (m <= 32 rounds of a workgroup-wide minimum over register-resident entries)
        constexpr int PER = 8192 / NTHREADS;
        const uint64_t *klist = a.cs.lists + (size_t)j * a.cs.cap;
        uint64_t e[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint32_t idx = (uint32_t)tid + (uint32_t)NTHREADS * i;
            e[i] = idx < a.fs.count ? __hip_atomic_load(&klist[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kEntryMax;
        }
#ifdef LB_DIAG
        if (tid == 0 && e[0] != 0) atomicMax(&g_fused_probe[6], (unsigned long long)__builtin_amdgcn_s_memrealtime()); // keys loaded
#endif
        // two levels, no workgroup barrier inside the rounds: each wave extracts the m smallest of ITS 2048 entries (m rounds
        // of a wave-wide minimum on DPP), then wave 0 extracts the m-th smallest of the 4 m survivors
        uint64_t *s_top = reinterpret_cast<uint64_t *>(lds_all) + 8; // [4][32]
        const int m = a.fs.m; // <= 32
        for (int r = 0; r < m; r++) {
            uint64_t v = e[0];
#pragma unroll
            for (int i = 1; i < PER; i++) v = e[i] < v ? e[i] : v;
            v = wave_min_u64(v);
            if (lane == 0) s_top[wave * 32 + r] = v;
            if (v != kEntryMax) {
#pragma unroll
                for (int i = 0; i < PER; i++)
                    if (e[i] == v) e[i] = kEntryMax; // entries are unique
            }
        }
        __syncthreads();
        kth = kEntryMax;
        if (wave == 0) {
            uint64_t c0 = lane < m ? s_top[lane] : kEntryMax, c1 = lane < m ? s_top[32 + lane] : kEntryMax;       // (lanes 0..31:
            uint64_t c2 = lane < m ? s_top[64 + lane] : kEntryMax, c3 = lane < m ? s_top[96 + lane] : kEntryMax;  //  4 entries each)
            if (lane >= 32) c0 = c1 = c2 = c3 = kEntryMax;
            for (int r = 0; r < m; r++) {
                uint64_t v = c0 < c1 ? c0 : c1;
                const uint64_t w2 = c2 < c3 ? c2 : c3;
                v = w2 < v ? w2 : v;
                v = wave_min_u64(v);
                kth = v;
                if (v == kEntryMax) break; // fewer than m visible sample rows: no threshold
                if (c0 == v) c0 = kEntryMax;
                if (c1 == v) c1 = kEntryMax;
                if (c2 == v) c2 = kEntryMax;
                if (c3 == v) c3 = kEntryMax;
            }
            if (lane == 0) s_top[0] = kth;
        }
        __syncthreads();
        kth = s_top[0];
        if (kth != kEntryMax) kth |= 0xffffffffull; // row bits saturated, as sample_tau_kernel
#ifdef LB_DIAG
        if (tid == 0) atomicMax(&g_fused_probe[7], (unsigned long long)__builtin_amdgcn_s_memrealtime()); // rounds done
#endif
    }
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(&a.cs.cnt[j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.cs.tau[j], kth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!all_in) *a.fs.fail_host = a.fs.epoch;
        // publish: cnt / tau above, then ready[j] with RELEASE semantics at agent scope; the waiters read ready[j] with
        // ACQUIRE.  (The asm wait stays: ROCm 7.2 can drop the release fence's own vmcnt wait when the scoreboard looks
        // empty to it -- MI355X_MICROARCH.md "Compiler hazard".)
#ifdef LB_DIAG
        if (a.fs.relaxed) { // A/B only: what the release costs
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&a.fs.ready[j], a.fs.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
#endif
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&a.fs.ready[j], a.fs.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef LB_DIAG
        atomicMax(&g_fused_probe[1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    }
}

// Requires D % 32 == 0 and 16-B aligned X / Q (the caller checks; otherwise the wide kernel runs).
void launch_gemm_filter_narrow(int metric, const float *X, const float *norm2, const float *rnorm,
                               int64_t row_begin, int64_t row_end, int D, const float *Q, int nq,
                               const uint8_t *mask, const uint32_t *rowmap, CandState cs, bool boot,
                               hipStream_t s, bool tile64, bool split)
{
    if (row_end <= row_begin || nq <= 0) return;
    NarrowArgs a;
    a.rowmap = rowmap;
    a.X = X; a.norm2 = norm2; a.rnorm = rnorm; a.row_begin = row_begin; a.row_end = row_end; a.D = D;
    a.Q = Q; a.nq = nq; a.mask = mask; a.cs = cs; a.boot = boot ? 1 : 0;
    const int bm = tile64 ? 128 : 256, bn = tile64 ? 64 : 32;
    a.n_row_tiles = (int)((row_end - row_begin + bm - 1) / bm);
    a.n_q_tiles = (nq + bn - 1) / bn;
    const int groups = (a.n_row_tiles + 7) / 8;
    dim3 grid((unsigned)(groups * 8 * a.n_q_tiles));
#define LB_NARROW_S(M, SP)                                                                                          \
    do {                                                                                                            \
        if (tile64) hipLaunchKernelGGL((gemm_filter_narrow_kernel<M, 128, 64, SP>), grid, dim3(NTHREADS), 0, s, a); \
        else hipLaunchKernelGGL((gemm_filter_narrow_kernel<M, 256, 32, SP>), grid, dim3(NTHREADS), 0, s, a);        \
    } while (0)
#define LB_NARROW(M)                        \
    do {                                    \
        if (split) LB_NARROW_S(M, true);    \
        else LB_NARROW_S(M, false);         \
    } while (0)
    if (metric == METRIC_L2) LB_NARROW(METRIC_L2);
    else if (metric == METRIC_COS) LB_NARROW(METRIC_COS);
    else LB_NARROW(METRIC_DOT);
#undef LB_NARROW
#undef LB_NARROW_S
}

uint32_t fused_sample_blocks(uint32_t count, int nq, bool tile64)
{
    const int bm = tile64 ? 128 : 256, bn = tile64 ? 64 : 32;
    return ((count + (uint32_t)bm - 1) / (uint32_t)bm) * (uint32_t)((nq + bn - 1) / bn);
}

// One launch: sampled threshold + candidate pass over [row_begin, row_end) (split contraction; see FusedSample).
void launch_gemm_filter_narrow_fused(int metric, const float *X, const float *norm2, const float *rnorm, int64_t row_begin,
                                     int64_t row_end, int D, const float *Q, int nq, const uint8_t *mask,
                                     const uint32_t *rowmap, CandState cs, hipStream_t s, bool tile64, FusedSample fs)
{
    if (row_end <= row_begin || nq <= 0) return;
    NarrowArgs a;
    a.rowmap = rowmap;
    a.X = X; a.norm2 = norm2; a.rnorm = rnorm; a.row_begin = row_begin; a.row_end = row_end; a.D = D;
    a.Q = Q; a.nq = nq; a.mask = mask; a.cs = cs; a.boot = 0;
    const int bm = tile64 ? 128 : 256, bn = tile64 ? 64 : 32;
    a.n_row_tiles = (int)((row_end - row_begin + bm - 1) / bm);
    a.n_q_tiles = (nq + bn - 1) / bn;
    fs.n_blocks = fused_sample_blocks(fs.count, nq, tile64);
    a.fs = fs;
    const int groups = ((a.n_row_tiles + 1) / 2 + 7) / 8; // a corpus workgroup contracts two row tiles
    dim3 grid((unsigned)(fs.n_blocks + (uint32_t)(groups * 8 * a.n_q_tiles)));
#define LB_NARROW_F(M)                                                                                                    \
    do {                                                                                                                  \
        if (tile64) hipLaunchKernelGGL((gemm_filter_narrow_kernel<M, 128, 64, true, true>), grid, dim3(NTHREADS), 0, s, a); \
        else hipLaunchKernelGGL((gemm_filter_narrow_kernel<M, 256, 32, true, true>), grid, dim3(NTHREADS), 0, s, a);        \
    } while (0)
    if (metric == METRIC_L2) LB_NARROW_F(METRIC_L2);
    else if (metric == METRIC_COS) LB_NARROW_F(METRIC_COS);
    else LB_NARROW_F(METRIC_DOT);
#undef LB_NARROW_F
}

} // namespace lb
