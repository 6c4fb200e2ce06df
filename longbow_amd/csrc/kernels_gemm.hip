// kernels_gemm.hip -- candidate generation for batched k-NN on gfx950.
//
// The one dense contraction of the path: S = X_tile . Q_tile^T on the f32 MFMA
// (v_mfma_f32_32x32x2_f32, exact f32 fma chain), with the metric key and the
// top-k admission test fused into the epilogue so the nq x N score matrix never
// reaches HBM.  Replaces the per-pair loops of simd.EuclideanDistanceBatchFlat /
// CosineDistanceBatch / DotProductBatch (internal/simd/batch_operations.go:64-157)
// when many queries are in flight; ranking semantics of
// BruteForceIndex.SearchVectors (internal/store/adaptive_index.go:161-225).
//
// Tile: 128 corpus rows x 128 queries per workgroup, BK = 32 floats (one 128-B
// line per row per step), 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles.
// LDS: [128][32] f32 per operand per stage, 16-B chunks XOR-swizzled by
// (row>>1)&7 so the ds_read_b128 fragment reads are bank-conflict free.
#include "lb_device.h"

namespace lb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128; // corpus rows per tile (MFMA A operand, output rows)
constexpr int BN = 128; // queries per tile   (MFMA B operand, output cols = lanes)
constexpr int BK = 32;
constexpr int GEMM_THREADS = 256;

struct GemmArgs {
    const float *X;
    const float *norm2;
    const float *rnorm;
    int64_t row_begin, row_end;
    int D;
    const float *Q;
    int nq;
    const uint8_t *mask;
    CandState cs;
    int n_row_tiles, n_q_tiles;
    int boot; // bootstrap chunk: store every row at list[row - row_begin], no test, no atomics
};

__device__ __forceinline__ int swz_off(int row, int chunk)
{
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2);
}

// MODE 2: D % 32 == 0 and 16-B aligned rows -> unconditional 16-B loads
// MODE 1: D % 4 == 0 and aligned -> 16-B loads, chunks past D read as zero
// MODE 0: anything else -> guarded scalar loads
template <int MODE>
__device__ __forceinline__ f32x4 load_chunk(const float *base, int64_t row, int D, int k)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float *p = base + row * (int64_t)D + k;
    if (MODE == 2) {
        v = *reinterpret_cast<const f32x4 *>(p);
    } else if (MODE == 1) {
        if (k < D) v = *reinterpret_cast<const f32x4 *>(p);
    } else {
        if (k + 0 < D) v.x = p[0];
        if (k + 1 < D) v.y = p[1];
        if (k + 2 < D) v.z = p[2];
        if (k + 3 < D) v.w = p[3];
    }
    return v;
}

template <int METRIC, int ALIGNED>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_filter_kernel(GemmArgs a)
{
    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so the
    // n_q_tiles query tiles of one corpus tile are issued back-to-back on ONE XCD and the
    // corpus tile is pulled from HBM into that XCD's L2 once.  Placement only affects speed.
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int in_xcd = b >> 3;
    const int qt = in_xcd % a.n_q_tiles;
    const int rt = (in_xcd / a.n_q_tiles) * 8 + xcd;
    if (rt >= a.n_row_tiles) return;

    __shared__ __attribute__((aligned(16))) float lds[2][2][BM * BK]; // [stage][A|B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const int64_t row0 = a.row_begin + (int64_t)rt * BM;
    const int q0 = qt * BN;
    const int64_t last_row = a.row_end - 1;
    const int last_q = a.nq - 1;

    // staging assignment: 4 chunks of 16 B per operand per thread
    int st_row[4], st_ch[4];
    int64_t st_xrow[4];
    int st_qrow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int c = tid + GEMM_THREADS * i;
        st_row[i] = c >> 3;
        st_ch[i] = c & 7;
        int64_t xr = row0 + st_row[i];
        st_xrow[i] = xr < last_row ? xr : last_row;
        int qr = q0 + st_row[i];
        st_qrow[i] = qr < last_q ? qr : last_q;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nk = (a.D + BK - 1) / BK;
    f32x4 ra[4], rb[4];

    // prologue: stage 0
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ra[i] = load_chunk<ALIGNED>(a.X, st_xrow[i], a.D, st_ch[i] * 4);
        rb[i] = load_chunk<ALIGNED>(a.Q, st_qrow[i], a.D, st_ch[i] * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        *reinterpret_cast<f32x4 *>(&lds[0][0][swz_off(st_row[i], st_ch[i])]) = ra[i];
        *reinterpret_cast<f32x4 *>(&lds[0][1][swz_off(st_row[i], st_ch[i])]) = rb[i];
    }
    __syncthreads();

    // Main loop.  Per K-step (BK = 32): the next stage's global loads are issued first, the
    // fragment reads of sub-step s+1 are issued before the 16 MFMAs of sub-step s (register
    // double buffer), and the LDS write of the next stage happens in the middle of the MFMA
    // stream (its target buffer was released by the barrier that ended the previous K-step), so
    // the only serial section left at the end of a K-step is the barrier itself.
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        const bool has_next = kt + 1 < nk;
        if (has_next) {
            const int k0 = (kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ra[i] = load_chunk<ALIGNED>(a.X, st_xrow[i], a.D, k0 + st_ch[i] * 4);
                rb[i] = load_chunk<ALIGNED>(a.Q, st_qrow[i], a.D, k0 + st_ch[i] * 4);
            }
        }
        const float *As = lds[cur][0];
        const float *Bs = lds[cur][1];
        f32x4 fa[2][2], fb[2][2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            fa[0][t] = *reinterpret_cast<const f32x4 *>(&As[swz_off(wr * 64 + t * 32 + l31, h)]);
            fb[0][t] = *reinterpret_cast<const f32x4 *>(&Bs[swz_off(wc * 64 + t * 32 + l31, h)]);
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int cb = s & 1, nb = cb ^ 1;
            if (s < 3) {
                const int ch = 2 * (s + 1) + h; // the two lane halves take alternate 16-B chunks;
                                                // the same k permutation is applied to A and B.
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    fa[nb][t] = *reinterpret_cast<const f32x4 *>(&As[swz_off(wr * 64 + t * 32 + l31, ch)]);
                    fb[nb][t] = *reinterpret_cast<const f32x4 *>(&Bs[swz_off(wc * 64 + t * 32 + l31, ch)]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][tm][e], fb[cb][tn][e],
                                                                          acc[tm][tn], 0, 0, 0);
            if (s == 1 && has_next) {
                const int nxt = cur ^ 1;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    *reinterpret_cast<f32x4 *>(&lds[nxt][0][swz_off(st_row[i], st_ch[i])]) = ra[i];
                    *reinterpret_cast<f32x4 *>(&lds[nxt][1][swz_off(st_row[i], st_ch[i])]) = rb[i];
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: key + admission -------------------------------------------------
    // C layout (32x32): col = lane&31 (query), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int qj = q0 + wc * 64 + tn * 32 + l31;
        const bool qok = qj < a.nq;
        const uint64_t tau = qok ? a.cs.tau[qj] : 0ull;
        uint64_t *list = a.cs.lists + (size_t)(qok ? qj : 0) * a.cs.cap;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t rbase = row0 + wr * 64 + tm * 32 + 8 * g + 4 * h;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int64_t ri = rbase + e;
                    if (ri >= a.row_end) continue;
                    const float dot = acc[tm][tn][4 * g + e];
                    float key;
                    if (METRIC == METRIC_L2) key = fmaf(-2.0f, dot, a.norm2[ri]);
                    else if (METRIC == METRIC_COS) key = -dot * a.rnorm[ri];
                    else key = -dot;
                    const uint64_t ent = pack_entry(key, (uint32_t)ri);
                    if (a.boot) {
                        if (qok) list[ri - a.row_begin] = (a.mask && !a.mask[ri]) ? kEntryMax : ent;
                    } else if (ent < tau) {
                        if (a.mask && !a.mask[ri]) continue;
                        uint32_t pos = atomicAdd(&a.cs.cnt[qj], 1u);
                        if (pos < a.cs.cap) list[pos] = ent;
                    }
                }
            }
        }
    }
}

void launch_gemm_filter(int metric, const float *X, const float *norm2, const float *rnorm,
                        int64_t row_begin, int64_t row_end, int D, const float *Q, int nq,
                        const uint8_t *mask, CandState cs, bool boot, hipStream_t s)
{
    if (row_end <= row_begin || nq <= 0) return;
    GemmArgs a;
    a.boot = boot ? 1 : 0;
    a.X = X; a.norm2 = norm2; a.rnorm = rnorm;
    a.row_begin = row_begin; a.row_end = row_end; a.D = D;
    a.Q = Q; a.nq = nq; a.mask = mask; a.cs = cs;
    a.n_row_tiles = (int)((row_end - row_begin + BM - 1) / BM);
    a.n_q_tiles = (nq + BN - 1) / BN;
    const int groups = (a.n_row_tiles + 7) / 8;
    dim3 grid((unsigned)(groups * 8 * a.n_q_tiles));
    const bool aligned = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(Q) & 15) == 0);
    const int mode = !aligned ? 0 : (D % BK == 0 ? 2 : 1);
#define LB_GEMM(M, AL) hipLaunchKernelGGL((gemm_filter_kernel<M, AL>), grid, dim3(GEMM_THREADS), 0, s, a)
#define LB_GEMM_M(M)                 \
    do {                             \
        if (mode == 2) LB_GEMM(M, 2); \
        else if (mode == 1) LB_GEMM(M, 1); \
        else LB_GEMM(M, 0);           \
    } while (0)
    if (metric == METRIC_L2) LB_GEMM_M(METRIC_L2);
    else if (metric == METRIC_COS) LB_GEMM_M(METRIC_COS);
    else LB_GEMM_M(METRIC_DOT);
#undef LB_GEMM_M
#undef LB_GEMM
}

} // namespace lb
