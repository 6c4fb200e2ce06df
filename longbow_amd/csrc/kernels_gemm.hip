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

#include <cstdlib>

namespace lb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128; // corpus rows per tile (MFMA A operand, output rows)
constexpr int BN = 128; // queries per tile   (MFMA B operand, output cols = lanes)
constexpr int BK = 32;
constexpr int GEMM_THREADS = 256;

__device__ unsigned long long g_clock_probe[8]; // profiling aid (ABL == 5 only)

struct GemmArgs {
    const float *X;
    const float *norm2;
    const float *rnorm;
    int64_t row_begin, row_end;
    int D;
    const float *Q;
    int nq;
    const uint8_t *mask;
    const uint32_t *rowmap; // filtered search over a compacted row list: position -> corpus row (or null)
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

// ABL: timing-only ablations for profiling (results are wrong unless ABL == 0):
//   1 = no barriers, 2 = no global loads / LDS writes in the loop, 3 = no fragment reads in the loop,
//   4 = MFMA only (1+2+3), 5 = normal + clock stamps (shader cycles vs 100 MHz real time),
//   6 = no epilogue at all, 7 = epilogue pass 1 only (no atomics / stores)
// GLDS: stage tiles with direct-to-LDS DMA loads (global_load_lds_dwordx4; needs ALIGNED == 2).
// The LDS image is lane-linear per wave instruction (base + lane*16), so the chunk swizzle is
// applied to the per-lane SOURCE address; the image is identical to the register-staged one.
// SPLIT: X and Q are the split-bf16 images produced by split_bf16_kernel: per row and group of 16
// k, 32 B of bf16 "hi" values followed by 32 B of bf16 "lo" values (x ~ hi + lo, |x-hi-lo| <=
// 2^-18|x|).  The inner product is then hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (exact
// bf16 products, f32 accumulation): ~f32-accurate candidate keys at 3/16 of the f32 MFMA cycles.
// Staging, LDS image and epilogue are shared with the f32 kernel (same bytes per row and K-step).
template <int METRIC, int ALIGNED, int ABL = 0, bool GLDS = false, int SPLIT = 0>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_filter_kernel(GemmArgs a)
{
    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so the
    // n_q_tiles query tiles of one corpus tile are issued back-to-back on ONE XCD and the
    // corpus tile is pulled from HBM into that XCD's L2 once.  Placement only affects speed.
    uint64_t stamp_entry = 0;
    if (ABL == 5) stamp_entry = __builtin_amdgcn_s_memtime();
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int in_xcd = b >> 3;
    const int qt = in_xcd % a.n_q_tiles;
    const int rt = (in_xcd / a.n_q_tiles) * 8 + xcd;
    if (rt >= a.n_row_tiles) return;

    // one LDS object: [stage][A|B][BM*BK] f32 tiles, then the tile's per-row side inputs
    // (norm / 1/norm and the predicate byte), fetched once at kernel entry
    __shared__ __attribute__((aligned(16))) float lds_all[2 * 2 * BM * BK + BM + BM + BM / 4];
    float(*lds)[2][BM * BK] = reinterpret_cast<float(*)[2][BM * BK]>(lds_all);
    float *s_aux = lds_all + 2 * 2 * BM * BK;
    uint32_t *s_rowid = reinterpret_cast<uint32_t *>(s_aux + BM); // corpus row of each tile row
    uint8_t *s_vis = reinterpret_cast<uint8_t *>(s_rowid + BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const int64_t row0 = a.row_begin + (int64_t)rt * BM;
    const int q0 = qt * BN;
    const int64_t last_row = a.row_end - 1;
    const int last_q = a.nq - 1;

    auto corpus_row = [&](int64_t pos) -> int64_t { // positions index the (possibly compacted) row list
        if (pos > last_row) pos = last_row;
        return a.rowmap ? (int64_t)a.rowmap[pos] : pos;
    };
    // (the tile's side inputs and thresholds are fetched AFTER the first stage's DMA has been issued, see
    // below: they are not needed before the epilogue and would otherwise put two to three serial memory
    // round trips in front of the first K-step)
    const int64_t side_ri = corpus_row(row0 + (tid & (BM - 1)));

    // staging assignment: 4 chunks of 16 B per operand per thread
    int st_row[4], st_ch[4];
    int64_t st_xrow[4];
    int st_qrow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int c = tid + GEMM_THREADS * i;
        st_row[i] = c >> 3;
        st_ch[i] = c & 7;
        st_xrow[i] = corpus_row(row0 + st_row[i]);
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

    // GLDS staging: wave w issues instructions 4w..4w+3 per operand; instruction j fills LDS rows
    // 8j..8j+7 (1 KiB); lane l lands at (row 8j + l/8, chunk position l%8) and therefore fetches
    // source chunk (l%8) ^ ((row>>1)&7) of that row.
    const float *gsrcA[4], *gsrcB[4];
    if (GLDS) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = wave * 4 + i;
            const int row = j * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int64_t xr = corpus_row(row0 + row);
            int qr = q0 + row;
            if (qr > last_q) qr = last_q;
            gsrcA[i] = a.X + xr * (int64_t)a.D + 4 * c;
            gsrcB[i] = a.Q + (int64_t)qr * a.D + 4 * c;
        }
    }
    auto glds_stage = [&](int stage, int k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = wave * 4 + i;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(gsrcA[i] + k0),
                (__attribute__((address_space(3))) void *)(&lds[stage][0][j * 8 * BK]), 16, 0,
                (ABL == 8) ? 0 : 2); // aux 2 = nt: the corpus streams through once; keeps Q resident in L2
                                     // (measured: L2-miss traffic 2.5x -> 1.8x algorithmic, same speed)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(gsrcB[i] + k0),
                (__attribute__((address_space(3))) void *)(&lds[stage][1][j * 8 * BK]), 16, 0, 0);
        }
    };

    // prologue: stage 0
    if (GLDS) glds_stage(0, 0);
    // one burst behind the DMA: side inputs of tile row (tid & 127) and this lane's two thresholds; nothing
    // is consumed before every load has been issued
    const float side_aux = METRIC == METRIC_L2 ? a.norm2[side_ri] : (METRIC == METRIC_COS ? a.rnorm[side_ri] : 0.f);
    uint8_t side_vis = 1;
    if (a.mask) side_vis = a.mask[side_ri];
    uint64_t tau_raw[2];
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int qj = q0 + wc * 64 + tn * 32 + l31;
        tau_raw[tn] = a.boot ? 0ull : a.cs.tau[qj < a.nq ? qj : a.nq - 1];
        if (qj >= a.nq) tau_raw[tn] = 0ull;
    }
    if (tid < BM) {
        s_aux[tid] = side_aux;
        s_vis[tid] = (row0 + tid <= last_row && side_vis) ? (uint8_t)1 : (uint8_t)0;
        s_rowid[tid] = (uint32_t)side_ri;
    }
    // admission thresholds of this lane's two queries, split into (key, row)
    float tau_key[2];
    uint32_t tau_row[2];
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        tau_key[tn] = tau_key_of(tau_raw[tn]);
        tau_row[tn] = entry_row(tau_raw[tn]);
    }
    if (!GLDS) {
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
    }
    __syncthreads();

    uint64_t stamp_t0 = 0, stamp_r0 = 0;
    if (ABL == 5) { stamp_t0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
    // Main loop.  Per K-step (BK = 32): the next stage's global loads are issued first, the
    // fragment reads of sub-step s+1 are issued before the 16 MFMAs of sub-step s (register
    // double buffer), and the LDS write of the next stage happens in the middle of the MFMA
    // stream (its target buffer was released by the barrier that ended the previous K-step), so
    // the only serial section left at the end of a K-step is the barrier itself.
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        const bool has_next = (ABL == 2 || ABL == 4) ? false : (kt + 1 < nk);
        if (has_next) {
            const int k0 = (kt + 1) * BK;
            if (GLDS) {
                glds_stage(cur ^ 1, k0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    ra[i] = load_chunk<ALIGNED>(a.X, st_xrow[i], a.D, k0 + st_ch[i] * 4);
                    rb[i] = load_chunk<ALIGNED>(a.Q, st_qrow[i], a.D, k0 + st_ch[i] * 4);
                }
            }
        }
        const float *As = lds[cur][0];
        const float *Bs = lds[cur][1];
        if (SPLIT == 2) {
            // split in registers (as gemm_filter_narrow_kernel<.., SPLIT>): the LDS image is plain f32; MFMA k-step ks
            // covers floats [16 ks, 16 ks + 16) of the K-step, lane half h supplies 8 of them = two 16-B chunks
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int ch = 4 * ks + 2 * h;
                bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int ra_ = wr * 64 + t * 32 + l31, rb_ = wc * 64 + t * 32 + l31;
                    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(&As[swz_off(ra_, ch)]), a1 = *reinterpret_cast<const f32x4 *>(&As[swz_off(ra_, ch + 1)]);
                    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(&Bs[swz_off(rb_, ch)]), b1 = *reinterpret_cast<const f32x4 *>(&Bs[swz_off(rb_, ch + 1)]);
                    const float xa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    const float xb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const __bf16 ha = (__bf16)xa[i], hb = (__bf16)xb[i];
                        ah[t][i] = ha;
                        al[t][i] = (__bf16)(xa[i] - (float)ha);
                        bh[t][i] = hb;
                        bl[t][i] = (__bf16)(xb[i] - (float)hb);
                    }
                }
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    }
            }
        } else if (SPLIT == 1) {
            // a row's 128-B piece = two 16-k groups of [hi 32 B | lo 32 B]: for the 16-k step ks, lane half h supplies
            // k = 8h .. 8h+7 of it: chunk 4 ks + h (hi) and 4 ks + 2 + h (lo)
            bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2]; // [buffer][tile]
            auto ldfrag = [&](int buf, int ks) {
                const int kb = 4 * ks + h;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int ra_ = wr * 64 + t * 32 + l31, rb_ = wc * 64 + t * 32 + l31;
                    ah[buf][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&As[swz_off(ra_, kb)]));
                    al[buf][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&As[swz_off(ra_, 2 + kb)]));
                    bh[buf][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Bs[swz_off(rb_, kb)]));
                    bl[buf][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Bs[swz_off(rb_, 2 + kb)]));
                }
            };
            ldfrag(0, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int cb = ks & 1;
                if (ks < 1) ldfrag(cb ^ 1, ks + 1);
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb][tm], bh[cb][tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb][tm], bl[cb][tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb][tm], bh[cb][tn], acc[tm][tn], 0, 0, 0);
                    }
            }
        } else {
        f32x4 fa[2][2], fb[2][2];
        if ((ABL != 3 && ABL != 4) || kt == 0) {
#pragma unroll
            for (int t = 0; t < 2; t++) {
                fa[0][t] = *reinterpret_cast<const f32x4 *>(&As[swz_off(wr * 64 + t * 32 + l31, h)]);
                fb[0][t] = *reinterpret_cast<const f32x4 *>(&Bs[swz_off(wc * 64 + t * 32 + l31, h)]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; t++) { fa[0][t] = ra[t]; fb[0][t] = rb[t]; }
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int cb = s & 1, nb = cb ^ 1;
            if (ABL == 3 || ABL == 4) {
#pragma unroll
                for (int t = 0; t < 2; t++) { fa[nb][t] = fa[cb][t] + 1.0f; fb[nb][t] = fb[cb][t]; }
            } else if (s < 3) {
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
            if (!GLDS && s == 1 && has_next) {
                const int nxt = cur ^ 1;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    *reinterpret_cast<f32x4 *>(&lds[nxt][0][swz_off(st_row[i], st_ch[i])]) = ra[i];
                    *reinterpret_cast<f32x4 *>(&lds[nxt][1][swz_off(st_row[i], st_ch[i])]) = rb[i];
                }
            }
        }
        }
        if (ABL != 1 && ABL != 4) __syncthreads();
    }
    if (ABL == 5 && threadIdx.x == 0) {
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd((unsigned long long *)&g_clock_probe[0], (unsigned long long)(t1 - stamp_t0));
        atomicAdd((unsigned long long *)&g_clock_probe[1], (unsigned long long)(r1 - stamp_r0));
        atomicAdd((unsigned long long *)&g_clock_probe[2], 1ull);
        atomicAdd((unsigned long long *)&g_clock_probe[3], (unsigned long long)(stamp_t0 - stamp_entry));
        stamp_t0 = t1; // reuse: epilogue start
    }

    // ---- epilogue: key + admission -------------------------------------------------
    // C layout (32x32): col = lane&31 (query), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // All per-row side inputs (norm / mask) are fetched up front in one burst: a load per
    // element inside the admission loop would serialise 64 L2 round trips per lane.
    if (ABL == 6) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    float aux[2][4][4];
    uint32_t rid[2][4][4]; // corpus row ids of this lane's 32 rows
    uint32_t vbits = 0; // bit (tm*16 + g*4 + e): row visible (in range and not masked out)
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int lr = wr * 64 + tm * 32 + 8 * g + 4 * h; // 4 consecutive local rows
            const f32x4 av = *reinterpret_cast<const f32x4 *>(&s_aux[lr]);
            const uint4 rv = *reinterpret_cast<const uint4 *>(&s_rowid[lr]);
            rid[tm][g][0] = rv.x; rid[tm][g][1] = rv.y; rid[tm][g][2] = rv.z; rid[tm][g][3] = rv.w;
            const uint32_t vv = *reinterpret_cast<const uint32_t *>(&s_vis[lr]); // 4 bytes of 0/1
            aux[tm][g][0] = av.x; aux[tm][g][1] = av.y; aux[tm][g][2] = av.z; aux[tm][g][3] = av.w;
            // gather the four 0/1 bytes into 4 adjacent bits
            const uint32_t nib = (vv & 1u) | ((vv >> 7) & 2u) | ((vv >> 14) & 4u) | ((vv >> 21) & 8u);
            vbits |= nib << (tm * 16 + g * 4);
        }
    auto key_of = [&](float dot, float ax) -> float {
        if (METRIC == METRIC_L2) return fmaf(-2.0f, dot, ax);
        if (METRIC == METRIC_COS) return -dot * ax;
        return -dot;
    };
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int qj = q0 + wc * 64 + tn * 32 + l31;
        const bool qok = qj < a.nq;
        uint64_t *list = a.cs.lists + (size_t)(qok ? qj : 0) * a.cs.cap;
        if (a.boot) {
            // bootstrap chunk: entry of row r goes to list[r - row_begin]; a lane's 4 consecutive
            // rows are 32 contiguous bytes -> two 16-B stores
            if (qok) {
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int64_t rbase = row0 + wr * 64 + tm * 32 + 8 * g + 4 * h;
                        uint64_t ent[4];
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            ent[e] = ((vbits >> (tm * 16 + g * 4 + e)) & 1u)
                                         ? pack_entry(key_of(acc[tm][tn][4 * g + e], aux[tm][g][e]), rid[tm][g][e])
                                         : kEntryMax;
                        uint64_t *dst = list + (rbase - a.row_begin);
                        if (rbase + 3 < a.row_end) {
                            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                            u64x2 v0 = {ent[0], ent[1]}, v1 = {ent[2], ent[3]};
                            *reinterpret_cast<u64x2 *>(dst) = v0;
                            *reinterpret_cast<u64x2 *>(dst + 2) = v1;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                if (rbase + e < a.row_end) dst[e] = ent[e];
                        }
                    }
            }
            continue;
        }
        // pass 1 (branch-free): which of this lane's 32 elements pass the admission test
        //   entry < tau  <=>  key < tau_key, or equal keys and a lower row
        // (float compares treat -0 == +0, matching the +0-canonical packed keys; tau of an
        //  out-of-range query decodes to NaN, so nothing passes).
        const float tk = tau_key[tn];
        const uint32_t tr = tau_row[tn];
        uint32_t bits = 0;
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float key = key_of(acc[tm][tn][4 * g + e], aux[tm][g][e]);
                    const uint32_t ri = rid[tm][g][e];
                    const uint32_t lt = (uint32_t)(key < tk) | ((uint32_t)(key == tk) & (uint32_t)(ri < tr));
                    bits |= lt << (tm * 16 + g * 4 + e);
                }
        bits &= vbits;
        if (ABL == 7) { asm volatile("" ::"v"(bits)); continue; }
        // ONE returning atomic per lane reserves the slots; the stores are fire-and-forget
        if (bits) {
            uint32_t pos = atomicAdd(&a.cs.cnt[qj], (uint32_t)__builtin_popcount(bits));
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (bits & (1u << (tm * 16 + g * 4 + e))) {
                            const uint32_t ri = rid[tm][g][e];
                            if (pos < a.cs.cap)
                                list[pos] = pack_entry(key_of(acc[tm][tn][4 * g + e], aux[tm][g][e]), ri);
                            pos++;
                        }
                    }
        }
    }
    if (ABL == 5) {
        __syncthreads();
        if (threadIdx.x == 0)
            atomicAdd((unsigned long long *)&g_clock_probe[4], (unsigned long long)(__builtin_amdgcn_s_memtime() - stamp_t0));
    }
}

void read_clock_probe(unsigned long long out[8], bool reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clock_probe), 8 * sizeof(unsigned long long));
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clock_probe), z, sizeof z);
    }
}

// f32 [rows][D] -> split-bf16 image of the same byte shape (D % 32 == 0): per row and 16-k group,
// 16 bf16 hi values (x rounded to nearest even) then 16 bf16 lo values (x - hi rounded): one MFMA k-block per 64 B.
__global__ __launch_bounds__(256) void split_bf16_kernel(const float *src, float *dst, int64_t n8)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + i * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(src + i * 8 + 4);
        const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const __bf16 hh = (__bf16)x[j];
            hi[j] = hh;
            lo[j] = (__bf16)(x[j] - (float)hh);
        }
        const int64_t grp = i >> 1; // 16-k group index over the flattened [rows*D/16]
        const int kb = (int)(i & 1);
        char *g = reinterpret_cast<char *>(dst) + grp * 64;
        *reinterpret_cast<bf16x8 *>(g + kb * 16) = hi;
        *reinterpret_cast<bf16x8 *>(g + 32 + kb * 16) = lo;
    }
}

void launch_split_bf16(const float *src, float *dst, int64_t rows, int D, hipStream_t s)
{
    const int64_t n8 = rows * (int64_t)D / 8;
    if (n8 <= 0) return;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n8);
}

int debug_gemm_occupancy()
{
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_filter_kernel<METRIC_COS, 2, 0>, GEMM_THREADS, 0);
    return nb;
}

int g_gemm_glds = 0;     // A/B switch (tools/ablate_gemm.py, diagnostic build): -1 = register staging
int g_gemm_ablation = 0; // profiling aid (diagnostic build only); always 0 in the product build

void launch_gemm_filter(int metric, const float *X, const float *norm2, const float *rnorm,
                        int64_t row_begin, int64_t row_end, int D, const float *Q, int nq,
                        const uint8_t *mask, const uint32_t *rowmap, CandState cs, bool boot, int split,
                        hipStream_t s)
{
    if (row_end <= row_begin || nq <= 0) return;
    GemmArgs a;
    a.rowmap = rowmap;
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
#ifdef LB_DIAG // timing-only ablations and the clock probe exist only in the diagnostic build
    if (split == 1 && g_gemm_ablation > 0 && metric == METRIC_COS) {
        switch (g_gemm_ablation) {
        case 1: hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 1, true, 1>), grid, dim3(GEMM_THREADS), 0, s, a); return;
        case 2: hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 2, true, 1>), grid, dim3(GEMM_THREADS), 0, s, a); return;
        case 5: hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 5, true, 1>), grid, dim3(GEMM_THREADS), 0, s, a); return;
        case 6: hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 6, true, 1>), grid, dim3(GEMM_THREADS), 0, s, a); return;
        default: break;
        }
    }
    static const int env_abl = lb_tunable("LB_GEMM_ABL", 0);
    if (env_abl > 0 && g_gemm_ablation == 0) g_gemm_ablation = env_abl;
    if (!split && g_gemm_ablation == 8 && metric == METRIC_COS && mode == 2) { // A/B: default cache policy on the corpus stream
        hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 8, true, 0>), grid, dim3(GEMM_THREADS), 0, s, a);
        return;
    }
    if (!split && g_gemm_ablation > 0 && metric == METRIC_COS && mode == 2) {
        switch (g_gemm_ablation) {
        case 1: hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 1>), grid, dim3(GEMM_THREADS), 0, s, a); return;
        case 2: hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 2>), grid, dim3(GEMM_THREADS), 0, s, a); return;
        case 3: hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 3>), grid, dim3(GEMM_THREADS), 0, s, a); return;
        case 4: hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 4>), grid, dim3(GEMM_THREADS), 0, s, a); return;
        case 5: {
            // LB_GEMM_1WG=1: pad the LDS request so that only ONE workgroup fits a CU (diagnostic: main-loop
            // cycles of a wave that has its SIMD to itself)
            static const int one_wg = lb_tunable("LB_GEMM_1WG", 0);
            const size_t pad = one_wg ? 40 * 1024 : 0;
            if (pad) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_filter_kernel<METRIC_COS, 2, 5>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
            hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 5>), grid, dim3(GEMM_THREADS), pad, s, a);
            return;
        }
        case 6: hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 6>), grid, dim3(GEMM_THREADS), 0, s, a); return;
        case 7: hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 7>), grid, dim3(GEMM_THREADS), 0, s, a); return;
        default: break;
        }
    }
#endif
    // direct-to-LDS staging is the default for the aligned, D % 32 == 0 case (LB_GEMM_GLDS=0 or
    // g_gemm_glds = -1 selects the register-staged pipeline for A/B runs)
    static const bool env_noglds = lb_tunable("LB_GEMM_GLDS", 1) == 0;
    if (split == 1) {
        // X / Q are split-bf16 images (caller guarantees D % 32 == 0 and 16-B alignment)
        if (metric == METRIC_L2) hipLaunchKernelGGL((gemm_filter_kernel<METRIC_L2, 2, 0, true, 1>), grid, dim3(GEMM_THREADS), 0, s, a);
        else if (metric == METRIC_COS) hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 0, true, 1>), grid, dim3(GEMM_THREADS), 0, s, a);
        else hipLaunchKernelGGL((gemm_filter_kernel<METRIC_DOT, 2, 0, true, 1>), grid, dim3(GEMM_THREADS), 0, s, a);
        return;
    }
    if (split == 2) {
        // X / Q are the plain f32 operands, split in registers (D % 32 == 0, 16-B aligned)
        if (metric == METRIC_L2) hipLaunchKernelGGL((gemm_filter_kernel<METRIC_L2, 2, 0, true, 2>), grid, dim3(GEMM_THREADS), 0, s, a);
        else if (metric == METRIC_COS) hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 0, true, 2>), grid, dim3(GEMM_THREADS), 0, s, a);
        else hipLaunchKernelGGL((gemm_filter_kernel<METRIC_DOT, 2, 0, true, 2>), grid, dim3(GEMM_THREADS), 0, s, a);
        return;
    }
    if (g_gemm_glds >= 0 && !env_noglds && g_gemm_ablation == 0 && mode == 2) {
        if (metric == METRIC_L2) hipLaunchKernelGGL((gemm_filter_kernel<METRIC_L2, 2, 0, true>), grid, dim3(GEMM_THREADS), 0, s, a);
        else if (metric == METRIC_COS) hipLaunchKernelGGL((gemm_filter_kernel<METRIC_COS, 2, 0, true>), grid, dim3(GEMM_THREADS), 0, s, a);
        else hipLaunchKernelGGL((gemm_filter_kernel<METRIC_DOT, 2, 0, true>), grid, dim3(GEMM_THREADS), 0, s, a);
        return;
    }
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
