// kernels_gemm_tall.hip -- candidate generation on the split-bf16 contraction with a 256-row tile.
//
// Same contract as gemm_filter_kernel (kernels_gemm.hip): S = X_tile . Q_tile^T, metric key and admission test
// fused into the epilogue; ranking semantics of BruteForceIndex.SearchVectors
// (internal/store/adaptive_index.go:161-225), per-pair arithmetic of the *Batch functions
// (internal/simd/batch_operations.go:64-157) approximated for the CANDIDATE keys only (the reported distances
// come from the exact re-rank).
//
// Why another tile: with the products on v_mfma_f32_32x32x16_bf16 (hi*hi + hi*lo + lo*hi) the 128 x 128 tile of
// kernels_gemm.hip is bound by LDS traffic, not by the matrix pipe -- per K-step of 32 a CU moves 2 x 32 KB of
// DMA writes and 8 waves x 16 ds_read_b128 through an LDS that delivers 128 B/clk, as many cycles as the MFMAs
// themselves.  Here a workgroup owns 256 corpus rows x 128 queries and each of its 4 waves a 128 x 64 block
// (4 x 2 MFMA tiles, 128 accumulator registers): 12 fragment reads per 24 MFMAs instead of 16, 24 KB of staging
// per 256 x 128 x 16 products instead of 32 KB per 128 x 128 x 32 -- three quarters of the LDS cycles per
// product on both counts, and three quarters of the conversions when the corpus operand is split in registers.
// K-step = 16 elements (one MFMA k-block, 64 B per row) keeps a stage at 24 KB, so a 3-stage ring is 72 KB and
// two workgroups share a CU (the other one's main loop covers this one's epilogue).  The ring is filled by
// direct-to-LDS DMA written as inline asm: the compiler's wait-count pass does not see it, so the only waits are
// the counted ones below (with __builtin_amdgcn_global_load_lds every later LDS read waits for vmcnt(0), i.e.
// for the stage that was just requested).
//
// Measured and dropped (1M x 768, 1024 queries, no effect beyond the 3 % run-to-run noise): wave priority raised over the
// MFMA block, a fixed priority difference between the two workgroups of a CU, the DMA requests issued behind the first
// fragment reads instead of in front of them, and an epilogue with one atomic round trip per lane instead of one per
// MFMA tile.  rocprofv3 (tools/prof_tall.sh): no LDS bank conflicts in the main loop, LDS index unit 13 % busy, shader clock 1.77 GHz under
// this kernel (2.10 under the f32 kernel): the matrix pipe is busy 60 % of the kernel's cycles.
//
// Operands.  Queries always come as the split image (launch_split_bf16 on the batch: hi / lo bf16 pairs in the
// bytes of the f32 row).  The corpus is either the same kind of image (ASPLIT == 1, LB_CAND_SPLIT_BF16) or the
// plain f32 rows split in registers after the LDS read (ASPLIT == 2: no second copy of the corpus).
// Image layout (kernels_gemm.hip: split_bf16_kernel): per row and group of 16 k, 32 B of hi then 32 B of lo.
#include "lb_device.h"

namespace lb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TBM = 256, TBK = 16;

struct TallArgs {
    const float *X;
    const float *norm2;
    const float *rnorm;
    int64_t row_begin, row_end;
    int D;
    const float *Q; // split image of the query batch
    int nq;
    const uint8_t *mask;
    const uint32_t *rowmap;
    CandState cs;
    int n_row_tiles, n_q_tiles;
    int boot;
    int abl; // diagnostic build, timing only (results are wrong): 1 = no staging DMA after the first stages, 3 = staging only
};

// 64-B rows: the 16-B chunk of row r lands at position chunk ^ ((r >> 2) & 3), which spreads the 16 lanes of a
// ds_read_b128 group (16 consecutive rows, same chunk) over all 64 banks
__device__ __forceinline__ int tswz(int row, int chunk) { return row * TBK + ((chunk ^ ((row >> 2) & 3)) << 2); }

// 16 B per lane straight into LDS (lane l lands at lds_addr + 16 l).  Default cache policy also for the corpus: a row's
// 128-B line is consumed over TWO K-steps here, so it has to survive in L2 in between (with the nt policy of the other
// kernels this one ran 30 % slower: 5.56 vs 4.28 ms at 1M x 768, 1024 queries).
__device__ __forceinline__ void tall_dma16(const void *gsrc, uint32_t lds_addr)
{
    uint32_t save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(gsrc), "s"(lds_addr) : "memory");
}
// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
template <int N>
__device__ __forceinline__ void tall_wait_vmcnt()
{
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void tall_split8(const f32x4 x0, const f32x4 x1, bf16x8 &hi, bf16x8 &lo)
{
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = h;
        lo[i] = (__bf16)(x[i] - (float)h);
    }
}

// NWC = wave columns: 2 -> 256 x 128 tile, 4 waves, 3-stage ring of 24 KB, two workgroups per CU;
//                     4 -> 256 x 256 tile, 8 waves, 4-stage ring of 32 KB, one workgroup per CU (two thirds of the
//                          staging bytes per product and 96 KB of them in flight per CU instead of ~72)
template <int METRIC, int ASPLIT, int NWC>
__global__ __launch_bounds__(128 * NWC, NWC == 2 ? 2 : 1) void gemm_filter_tall_kernel(TallArgs a)
{
    constexpr int TBN = 64 * NWC;                // queries per tile
    constexpr int TTHREADS = 128 * NWC;          // 2 x NWC waves, each 128 rows x 64 queries
    constexpr int TNST = NWC == 2 ? 3 : 4;       // ring stages
    constexpr int T_STAGE_F = (TBM + TBN) * TBK; // floats per stage
    constexpr int NA = 8 / NWC;                  // DMA instructions per wave and stage: NA x 16 corpus rows ...
    constexpr int NB = 2;                        // ... + 2 x 16 query rows
    constexpr int T_NI = NA + NB;
    constexpr int AROWS = 16 * NA;               // corpus rows staged by one wave
    // XCD-aware order as in gemm_filter_kernel: the query tiles of one corpus tile run back to back on one XCD
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int in_xcd = b >> 3;
    const int qt = in_xcd % a.n_q_tiles;
    const int rt = (in_xcd / a.n_q_tiles) * 8 + xcd;
    if (rt >= a.n_row_tiles) return;

    extern __shared__ __attribute__((aligned(16))) float tlds[];
    float *ring = tlds;                                              // [TNST][A 256x16 | B 128x16]
    float *s_aux = ring + TNST * T_STAGE_F;                          // [TBM]
    uint32_t *s_rowid = reinterpret_cast<uint32_t *>(s_aux + TBM);   // [TBM]
    uint8_t *s_vis = reinterpret_cast<uint8_t *>(s_rowid + TBM);     // [TBM]
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)ring;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWC, wc = wave % NWC;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t row0 = a.row_begin + (int64_t)rt * TBM;
    const int q0 = qt * TBN;
    const int64_t last_row = a.row_end - 1;
    const int last_q = a.nq - 1;

    auto corpus_row = [&](int64_t pos) -> int64_t {
        if (pos > last_row) pos = last_row;
        return a.rowmap ? (int64_t)a.rowmap[pos] : pos;
    };
    const int64_t side_ri = corpus_row(row0 + (tid & (TBM - 1))); // threads 0 .. TBM-1 carry one side input each

    // DMA sources.  Instruction i of this wave fills 16 rows (1 KiB): corpus rows AROWS wave + 16 i .. +15, query rows
    // 32 wave + 16 i .. +15; lane l lands at (row l / 4, chunk position l % 4).
    const float *srcA[NA], *srcB[NB];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int row = wave * AROWS + i * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        srcA[i] = a.X + corpus_row(row0 + row) * (int64_t)a.D + 4 * c;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int row = wave * 32 + i * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        int qr = q0 + row;
        if (qr > last_q) qr = last_q;
        srcB[i] = a.Q + (int64_t)qr * a.D + 4 * c;
    }
    auto issue = [&](int kt) {
        const uint32_t A = ring_base + (uint32_t)(kt % TNST) * (T_STAGE_F * 4);
        const uint32_t B = A + TBM * TBK * 4;
        const int k0 = kt * TBK;
#pragma unroll
        for (int i = 0; i < NA; i++) tall_dma16(srcA[i] + k0, A + (uint32_t)((wave * AROWS + i * 16) * TBK * 4));
#pragma unroll
        for (int i = 0; i < NB; i++) tall_dma16(srcB[i] + k0, B + (uint32_t)((wave * 32 + i * 16) * TBK * 4));
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nk = a.D / TBK; // D % 32 == 0 (launcher)
    issue(0);
    // one burst behind the first stage (not needed before the epilogue): side inputs and thresholds
    const float side_aux = METRIC == METRIC_L2 ? a.norm2[side_ri] : (METRIC == METRIC_COS ? a.rnorm[side_ri] : 0.f);
    uint8_t side_vis = 1;
    if (a.mask) side_vis = a.mask[side_ri];
    float tk[2];
    uint32_t tr[2];
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int qj = q0 + wc * 64 + tn * 32 + l31;
        uint64_t tau = a.boot ? 0ull : a.cs.tau[qj < a.nq ? qj : a.nq - 1];
        if (qj >= a.nq) tau = 0ull;
        tk[tn] = tau_key_of(tau);
        tr[tn] = entry_row(tau);
    }
    if (tid < TBM) {
        s_aux[tid] = side_aux;
        s_vis[tid] = (row0 + tid <= last_row && side_vis) ? (uint8_t)1 : (uint8_t)0;
        s_rowid[tid] = (uint32_t)side_ri;
    }
    tall_wait_vmcnt<0>(); // stage 0 and every load above (the compiler cannot count the DMA)
    __syncthreads();      // side inputs written (the loop below uses bare barriers: no LDS writes of its own)
#pragma unroll
    for (int st = 1; st < TNST - 1; st++)
        if (st < nk) issue(st);

    for (int kt = 0; kt < nk; kt++) {
        // stage kt has landed once only the requests of the stages behind it (up to TNST - 2 of them) are outstanding
        {
            const int ahead = nk - 1 - kt < TNST - 2 ? nk - 1 - kt : TNST - 2;
            if (ahead >= 2) tall_wait_vmcnt<2 * T_NI>();
            else if (ahead == 1) tall_wait_vmcnt<T_NI>();
            else tall_wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier(); // everyone's part of stage kt is in; everyone is done reading stage kt - 1
        asm volatile("" ::: "memory");
#ifdef LB_DIAG
        if (kt + TNST - 1 < nk && a.abl != 1) issue(kt + TNST - 1);
#else
        if (kt + TNST - 1 < nk) issue(kt + TNST - 1); // into the slot read at step kt - 1
#endif
#ifdef LB_DIAG
        if (a.abl == 3) continue; // staging only (timing)
#endif
        const float *As = ring + (kt % TNST) * T_STAGE_F;
        const float *Bs = As + TBM * TBK;
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int r = wc * 64 + t * 32 + l31;
            bh[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Bs[tswz(r, h)]));
            bl[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Bs[tswz(r, 2 + h)]));
        }
        f32x4 fa[2][2]; // [buffer][chunk]: raw 16-B chunks of the next corpus tile row (hi/lo halves, or 8 floats)
        auto ldA = [&](int buf, int tm) {
            const int r = wr * 128 + tm * 32 + l31;
            if (ASPLIT == 1) {
                fa[buf][0] = *reinterpret_cast<const f32x4 *>(&As[tswz(r, h)]);
                fa[buf][1] = *reinterpret_cast<const f32x4 *>(&As[tswz(r, 2 + h)]);
            } else {
                fa[buf][0] = *reinterpret_cast<const f32x4 *>(&As[tswz(r, 2 * h)]);
                fa[buf][1] = *reinterpret_cast<const f32x4 *>(&As[tswz(r, 2 * h + 1)]);
            }
        };
        ldA(0, 0);
#pragma unroll
        for (int tm = 0; tm < 4; tm++) {
            const int cb = tm & 1;
            if (tm < 3) ldA(cb ^ 1, tm + 1);
            bf16x8 ah, al;
            if (ASPLIT == 1) {
                ah = __builtin_bit_cast(bf16x8, fa[cb][0]);
                al = __builtin_bit_cast(bf16x8, fa[cb][1]);
            } else {
                tall_split8(fa[cb][0], fa[cb][1], ah, al);
            }
#pragma unroll
            for (int tn = 0; tn < 2; tn++) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: key + admission, one MFMA row tile (this lane's 16 rows of it) at a time ----------------
    // C layout (32x32): col = lane & 31 (query), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    // (A variant with a memory-free first pass over all 128 elements and ONE atomic round trip per lane was measured
    // 3 % slower: the other workgroup's MFMAs already cover these round trips.)
    auto key_of = [&](float dot, float ax) -> float {
        if (METRIC == METRIC_L2) return fmaf(-2.0f, dot, ax);
        if (METRIC == METRIC_COS) return -dot * ax;
        return -dot;
    };
    // workgroup-local admission list, carved from the ring (free: every wave is past the last K-step's reads once the
    // barrier below is behind it)
    constexpr int FL_CAP = 2048;
    uint32_t *s_lcnt = reinterpret_cast<uint32_t *>(ring);              // entries in the list
    uint32_t *s_qcnt = s_lcnt + 1;                                      // [TBN] of them per query of the tile ...
    uint32_t *s_qbase = s_qcnt + TBN;                                   // [TBN] ... and where they start in the query's list
    uint64_t *s_lent = reinterpret_cast<uint64_t *>(ring + 1024);
    uint16_t *s_lq = reinterpret_cast<uint16_t *>(ring + 1024 + 2 * FL_CAP);
    uint16_t *s_lr = s_lq + FL_CAP;                                     // rank of the entry among its query's
    __syncthreads();
    if (tid == 0) *s_lcnt = 0;
    if (tid < TBN) s_qcnt[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < 4; tm++) {
        float aux[4][4];
        uint32_t rid[4][4];
        uint32_t vbits = 0; // bit (g*4 + e): row visible (in range and not masked out)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int lr = wr * 128 + tm * 32 + 8 * g + 4 * h; // 4 consecutive local rows
            const f32x4 av = *reinterpret_cast<const f32x4 *>(&s_aux[lr]);
            const uint4 rv = *reinterpret_cast<const uint4 *>(&s_rowid[lr]);
            const uint32_t vv = *reinterpret_cast<const uint32_t *>(&s_vis[lr]);
            aux[g][0] = av.x; aux[g][1] = av.y; aux[g][2] = av.z; aux[g][3] = av.w;
            rid[g][0] = rv.x; rid[g][1] = rv.y; rid[g][2] = rv.z; rid[g][3] = rv.w;
            const uint32_t nib = (vv & 1u) | ((vv >> 7) & 2u) | ((vv >> 14) & 4u) | ((vv >> 21) & 8u);
            vbits |= nib << (g * 4);
        }
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
            const int qj = q0 + wc * 64 + tn * 32 + l31;
            const bool qok = qj < a.nq;
            uint64_t *list = a.cs.lists + (size_t)(qok ? qj : 0) * a.cs.cap;
            if (a.boot) { // sample pass: the entry of position p goes to list[p - row_begin]
                if (qok) {
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int64_t rbase = row0 + wr * 128 + tm * 32 + 8 * g + 4 * h;
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (rbase + e < a.row_end)
                                list[rbase + e - a.row_begin] =
                                    ((vbits >> (g * 4 + e)) & 1u) ? pack_entry(key_of(acc[tm][tn][4 * g + e], aux[g][e]), rid[g][e])
                                                                  : kEntryMax;
                    }
                }
                continue;
            }
            // entry < tau  <=>  key < tau_key, or equal keys and a lower row (a padded query's tau decodes to NaN)
            uint32_t bits = 0;
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float key = key_of(acc[tm][tn][4 * g + e], aux[g][e]);
                    const uint32_t lt = (uint32_t)(key < tk[tn]) | ((uint32_t)(key == tk[tn]) & (uint32_t)(rid[g][e] < tr[tn]));
                    bits |= lt << (g * 4 + e);
                }
            bits &= vbits;
            if (bits) { // workgroup-local list first (as the fused narrow kernel): one LDS atomic per lane
                const uint32_t n = (uint32_t)__builtin_popcount(bits);
                uint32_t lp = atomicAdd(s_lcnt, n);
                if (lp + n <= (uint32_t)FL_CAP) {
                    uint32_t lr = atomicAdd(&s_qcnt[qj - q0], n);
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (bits & (1u << (g * 4 + e))) {
                                s_lent[lp] = pack_entry(key_of(acc[tm][tn][4 * g + e], aux[g][e]), rid[g][e]);
                                s_lq[lp] = (uint16_t)(qj - q0);
                                s_lr[lp] = (uint16_t)lr;
                                lp++;
                                lr++;
                            }
                    bits = 0;
                } else {
                    for (uint32_t i = lp; i < lp + n && i < (uint32_t)FL_CAP; i++) s_lent[i] = kEntryMax; // reserved, unused
                }
            }
            if (bits) { // (local list full) one returning atomic reserves the lane's slots; the stores are fire-and-forget
                uint32_t pos = atomicAdd(&a.cs.cnt[qj], (uint32_t)__builtin_popcount(bits));
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (bits & (1u << (g * 4 + e))) {
                            if (pos < a.cs.cap) list[pos] = pack_entry(key_of(acc[tm][tn][4 * g + e], aux[g][e]), rid[g][e]);
                            pos++;
                        }
            }
        }
    }
    if (!a.boot) { // flush the workgroup-local admissions: ONE returning global atomic per query of the tile, all in flight
        __syncthreads();
        if (tid < TBN) {
            const uint32_t n = s_qcnt[tid];
            s_qbase[tid] = n ? atomicAdd(&a.cs.cnt[q0 + tid], n) : 0u; // (n != 0 implies a real query)
        }
        __syncthreads();
        const uint32_t total = *s_lcnt < (uint32_t)FL_CAP ? *s_lcnt : (uint32_t)FL_CAP;
        for (uint32_t i = tid; i < total; i += TTHREADS) {
            const uint64_t ent = s_lent[i];
            if (ent == kEntryMax) continue;
            const int ql = (int)s_lq[i];
            const uint32_t pos = s_qbase[ql] + (uint32_t)s_lr[i];
            if (pos < a.cs.cap) a.cs.lists[(size_t)(q0 + ql) * a.cs.cap + pos] = ent;
        }
    }
}

// Requires D % 32 == 0, 16-B aligned X / Q; Q is the split image of the batch; asplit: 1 = X is the split image of
// the corpus, 2 = X is the plain f32 corpus.
void launch_gemm_filter_tall(int metric, const float *X, const float *norm2, const float *rnorm, int64_t row_begin,
                             int64_t row_end, int D, const float *Qs, int nq, const uint8_t *mask, const uint32_t *rowmap,
                             CandState cs, bool boot, int asplit, hipStream_t s)
{
    if (row_end <= row_begin || nq <= 0) return;
    TallArgs a;
    a.rowmap = rowmap;
    a.X = X; a.norm2 = norm2; a.rnorm = rnorm; a.row_begin = row_begin; a.row_end = row_end; a.D = D;
    a.Q = Qs; a.nq = nq; a.mask = mask; a.cs = cs; a.boot = boot ? 1 : 0;
    static const int abl = lb_tunable("LB_TALL_ABL", 0);
    a.abl = abl;
    // 256-query tiles for large batches on the corpus image (measured at 1M x 768, 1024 queries: 4.09 vs 4.27 ms; with
    // the corpus split in registers, or at 256-384 queries, the 128-query tile is the faster one)
    static const int wide_min = lb_tunable("LB_TALL_WIDE_MINQ", 512);
    const bool wide = asplit == 1 && nq >= wide_min && nq % 256 == 0;
    const int tbn = wide ? 256 : 128;
    a.n_row_tiles = (int)((row_end - row_begin + TBM - 1) / TBM);
    a.n_q_tiles = (nq + tbn - 1) / tbn;
    const int groups = (a.n_row_tiles + 7) / 8;
    dim3 grid((unsigned)(groups * 8 * a.n_q_tiles));
    const size_t shmem = (size_t)(wide ? 4 : 3) * (TBM + tbn) * TBK * 4 + TBM * 4 + TBM * 4 + TBM;
#define LB_TALL(M, SP, NWC)                                                                                         \
    do {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_filter_tall_kernel<M, SP, NWC>),             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); /* per device */         \
        hipLaunchKernelGGL((gemm_filter_tall_kernel<M, SP, NWC>), grid, dim3(128 * NWC), shmem, s, a);              \
    } while (0)
#define LB_TALL_M(M)                                  \
    do {                                              \
        if (asplit == 1) {                            \
            if (wide) LB_TALL(M, 1, 4);               \
            else LB_TALL(M, 1, 2);                    \
        } else {                                      \
            if (wide) LB_TALL(M, 2, 4);               \
            else LB_TALL(M, 2, 2);                    \
        }                                             \
    } while (0)
    if (metric == METRIC_L2) LB_TALL_M(METRIC_L2);
    else if (metric == METRIC_COS) LB_TALL_M(METRIC_COS);
    else LB_TALL_M(METRIC_DOT);
#undef LB_TALL_M
#undef LB_TALL
}

} // namespace lb
