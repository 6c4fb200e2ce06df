// EXPERIMENT, NOT BUILT (round 2): a persistent one-workgroup-per-CU version of the narrow split-bf16 kernel with
// a 4- / 6-stage direct-to-LDS ring (96 KB of corpus bytes in flight per CU instead of 64) and counted vmcnt
// waits.  Measured at 1M x 768: 531 us per pass at 8 queries against 491 us for the two-stage kernel, 915 vs 590
// at 64 -- slower.  The ring buys nothing because the LDS-DMA path itself tops out near 6.0-6.1 TB/s from HBM
// (MI355X_MICROARCH.md, "Indexed rows: gather into LDS": 23-24 GB/s per CU), which the two-stage kernel already
// reaches, while four waves per CU leave the conversions and LDS reads of a K-step without latency cover.  Kept
// for the record of two compiler findings: (1) after __builtin_amdgcn_global_load_lds the wait-count pass puts
// s_waitcnt vmcnt(0) in front of EVERY later LDS access, so a ring deeper than one stage needs the DMA as inline
// asm; (2) a returning global atomic on any path of the loop leaves a pending register that forces vmcnt(0) on
// the common path.  Also known to be wrong for D / 32 < 3 (side-input double buffer).
// kernels_gemm_stream.hip -- candidate generation for 5..384-query batches as ONE uninterrupted HBM stream.
//
// Same contract as gemm_filter_narrow_kernel<.., SPLIT> (kernels_gemm_narrow.hip): inner products of the query
// tile with every corpus row as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (operands split in
// registers), metric key, admission against the per-query threshold.  What differs is the pipeline.  With the
// contraction on bf16 the MFMA pipe is ~13 % busy, so the only thing that matters is how many corpus bytes a
// CU keeps in flight: the two-stage kernel (2 workgroups x one 32-KB stage per CU) is latency-bound at
// in_flight / latency = 64 KB / ~2.5 us = 5.8-6.3 TB/s.  Here ONE persistent workgroup per CU owns a ring of
// direct-to-LDS stages that fills the whole 160 KB (4 x 36 KB or 6 x 24 KB: three / five stages, 96 KB of
// corpus bytes, in flight), walks a flat (tile, K-step) sequence so the ring never drains between tiles, and
// waits with counted vmcnt for the OLDEST stage only.  Two things keep the vector-memory queue free of
// anything the ring would have to wait behind:
//   * the tile's side inputs (||x||^2 or 1/||x||) arrive by the same DMA path, re-issued with every stage, so
//     every stage is exactly NI instructions and the counted wait is a constant;
//   * admitted entries go to a workgroup-local LDS list with LDS atomics (lgkmcnt, not vmcnt) and are
//     flushed to the per-query HBM lists between tiles when half full and at the end (a returning global
//     atomic in the epilogue would sit behind the newest DMA in the in-order vmcnt queue, i.e. cost one full
//     ring latency per tile).
// Unfiltered, non-bootstrap passes only (the sampled-threshold main pass); everything else takes the two-stage
// kernel.  Requires D % 32 == 0 and 16-B aligned X / Q.
#include "lb_device.h"

namespace lb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SBK = 32;        // floats per row per K-step (one 128-B line)
constexpr int STHREADS = 256;
constexpr int SCL = 512;       // workgroup-local candidate list (entries)
constexpr int SMAXQ = 384;     // thresholds staged in LDS

struct StreamArgs {
    const float *X;
    const float *aux; // norm2 (L2) or rnorm (cosine); unused for dot
    int64_t row_begin, row_end;
    int D;
    const float *Q;
    int nq;
    CandState cs;
    int n_row_tiles, n_q_tiles;
};

__device__ __forceinline__ int sswz(int row, int chunk) { return row * SBK + ((chunk ^ ((row >> 1) & 7)) << 2); }

__device__ __forceinline__ void ssplit8(const f32x4 x0, const f32x4 x1, bf16x8 &hi, bf16x8 &lo)
{
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = h;
        lo[i] = (__bf16)(x[i] - (float)h);
    }
}

// Direct-to-LDS loads as inline asm (M0 = wave-uniform LDS byte address; lane l lands at M0 + l * size), with M0
// saved and restored around the instruction.  Why not __builtin_amdgcn_global_load_lds: the compiler's
// wait-count pass treats every LDS access that follows a builtin LDS-DMA as possibly aliasing it and puts
// s_waitcnt vmcnt(0) in front of the ring's ds_reads and of the epilogue's LDS traffic -- which drains the
// whole ring every K-step.  The asm form is opaque to that pass; ordering is kept by hand (counted vmcnt wait
// + barrier before a stage is read).  Extra VMEM operations the pass does not know about only make ITS waits
// more conservative (vmcnt completes in order).
__device__ __forceinline__ void dma16_nt(const void *gsrc, uint32_t lds_addr)
{
    uint32_t save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(gsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma16(const void *gsrc, uint32_t lds_addr)
{
    uint32_t save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(gsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma4(const void *gsrc, uint32_t lds_addr)
{
    uint32_t save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// s_waitcnt vmcnt(N).  gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] |
// vmcnt[5:4] << 14; expcnt = 7 and lgkmcnt = 15 mean "do not wait".
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

template <int METRIC, int NBM, int NBN> // <256, 32> or <128, 64>
__global__ __launch_bounds__(STHREADS, 1) void gemm_filter_stream_kernel(StreamArgs a)
{
    constexpr int WROWS = NBM / 4;
    constexpr int TM = WROWS / 32;
    constexpr int TN = NBN / 32;
    constexpr int NA = WROWS / 8;
    constexpr int NB = NBN / 32;
    constexpr bool HAS_AUX = METRIC != METRIC_DOT;
    constexpr int NI = NA + NB + (HAS_AUX ? 1 : 0); // DMA instructions per wave and stage
    constexpr int STAGE_F = (NBM + NBN) * SBK;
    constexpr int NST = NBM == 256 ? 4 : 6;          // 4 x 36 KB or 6 x 24 KB = 144 KB
    constexpr int WAITN = (NST - 2) * NI;            // stages issued after the one about to be consumed
    static_assert(TM * TN == 2, "two MFMA tiles per wave");
    static_assert(WAITN < 64, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) float slds[];
    float *ring = slds;
    float *s_aux = ring + NST * STAGE_F;                           // [2][NBM + 32]
    float *s_tk = s_aux + 2 * (NBM + 32);                          // [SMAXQ] threshold keys
    uint32_t *s_tr = reinterpret_cast<uint32_t *>(s_tk + SMAXQ);   // [SMAXQ] threshold rows
    uint64_t *s_cl = reinterpret_cast<uint64_t *>(s_tr + SMAXQ);   // [SCL] admitted entries
    uint16_t *s_clq = reinterpret_cast<uint16_t *>(s_cl + SCL);    // [SCL] their query slots
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_clq + SCL);   // [1]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int nk = a.D / SBK;
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)ring;
    const uint32_t aux_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)s_aux;
    // item j of this workgroup: corpus tile blockIdx.x + (j / n_q_tiles) * gridDim.x, query tile j % n_q_tiles --
    // the query tiles of one corpus tile run back to back on the same CU (the re-read comes from the caches)
    const int64_t my_row_tiles = ((int64_t)blockIdx.x < a.n_row_tiles) ? (a.n_row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int64_t my_items = my_row_tiles * a.n_q_tiles;
    const int64_t total = my_items * nk;
    const int64_t last_row = a.row_end - 1;
    const int last_q = a.nq - 1;

    // thresholds of every query of the launch -> LDS (read in the epilogues; nothing else uses plain loads)
    for (int q = tid; q < SMAXQ; q += STHREADS) {
        const uint64_t tau = q < a.nq ? a.cs.tau[q] : 0ull;
        s_tk[q] = q < a.nq ? tau_key_of(tau) : -__builtin_huge_valf();
        s_tr[q] = entry_row(tau);
    }
    if (tid == 0) *s_cnt = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (total == 0) return;

    // ---- issue side: DMA sources of the item being issued --------------------------------------
    int64_t is_item = 0; // index into this workgroup's item sequence
    int is_kt = 0;
    const float *srcA[NA];
    const float *srcB[NB];
    const float *srcAux = nullptr;
    auto setup_item = [&](int64_t j) {
        const int64_t rt = (int64_t)blockIdx.x + (j / a.n_q_tiles) * gridDim.x;
        const int qt = (int)(j % a.n_q_tiles);
        const int64_t row0 = a.row_begin + rt * NBM;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int row = wave * WROWS + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int64_t gr = row0 + row;
            if (gr > last_row) gr = last_row;
            srcA[i] = a.X + gr * (int64_t)a.D + 4 * c;
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int row = (wave * NB + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int qr = qt * NBN + row;
            if (qr > last_q) qr = last_q;
            srcB[i] = a.Q + (int64_t)qr * a.D + 4 * c;
        }
        if (HAS_AUX) {
            int64_t gr = row0 + wave * WROWS + lane; // 64 dwords per wave (rows past the wave's share are harmless)
            if (gr > last_row) gr = last_row;
            srcAux = a.aux + gr;
        }
    };
    auto issue = [&](int64_t g) {
        if (is_kt == 0) setup_item(is_item);
        const int slot = (int)(g % NST);
        const uint32_t A = ring_base + (uint32_t)slot * (STAGE_F * 4);
        const uint32_t B = A + NBM * SBK * 4;
        const int k0 = is_kt * SBK;
        // n_q_tiles > 1: the same corpus tile is read again by this workgroup's next item -- keep it cacheable
        const bool stream_once = a.n_q_tiles == 1;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            if (stream_once) dma16_nt(srcA[i] + k0, A + (uint32_t)((wave * WROWS + i * 8) * SBK * 4));
            else dma16(srcA[i] + k0, A + (uint32_t)((wave * WROWS + i * 8) * SBK * 4));
        }
#pragma unroll
        for (int i = 0; i < NB; i++) dma16(srcB[i] + k0, B + (uint32_t)((wave * NB + i) * 8 * SBK * 4));
        if (HAS_AUX) dma4(srcAux, aux_base + (uint32_t)(((is_item & 1) * (NBM + 32) + wave * WROWS) * 4));
        if (++is_kt == nk) {
            is_kt = 0;
            is_item++;
        }
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    };
    auto key_of = [&](float dot, float ax) -> float {
        if (METRIC == METRIC_L2) return fmaf(-2.0f, dot, ax);
        if (METRIC == METRIC_COS) return -dot * ax;
        return -dot;
    };
    // flush the workgroup-local list to the per-query lists (all threads; the ring must be quiet or drained)
    auto flush_list = [&]() {
        __syncthreads();
        const uint32_t n = *s_cnt < (uint32_t)SCL ? *s_cnt : (uint32_t)SCL;
        for (uint32_t i = tid; i < n; i += STHREADS) {
            const uint64_t ent = s_cl[i];
            if (ent == kEntryMax) continue; // reserved by a lane that went straight to HBM
            const int q = s_clq[i];
            const uint32_t pos = atomicAdd(&a.cs.cnt[q], 1u);
            if (pos < a.cs.cap) a.cs.lists[(size_t)q * a.cs.cap + pos] = ent;
        }
        // leave nothing the compiler knows about in flight: a pending atomic return or store would make its
        // wait-count pass put vmcnt(0) on the common path (register reuse), draining the ring every K-step
        wait_vmcnt<0>();
        __syncthreads();
        if (tid == 0) *s_cnt = 0;
        __syncthreads();
    };

    // ---- prologue: fill the ring -------------------------------------------------------------
    for (int64_t g = 0; g < NST - 1 && g < total; g++) issue(g);
    zero_acc();
    int64_t c_item = 0;
    int c_kt = 0;
    for (int64_t g = 0; g < total; g++) {
        // the oldest stage has landed once at most (NST - 2) later stages are outstanding
        if (g + NST - 1 <= total) wait_vmcnt<WAITN>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c_kt == 0 && c_item > 0 && *s_cnt > (uint32_t)(SCL / 2)) flush_list(); // (uniform: read after the barrier)
        if (g + NST - 1 < total) issue(g + NST - 1); // into the slot consumed at step g - 1

        const float *As = ring + (int)(g % NST) * STAGE_F;
        const float *Bs = As + NBM * SBK;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int ch = 4 * ks + 2 * h;
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int t = 0; t < TM; t++) {
                const int r = wave * WROWS + t * 32 + l31;
                ssplit8(*reinterpret_cast<const f32x4 *>(&As[sswz(r, ch)]), *reinterpret_cast<const f32x4 *>(&As[sswz(r, ch + 1)]), ah[t], al[t]);
            }
#pragma unroll
            for (int t = 0; t < TN; t++) {
                const int r = t * 32 + l31;
                ssplit8(*reinterpret_cast<const f32x4 *>(&Bs[sswz(r, ch)]), *reinterpret_cast<const f32x4 *>(&Bs[sswz(r, ch + 1)]), bh[t], bl[t]);
            }
#pragma unroll
            for (int tm = 0; tm < TM; tm++)
#pragma unroll
                for (int tn = 0; tn < TN; tn++) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                }
        }

        if (++c_kt == nk) { // the item's inner products are complete: key + admission
            const int64_t rt = (int64_t)blockIdx.x + (c_item / a.n_q_tiles) * gridDim.x;
            const int qt = (int)(c_item % a.n_q_tiles);
            const int64_t row0 = a.row_begin + rt * NBM;
            const float *auxv = s_aux + (c_item & 1) * (NBM + 32);
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                const int qj = qt * NBN + tn * 32 + l31;
                const bool qok = qj < a.nq;
                const float tk = s_tk[qok ? qj : 0];
                const uint32_t tr = s_tr[qok ? qj : 0];
                uint32_t bits = 0;
#pragma unroll
                for (int tm = 0; tm < TM; tm++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const int lr = wave * WROWS + tm * 32 + 8 * gq + 4 * h;
                        f32x4 av = {0.f, 0.f, 0.f, 0.f};
                        if (HAS_AUX) av = *reinterpret_cast<const f32x4 *>(&auxv[lr]);
                        const float ax[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int64_t gr = row0 + lr + e;
                            const float key = key_of(acc[tm][tn][4 * gq + e], ax[e]);
                            const uint32_t lt = (uint32_t)(key < tk) | ((uint32_t)(key == tk) & (uint32_t)((uint32_t)gr < tr));
                            bits |= ((gr <= last_row) ? lt : 0u) << (tm * 16 + gq * 4 + e);
                        }
                    }
                if (!qok) bits = 0;
                if (bits) {
                    const uint32_t cnt = (uint32_t)__builtin_popcount(bits);
                    uint32_t pos = atomicAdd(s_cnt, cnt); // LDS atomic
                    const bool local = pos + cnt <= (uint32_t)SCL;
                    if (!local) {
                        // The list is full (only when the threshold is far too loose: ~15 entries arrive per
                        // tile).  No returning global atomic here -- its pending result would make the
                        // compiler drain the ring on the common path -- the query is flagged as overflowed
                        // instead (fire-and-forget OR) and the host redoes it on the fallback schedule.
                        atomicOr(&a.cs.flags[qj], 1u);
                        for (uint32_t i = pos; i < (uint32_t)SCL; i++) s_cl[i] = kEntryMax; // reserved, never filled
                    } else {
#pragma unroll
                        for (int tm = 0; tm < TM; tm++)
#pragma unroll
                            for (int gq = 0; gq < 4; gq++) {
                                const int lr = wave * WROWS + tm * 32 + 8 * gq + 4 * h;
                                f32x4 av = {0.f, 0.f, 0.f, 0.f};
                                if (HAS_AUX) av = *reinterpret_cast<const f32x4 *>(&auxv[lr]);
                                const float ax[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                                for (int e = 0; e < 4; e++)
                                    if (bits & (1u << (tm * 16 + gq * 4 + e))) {
                                        s_cl[pos] = pack_entry(key_of(acc[tm][tn][4 * gq + e], ax[e]), (uint32_t)(row0 + lr + e));
                                        s_clq[pos] = (uint16_t)qj;
                                        pos++;
                                    }
                            }
                    }
                }
            }
            zero_acc();
            c_kt = 0;
            c_item++;
        }
    }
    flush_list(); // (clamps to what the list really holds)
}

template <int METRIC, int NBM, int NBN>
static void launch_stream_t(const StreamArgs &a, hipStream_t s)
{
    constexpr int NST = NBM == 256 ? 4 : 6;
    const size_t shmem = (size_t)NST * (NBM + NBN) * SBK * 4 + 2 * (NBM + 32) * 4 + SMAXQ * 8 + SCL * 8 + SCL * 2 + 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_filter_stream_kernel<METRIC, NBM, NBN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    const unsigned grid = (unsigned)(a.n_row_tiles < 256 ? a.n_row_tiles : 256);
    hipLaunchKernelGGL((gemm_filter_stream_kernel<METRIC, NBM, NBN>), dim3(grid), dim3(STHREADS), shmem, s, a);
}

// Requires: D % 32 == 0, 16-B aligned X / Q, nq <= 384, no mask / row map, not a bootstrap chunk.
void launch_gemm_filter_stream(int metric, const float *X, const float *norm2, const float *rnorm, int64_t row_begin,
                               int64_t row_end, int D, const float *Q, int nq, CandState cs, hipStream_t s, bool tile64)
{
    if (row_end <= row_begin || nq <= 0) return;
    StreamArgs a;
    a.X = X;
    a.aux = metric == METRIC_L2 ? norm2 : rnorm;
    a.row_begin = row_begin; a.row_end = row_end; a.D = D; a.Q = Q; a.nq = nq; a.cs = cs;
    const int bm = tile64 ? 128 : 256, bn = tile64 ? 64 : 32;
    a.n_row_tiles = (int)((row_end - row_begin + bm - 1) / bm);
    a.n_q_tiles = (nq + bn - 1) / bn;
#define LB_STREAM(M)                                      \
    do {                                                  \
        if (tile64) launch_stream_t<M, 128, 64>(a, s);    \
        else launch_stream_t<M, 256, 32>(a, s);           \
    } while (0)
    if (metric == METRIC_L2) LB_STREAM(METRIC_L2);
    else if (metric == METRIC_COS) LB_STREAM(METRIC_COS);
    else LB_STREAM(METRIC_DOT);
#undef LB_STREAM
}

} // namespace lb
