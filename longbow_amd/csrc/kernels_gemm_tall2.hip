// kernels_gemm_tall2.hip -- candidate generation on the split-bf16 contraction, 256 rows x 256 queries per workgroup,
// whole 128-B lines per row and K-step.
//
// Contract: S = X_tile . Q_tile^T as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16, metric key and admission test fused into the epilogue; ranking semantics of
// BruteForceIndex.SearchVectors (internal/store/adaptive_index.go:161-225), per-pair arithmetic of the *Batch functions
// (internal/simd/batch_operations.go:64-157) approximated for the CANDIDATE keys only (the reported distances come from the
// exact re-rank).
//
// Why this shape.  The 256 x 128 tile of rounds 2-3 (removed in round 4) staged 64 B per row and K-step of 16: HALF a cache line.  The other half of the
// line is wanted one K-step later, by which time the CU's 32 KB vector L1 has turned over, so every line crosses the
// L2 -> L1 -> LDS path twice (and, measured with rocprofv3, reaches the fabric 1.99 times per algorithmic byte).  At
// 1024 queries the kernel asked its L2 for 8.8 TB/s of 64-B pieces -- half of what the same path delivers in whole lines
// (16.8-18.8 TB/s, MI355X_MICROARCH.md "Indexed rows: gather into LDS") -- and the matrix pipe sat at 53-60 % busy waiting
// for them (SQ_WAIT_ANY 38 % of the wave cycles, profiles/r02_pmc_sq_tall.txt).  Here a K-step is 32 floats = one 128-B line
// per row, a stage is 512 lines (64 KB: 256 corpus rows + 256 query rows), and 256 x 256 x 32 x 3 products are done per
// stage: 0.67 staged bytes per 1000 bf16 flop against 1.0 for the 256 x 128 tile, none of them fetched twice.
//
// Shape.  8 waves as 4 (rows) x 2 (queries); a wave owns 64 rows x 128 queries = 2 x 4 MFMA tiles (128 accumulator
// registers).  With the corpus operand split in registers (ASPLIT == 2) each 32-row fragment is split by the TWO waves that
// share it, not by four as a 128 x 64 wave tile in this workgroup would have it: 2 conversion instructions per MFMA instead
// of 4 (the matrix pipe leaves ~6 vector issue slots per 32x32x16 MFMA to ONE wave, and two waves share a SIMD).
// One workgroup per CU (2 x 64 KB ring + side inputs), one s_barrier per K-step of 32 = per 48 MFMAs of every wave; the DMA
// of stage k+1 is issued right behind the barrier that frees its slot and has the whole step (>= 3072 matrix-pipe cycles per
// SIMD) to land.
//
// Operands as in the first tall kernel: queries always as the split image (launch_split_bf16), the corpus as the same kind
// of image (ASPLIT == 1) or as plain f32 rows split in registers after the LDS read (ASPLIT == 2).
// Image layout (kernels_gemm.hip: split_bf16_kernel): per row and group of 16 k, 32 B of hi then 32 B of lo.
#include "lb_device.h"

#include <type_traits>

namespace lb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ unsigned long long g_tall2_probe[8]; // diagnostic build, abl == 5: cycle stamps summed over waves

namespace {

constexpr int W_BM = 256, W_BN = 256, W_BK = 32; // tile rows, tile queries, floats per row and stage
constexpr int W_THREADS = 512;
constexpr int W_STAGE_F = (W_BM + W_BN) * W_BK; // floats per stage (64 KB)
constexpr int W_NST = 2;
constexpr int W_NI = 8; // DMA instructions per wave and stage: 4 x 8 corpus rows + 4 x 8 query rows
constexpr int W_DEFAULT_LOADERS = 8; // (see the LOADERS template parameter)

struct Tall2Args {
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
    int abl; // diagnostic build, timing only (results are wrong, nothing is admitted): 1 = no staging DMA behind the first
             // stage, 2 = staging only (no LDS reads, no MFMAs)
};

// 128-B rows, eight 16-B chunks: chunk c of row r sits at position c ^ ((r >> 1) & 7), which spreads the 16 lanes of every
// ds_read_b128 group (16 rows that are distinct mod 16, same chunk) over all 64 banks
__device__ __forceinline__ int wswz(int row, int chunk) { return row * W_BK + ((chunk ^ ((row >> 1) & 7)) << 2); }

// 16 B per lane straight into LDS (lane l lands at lds_addr + 16 l); default cache policy: the query rows are re-read by
// every corpus tile, and a corpus line is shared by the query-tile workgroups that run side by side on the XCD
__device__ __forceinline__ void w_dma16(const void *gsrc, uint32_t lds_addr)
{
    uint32_t save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(gsrc), "s"(lds_addr) : "memory");
}
// the same under a wave-uniform EXEC mask (all ones or zero) set inside the asm: the request is part of every wave's
// instruction stream but only the waves whose mask is set issue it.  (A wave-dependent BRANCH around the request -- or two
// copies of the loop -- keeps the accumulators from being promoted to registers: 380 "spills".)  No "memory" clobber: the
// slot being filled is not touched by any compiler-visible access between the barriers that fence it.
__device__ __forceinline__ void w_dma16_masked(const void *gsrc, uint32_t lds_addr, uint32_t mask32)
{
    uint32_t save;
    uint64_t sexec;
    mask32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mask32); // (wave-uniform by construction; makes it an SGPR)
    lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %4\n\ts_mov_b32 exec_hi, %4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0\n\ts_mov_b64 exec, %1"
                 : "=&s"(save), "=&s"(sexec) : "v"(gsrc), "s"(lds_addr), "s"(mask32));
}
// Four requests behind ONE M0 write: the instruction offset moves the LDS destination (and the global address, which the
// callers pre-compensate) by 1 KiB per request.  An M0 write behind a request has to wait until the vector-memory unit has
// taken that request -- with one M0 value per request every request cost the wave its full acceptance time (~230 cycles
// each by the in-kernel stamps), two M0 writes per stage let the four requests of a group queue back to back.
__device__ __forceinline__ void w_dma16x4(const void *g0, const void *g1, const void *g2, const void *g3, uint32_t lds_addr)
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
__device__ __forceinline__ void w_wait_vm0()
{
    __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8) | ((0 >> 4) << 14));
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void w_split8(const f32x4 x0, const f32x4 x1, bf16x8 &hi, bf16x8 &lo)
{
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = h;
        lo[i] = (__bf16)(x[i] - (float)h);
    }
}

// LOADERS (compile time): 8 = every wave requests its share of the next stage, 4 = waves 0-3 request all of it (see the
// DMA set-up in the kernel).
// Tried on this kernel and dropped:
//  * the next stage's requests spread between the MFMAs of the step (two behind every 6 or 12 MFMAs): any inline-asm
//    statement inside the unrolled MFMA block makes hipcc (ROCm 7.2) spill 130-230 registers -- and so do a wave-dependent
//    branch around the requests and two copies of the loop (the accumulator arrays are then not promoted to registers);
//  * an L2 prefetch of the stage after next by a plain global_load_dword into a register nobody reads: the load lands after
//    the asm statement has ended, the allocator has re-used the register by then and a pointer got overwritten (wrong
//    candidates and a memory fault).  An in-flight load needs a destination the compiler cannot touch.
template <int METRIC, int ASPLIT, int LOADERS>
__global__ __launch_bounds__(W_THREADS, 2) void gemm_filter_tall2_kernel(Tall2Args a)
{
    // XCD-aware order as in gemm_filter_kernel: the query tiles of one corpus tile run side by side on one XCD
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int in_xcd = b >> 3;
    const int qt = in_xcd % a.n_q_tiles;
    const int rt = (in_xcd / a.n_q_tiles) * 8 + xcd;
    if (rt >= a.n_row_tiles) return;

    extern __shared__ __attribute__((aligned(16))) float wlds[];
    float *ring = wlds;                                              // [W_NST][A 256x32 | B 256x32]
    float *s_aux = ring + W_NST * W_STAGE_F;                         // [W_BM]
    uint32_t *s_rowid = reinterpret_cast<uint32_t *>(s_aux + W_BM);  // [W_BM]
    uint8_t *s_vis = reinterpret_cast<uint8_t *>(s_rowid + W_BM);    // [W_BM]
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)ring;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1; // 4 x 2 waves: rows 64 wr .. +63, queries 128 wc .. +127
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t row0 = a.row_begin + (int64_t)rt * W_BM;
    const int q0 = qt * W_BN;
    const int64_t last_row = a.row_end - 1;
    const int last_q = a.nq - 1;

    auto corpus_row = [&](int64_t pos) -> int64_t {
        if (pos > last_row) pos = last_row;
        return a.rowmap ? (int64_t)a.rowmap[pos] : pos;
    };
    const int64_t side_ri = corpus_row(row0 + (tid & (W_BM - 1))); // threads 0 .. W_BM-1 carry one side input each

    // DMA sources.  A request fills 8 rows (1 KiB): lane l lands at (row l / 8, chunk POSITION l % 8) and therefore fetches
    // the chunk that belongs there: position ^ ((row >> 1) & 7).
    // LOADERS == 8: every wave requests its 32 corpus rows + 32 query rows (8 requests per stage).
    // LOADERS == 4: waves 0-3 request 64 + 64 rows each (16 requests), waves 4-7 none.  Wave w and wave w + 4 share a SIMD:
    //   while the loader sits in its requests (~1800 cycles per stage: the vector-memory path takes a 1-KiB request every
    //   ~29 cycles per CU and both waves used to queue there at the same time) the other has the matrix pipe to itself,
    //   and the loader then has it while the other waits at the barrier.  The requests are in every wave's instruction
    //   stream, under an EXEC mask that is zero in waves 4-7 (a wave-dependent branch around them costs the accumulators
    //   their registers).
    constexpr int NI = LOADERS == 8 ? 8 : 16;
    const int lw = LOADERS == 8 ? wave : (wave & 3);
    constexpr int RPW = 256 / LOADERS; // rows of each operand a loader wave stages
    const float *src[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int row = lw * RPW + (i % (NI / 2)) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        if (i < NI / 2) {
            src[i] = a.X + corpus_row(row0 + row) * (int64_t)a.D + 4 * c;
        } else {
            int qr = q0 + row;
            if (qr > last_q) qr = last_q;
            src[i] = a.Q + (int64_t)qr * a.D + 4 * c;
        }
    }
    const uint32_t load_mask = (uint32_t)__builtin_amdgcn_readfirstlane((LOADERS == 8 || wave < 4) ? -1 : 0);
    if (LOADERS == 8) { // grouped requests: the instruction offset (i & 3) KiB also moves the global address
#pragma unroll
        for (int i = 0; i < NI; i++) src[i] -= (i & 3) * 256;
    }
    auto issue = [&](int kt) {
        const uint32_t A = ring_base + (uint32_t)(kt & 1) * (W_STAGE_F * 4);
        const uint32_t B = A + W_BM * W_BK * 4;
        if (LOADERS == 8) {
            const int k0 = kt * W_BK;
            w_dma16x4(src[0] + k0, src[1] + k0, src[2] + k0, src[3] + k0, A + (uint32_t)(lw * RPW * W_BK * 4));
            w_dma16x4(src[4] + k0, src[5] + k0, src[6] + k0, src[7] + k0, B + (uint32_t)(lw * RPW * W_BK * 4));
            return;
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const uint32_t dst = (i < NI / 2 ? A : B) + (uint32_t)((lw * RPW + (i % (NI / 2)) * 8) * W_BK * 4);
            w_dma16_masked(src[i] + kt * W_BK, dst, load_mask);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nk = a.D / W_BK; // D % 32 == 0 (launcher)
    issue(0);
    // one burst behind the first stage (not needed before the epilogue): side inputs and thresholds
    const float side_aux = METRIC == METRIC_L2 ? a.norm2[side_ri] : (METRIC == METRIC_COS ? a.rnorm[side_ri] : 0.f);
    uint8_t side_vis = 1;
    if (a.mask) side_vis = a.mask[side_ri];
    float tk[4];
    uint32_t tr[4];
#pragma unroll
    for (int tn = 0; tn < 4; tn++) {
        const int qj = q0 + wc * 128 + tn * 32 + l31;
        uint64_t tau = a.boot ? 0ull : a.cs.tau[qj < a.nq ? qj : a.nq - 1];
        if (qj >= a.nq) tau = 0ull;
        tk[tn] = tau_key_of(tau);
        tr[tn] = entry_row(tau);
    }
    if (tid < W_BM) {
        s_aux[tid] = side_aux;
        s_vis[tid] = (row0 + tid <= last_row && side_vis) ? (uint8_t)1 : (uint8_t)0;
        s_rowid[tid] = (uint32_t)side_ri;
    }

    // one K-step; MORE (compile time): there is a next stage to request
#ifdef LB_DIAG
    unsigned long long pr_wait = 0, pr_bar = 0, pr_issue = 0, pr_t0 = 0, pr_r0 = 0;
    const bool probe = a.abl == 5;
    if (probe) { pr_t0 = __builtin_amdgcn_s_memtime(); pr_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    auto step = [&](int kt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
#ifdef LB_DIAG
        unsigned long long s0 = 0, s1 = 0, s2 = 0;
        if (probe) s0 = __builtin_amdgcn_s_memtime();
#endif
        w_wait_vm0();                 // this wave's part of stage kt has landed (issued one whole K-step ago)
#ifdef LB_DIAG
        if (probe) s1 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier(); // everyone's part is in; everyone is done reading stage kt - 1 (and, at kt = 0, has
                                      // written its side inputs)
        asm volatile("" ::: "memory");
#ifdef LB_DIAG
        if (probe) { s2 = __builtin_amdgcn_s_memtime(); pr_wait += s1 - s0; pr_bar += s2 - s1; }
#endif
#ifdef LB_DIAG
        const bool dma_on = a.abl != 1;
#else
        constexpr bool dma_on = true;
#endif
        if (MORE && dma_on) issue(kt + 1); // stage kt + 1 goes into the slot read at step kt - 1
#ifdef LB_DIAG
        if (probe) pr_issue += __builtin_amdgcn_s_memtime() - s2;
#endif
#ifdef LB_DIAG
        if (a.abl == 2) return;
#endif
        const float *As = ring + (kt & 1) * W_STAGE_F;
        const float *Bs = As + W_BM * W_BK;
        // The stage's two MFMA k-blocks of 16.  Per k-block: both corpus fragments (split once, used by all four query
        // tiles), then the query fragments one tile at a time (16 + 8 fragment registers live instead of 8 + 32).  The
        // second k-block's corpus fragments are fetched and split BEHIND the first query tile of the first k-block, so
        // that their ~50 conversion instructions issue between that k-block's MFMAs (which leave three quarters of the
        // vector issue slots free) instead of in front of the second k-block, where both waves of the SIMD would sit
        // through them at the same time.
        bf16x8 ah[2][2], al[2][2];
        auto load_a = [&](int kb) {
#pragma unroll
            for (int tm = 0; tm < 2; tm++) {
                const int r = wr * 64 + tm * 32 + l31;
                if (ASPLIT == 1) {
                    ah[kb][tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&As[wswz(r, 4 * kb + h)]));
                    al[kb][tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&As[wswz(r, 4 * kb + 2 + h)]));
                } else {
                    const f32x4 x0 = *reinterpret_cast<const f32x4 *>(&As[wswz(r, 4 * kb + 2 * h)]);
                    const f32x4 x1 = *reinterpret_cast<const f32x4 *>(&As[wswz(r, 4 * kb + 2 * h + 1)]);
                    w_split8(x0, x1, ah[kb][tm], al[kb][tm]);
                }
            }
        };
        load_a(0);
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                const int r = wc * 128 + tn * 32 + l31;
                const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Bs[wswz(r, 4 * kb + h)]));
                const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Bs[wswz(r, 4 * kb + 2 + h)]));
#pragma unroll
                for (int tm = 0; tm < 2; tm++) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kb][tm], bh, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb][tm], bl, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb][tm], bh, acc[tm][tn], 0, 0, 0);
                }
                if (kb == 0 && tn == 0) load_a(1);
            }
        }
    };
    for (int kt = 0; kt + 1 < nk; kt++) step(kt, std::true_type{});
    step(nk - 1, std::false_type{});
#ifdef LB_DIAG
    if (probe && lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&g_tall2_probe[0], t1 - pr_t0); // loop cycles
        atomicAdd(&g_tall2_probe[1], r1 - pr_r0); // the same in 100 MHz ticks
        atomicAdd(&g_tall2_probe[2], 1ull);       // waves
        atomicAdd(&g_tall2_probe[3], pr_wait);    // in s_waitcnt vmcnt(0)
        atomicAdd(&g_tall2_probe[4], pr_bar);     // in s_barrier
        atomicAdd(&g_tall2_probe[5], pr_issue);   // issuing the next stage's requests
    }
    if (a.abl != 0 && a.abl != 5) return; // timing-only ablations admit nothing (their sums are not inner products)
#endif

    // ---- epilogue: key + admission, one MFMA row tile (this lane's 16 rows of it) at a time ----------------
    // C layout (32x32): col = lane & 31 (query), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    auto key_of = [&](float dot, float ax) -> float {
        if (METRIC == METRIC_L2) return fmaf(-2.0f, dot, ax);
        if (METRIC == METRIC_COS) return -dot * ax;
        return -dot;
    };
    // workgroup-local admission list, carved from the ring (free: every wave is past the last K-step's reads once the
    // barrier below is behind it)
    constexpr int FL_CAP = 4096;
    uint32_t *s_lcnt = reinterpret_cast<uint32_t *>(ring);              // entries in the list
    uint32_t *s_qcnt = s_lcnt + 1;                                      // [W_BN] of them per query of the tile ...
    uint32_t *s_qbase = s_qcnt + W_BN;                                  // [W_BN] ... and where they start in the query's list
    uint64_t *s_lent = reinterpret_cast<uint64_t *>(ring + 1024);
    uint16_t *s_lq = reinterpret_cast<uint16_t *>(ring + 1024 + 2 * FL_CAP);
    uint16_t *s_lr = s_lq + FL_CAP;                                     // rank of the entry among its query's
    __syncthreads();
    if (tid == 0) *s_lcnt = 0;
    if (tid < W_BN) s_qcnt[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
        float aux[4][4];
        uint32_t rid[4][4];
        uint32_t vbits = 0; // bit (g*4 + e): row visible (in range and not masked out)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int lr = wr * 64 + tm * 32 + 8 * g + 4 * h; // 4 consecutive local rows
            const f32x4 av = *reinterpret_cast<const f32x4 *>(&s_aux[lr]);
            const uint4 rv = *reinterpret_cast<const uint4 *>(&s_rowid[lr]);
            const uint32_t vv = *reinterpret_cast<const uint32_t *>(&s_vis[lr]);
            aux[g][0] = av.x; aux[g][1] = av.y; aux[g][2] = av.z; aux[g][3] = av.w;
            rid[g][0] = rv.x; rid[g][1] = rv.y; rid[g][2] = rv.z; rid[g][3] = rv.w;
            const uint32_t nib = (vv & 1u) | ((vv >> 7) & 2u) | ((vv >> 14) & 4u) | ((vv >> 21) & 8u);
            vbits |= nib << (g * 4);
        }
#pragma unroll
        for (int tn = 0; tn < 4; tn++) {
            const int qj = q0 + wc * 128 + tn * 32 + l31;
            const bool qok = qj < a.nq;
            uint64_t *list = a.cs.lists + (size_t)(qok ? qj : 0) * a.cs.cap;
            if (a.boot) { // sample pass: the entry of position p goes to list[p - row_begin]
                if (qok) {
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int64_t rbase = row0 + wr * 64 + tm * 32 + 8 * g + 4 * h;
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
            if (bits) { // workgroup-local list first: two LDS atomics per lane with admissions
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
        if (tid < W_BN) {
            const uint32_t n = s_qcnt[tid];
            s_qbase[tid] = n ? atomicAdd(&a.cs.cnt[q0 + tid], n) : 0u; // (n != 0 implies a real query)
        }
        __syncthreads();
        const uint32_t total = *s_lcnt < (uint32_t)FL_CAP ? *s_lcnt : (uint32_t)FL_CAP;
        for (uint32_t i = tid; i < total; i += W_THREADS) {
            const uint64_t ent = s_lent[i];
            if (ent == kEntryMax) continue;
            const int ql = (int)s_lq[i];
            const uint32_t pos = s_qbase[ql] + (uint32_t)s_lr[i];
            if (pos < a.cs.cap) a.cs.lists[(size_t)(q0 + ql) * a.cs.cap + pos] = ent;
        }
    }
}

} // namespace

void read_tall2_probe(unsigned long long out[8], bool reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tall2_probe), 8 * sizeof(unsigned long long));
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tall2_probe), z, sizeof z);
    }
}

// Requires D % 32 == 0, 16-B aligned X / Q; Q is the split image of the batch; asplit: 1 = X is the split image of
// the corpus, 2 = X is the plain f32 corpus.
void launch_gemm_filter_tall2(int metric, const float *X, const float *norm2, const float *rnorm, int64_t row_begin,
                              int64_t row_end, int D, const float *Qs, int nq, const uint8_t *mask, const uint32_t *rowmap,
                              CandState cs, bool boot, int asplit, hipStream_t s)
{
    if (row_end <= row_begin || nq <= 0) return;
    Tall2Args a;
    a.rowmap = rowmap;
    a.X = X; a.norm2 = norm2; a.rnorm = rnorm; a.row_begin = row_begin; a.row_end = row_end; a.D = D;
    a.Q = Qs; a.nq = nq; a.mask = mask; a.cs = cs; a.boot = boot ? 1 : 0;
    static const int abl = lb_tunable("LB_TALL2_ABL", 0);
    a.abl = abl;
    static const int opt = lb_tunable("LB_TALL2_LOADERS", W_DEFAULT_LOADERS);
    a.n_row_tiles = (int)((row_end - row_begin + W_BM - 1) / W_BM);
    a.n_q_tiles = (nq + W_BN - 1) / W_BN;
    const int groups = (a.n_row_tiles + 7) / 8;
    dim3 grid((unsigned)(groups * 8 * a.n_q_tiles));
    const size_t shmem = (size_t)W_NST * W_STAGE_F * 4 + W_BM * 4 + W_BM * 4 + W_BM;
#define LB_TALL2(M, SP, O)                                                                                          \
    do {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_filter_tall2_kernel<M, SP, O>),              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); /* per device */         \
        hipLaunchKernelGGL((gemm_filter_tall2_kernel<M, SP, O>), grid, dim3(W_THREADS), shmem, s, a);               \
    } while (0)
#ifdef LB_DIAG
#define LB_TALL2_O(M, SP)                    \
    do {                                     \
        if (opt == 4) LB_TALL2(M, SP, 4);    \
        else LB_TALL2(M, SP, 8);             \
    } while (0)
#else
#define LB_TALL2_O(M, SP) LB_TALL2(M, SP, W_DEFAULT_LOADERS)
#endif
#define LB_TALL2_M(M)                      \
    do {                                   \
        if (asplit == 1) LB_TALL2_O(M, 1); \
        else LB_TALL2_O(M, 2);             \
    } while (0)
    if (metric == METRIC_L2) LB_TALL2_M(METRIC_L2);
    else if (metric == METRIC_COS) LB_TALL2_M(METRIC_COS);
    else LB_TALL2_M(METRIC_DOT);
#undef LB_TALL2_M
#undef LB_TALL2_O
#undef LB_TALL2
    (void)opt;
}

} // namespace lb
