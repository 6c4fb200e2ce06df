// kernels_gemm_tall16.hip -- candidate generation with ONE fp16 MFMA product per (row, query, k): 256 rows x 256 queries
// per workgroup, the corpus read as f32 and rounded to fp16 in registers, the queries as a pre-scaled fp16 image.
//
// Same contract as the other candidate kernels (kernels_gemm_tall2.hip): S = X_tile . Q_tile^T, metric key and admission
// test fused into the epilogue; ranking semantics of BruteForceIndex.SearchVectors
// (internal/store/adaptive_index.go:161-225); the keys are CANDIDATE keys only -- every reported distance comes from the
// exact f32 re-rank, and the re-rank's containment proof (index.hip) decides with THIS contraction's error bound whether
// the candidate list provably holds the true top-k; a query for which it does not is redone by the exact scan.
//
// Why one product is enough.  The split-bf16 contraction (three MFMA products, 2^-17 relative) is far more accurate than the
// proof needs: it only has to separate the k-th from the kc-th best row (kc = 256 for k = 100), a gap of several 1e-3
// of |q||x| on embedding-like data.  fp16 operands rounded to nearest carry 2^-11 each, so
//     | sum fp16(s q_i) fp16(x_i) / s  -  q.x |  <=  (2^-10 + 2^-21) sum|q_i x_i|            (normal range)
//                                                  + 2^-25 sqrt(D) |q| (1 + |x|)             (subnormal range, see below)
//                                                  + (D + 8) 2^-24 sum|q_i x_i|              (f32 accumulation)
// i.e. gamma ~ 1.1e-3 of |q||x| at D = 768 -- inside the gap with a factor to spare (0 fallbacks of 1024 on the benchmark
// data) -- for ONE third of the matrix work and half the query-side staging bytes of the split contraction.
//
// Range.  Each query is scaled by a power of two s so that 1 <= |s q| < 2 (exact; the epilogue multiplies the product by 1/s,
// also exact), so query elements never overflow and their subnormal rounding (absolute 2^-25) is negligible against |s q| >= 1.
// Corpus elements are used as they are: the route is offered only while max |x| <= 2^13 (no overflow: fp16 max 65504) and the
// smallest non-zero row norm is >= 2^-6, which bounds the subnormal term above by 2^-25 sqrt(D) (2^6 + 1) of |q||x|
// (index.hip tracks both norms at Add).
//
// Shape: as kernels_gemm_tall2.hip (8 waves = 4 x 2, a wave owns 64 rows x 128 queries, K-step of 32 = one 128-B line per
// corpus row).  Three kernels, one contraction:
//   gemm_filter_tall16p_kernel   persistent (one workgroup per CU walks its corpus tiles as one flat pipeline); corpus from the
//                                index's fp16 copy (stage 16 + 16 KB, four-stage ring, fragment reads pipelined behind a
//                                mid-step barrier) or from the f32 rows rounded in registers (stage 32 + 16 KB, three stages)
//   gemm_filter_narrow16p_kernel the same pipeline on a 256 x 64 tile over the fp16 copy, six stages: 1 .. 64 queries at the
//                                copy's HBM stream
//   gemm_filter_tall16_kernel    one workgroup per tile, f32 rows through a row map / mask: filtered searches, and batches
//                                with more query tiles than an XCD has workgroup slots
// What each design decision bought is in LABNOTES.md 4.2 (measured with tools/experiments/dma_patterns.hip).
#include "lb_device.h"

#include <type_traits>

namespace lb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int H_BM = 256, H_BN = 256, H_BK = 32;
// The corpus image is blocked in planes of H_XP dimensions: [Dp / H_XP][cap][H_XP] fp16.  H_XP = 32 (the K-step): the 64 B a
// K-step needs of 16 consecutive rows are one KiB of whole lines.  H_XP = 64 (a row's piece = one whole 128-B line, the two
// K-steps of a plane taking half each) was built for row lists -- HBM is fetched in whole lines, so a filtered view pays for
// the neighbouring row's half -- and measured (round 4, 1.25M x 1536, 10 % visible): the gather pass 119 -> 94 us at 32
// queries, 234 -> 216 at 256, but the UNFILTERED stream 234 -> 339 us per 1M x 768 (half-line requests under the
// non-temporal policy fetch every line twice).  Not worth it: 32 stays; -DLB_XH_PLANE=64 rebuilds the other.
#ifndef LB_XH_PLANE
#define LB_XH_PLANE 32
#endif
constexpr int H_XP = LB_XH_PLANE, H_XROW = H_XP * 2, H_KPP = H_XP / H_BK; // plane dims, bytes of a row's piece, K-steps per plane
__device__ __forceinline__ int64_t h_xoff(int ke, int64_t plane_bytes) // byte offset of K-step ke within a row's pieces
{
    return H_KPP == 1 ? ke * plane_bytes : (int64_t)(ke / H_KPP) * plane_bytes + (ke % H_KPP) * (H_BK * 2);
}
constexpr int H_THREADS = 512;
constexpr int H_A_BYTES = H_BM * H_BK * 4;            // 32 KB: corpus rows as f32
constexpr int H_B_BYTES = H_BN * H_BK * 2;            // 16 KB: query rows as fp16
constexpr int H_STAGE_BYTES = H_A_BYTES + H_B_BYTES;  // 48 KB
constexpr int H_NST = 3;
constexpr int H_NI = 6; // DMA requests per wave and stage: 4 x 8 corpus rows + 2 x 16 query rows

#ifdef LB_DIAG
__device__ unsigned long long g_tall16_probe[8]; // ABL == 5: cycle stamps summed over waves (tools/tall16_probe.py)
#endif

struct Tall16Args {
    const float *X;
    const float *norm2;
    const float *rnorm;
    int64_t row_begin, row_end;
    int D;
    const _Float16 *Qh;  // [D / 32][nq][32] fp16 image of the batch (K-blocked: the 64 B a K-step needs of every query lie side
                         // by side, so a request of 16 query rows is ONE KiB of whole lines instead of 16 half lines: -10 %
                         // on the kernel), each query scaled by a power of two to 1 <= |q| < 2
    const float *qinv;   // [nq] 1 / that scale (exact)
    const float *qnrm;   // [nq] an upper bound of |q| (dot product, persistent forms: the lower-bound key, see gsum)
    float gsum;          // dot product, persistent forms: gamma_a + gamma_o.  The candidate key is -(q.x)~ / G - |x| with
                         // G = gsum |q| -- a lower bound of -q.x (in units of G) whatever the row's norm, so that a few very long
                         // rows, whose products are uncertain by gamma_a |q||x|, sort to the FRONT of the list and are scored
                         // exactly instead of widening every other row's error bound (kernels_finish.hip: dot_lb)
    int nq;              // queries of this launch (Qh, qinv and the candidate state start at its first one)
    int q_stride;        // query rows per K-block plane of Qh (the whole batch; a launch may cover a part of it)
    const uint8_t *mask;
    const uint32_t *rowmap;
    CandState cs;
    int n_row_tiles, n_q_tiles;
    int boot;
    int abl; // diagnostic build: timing-only ablations of the persistent form (6 = no epilogue, 7 = no flush)
    int rot; // persistent form: the query tiles of a corpus tile walk K rotated by rot K-steps against each other
    const _Float16 *Xh; // or null: fp16 image of the corpus, [Dp / H_XP][xh_cap][H_XP] (index.hip: sync_f16_image)
    int64_t xh_cap;     // rows per K-block plane of Xh
    uint32_t gstride;   // persistent forms, sample pass: 0 = position p is row row_begin + p; else the positions are granules of 16
                        // consecutive rows, granule j starting at row row_begin + j * gstride (an evenly spaced sample whose
                        // requests are still whole KiB of the K-blocked image)
    // one-tile form, TAUIN instantiation: the thresholds are computed INSIDE the launch from the sample the launch before left
    // in lists[q][0 .. tin_count) (see the kernel); tau[q] = 0 means "not out yet".  A wait that gives up stores tin_tag to
    // the pinned tin_fail.
    uint32_t tin_count, tin_tag, tin_dr;
    int tin_m, tin_order;
    uint32_t *tin_fail;
    const float *tin_Q; // f32 queries [nq][D] and (cosine, or null) where their exact squared norms go
    float *tin_qna;
    int qsplit; // one-tile form, sample pass of a batch beyond 128 queries: 0, or the number of 128-query windows the launch's
                // workgroups divide into
};

// Persistent forms: the launch's positions [0, n_pos) are dealt to the workgroups (narrow form) / workgroup groups (256-query
// form) as CONTIGUOUS ranges whose bounds are multiples of 16 -- every range takes the same time whatever n_pos is (tiles dealt
// round-robin left 16 tiles to some workgroups and 15 to others at 1M rows: 4 % of the pass), and a short launch (the 8192-row
// sample) spreads over all of them.  A range's last tile is partial: positions beyond it read its last row again (L2 hits)
// and carry a NaN side input.
// sample / bootstrap epilogue: the entries of positions p .. p + 3 (p a multiple of 4) of one query's list.  vec: the list is
// 16-B aligned at every such p (cap a multiple of 4) -- two 16-B stores instead of four of 8 (a store instruction of this
// epilogue touches 32 lists: a quarter of the instructions, 32-B segments instead of 8-B ones)
__device__ __forceinline__ void h_store_entries4(uint64_t *list, uint32_t p, uint32_t last_pos, const uint64_t (&ent)[4], bool vec)
{
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    if (vec && p + 3u <= last_pos) {
        u64x2 *d = reinterpret_cast<u64x2 *>(list + p);
        d[0] = u64x2{ent[0], ent[1]};
        d[1] = u64x2{ent[2], ent[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (p + (uint32_t)e <= last_pos) list[p + (uint32_t)e] = ent[e];
    }
}

__device__ __forceinline__ void h_range(uint32_t n_pos, int gi, int ng, uint32_t &lo, uint32_t &hi)
{
    const uint32_t q = n_pos / (uint32_t)ng, r = n_pos % (uint32_t)ng; // n_pos gi / ng = q gi + r gi / ng (no 64-bit division)
    lo = (q * (uint32_t)gi + (r * (uint32_t)gi) / (uint32_t)ng) & ~15u;
    hi = gi + 1 < ng ? ((q * (uint32_t)(gi + 1) + (r * (uint32_t)(gi + 1)) / (uint32_t)ng) & ~15u) : n_pos;
    // (the divisions run on the vector unit: tell the compiler the results are wave-uniform -- loop counts and the M0
    // operands of the LDS-DMA requests derive from them)
    lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
}
__device__ __forceinline__ uint32_t h_rowof(uint32_t pos, uint32_t gstride) { return gstride ? (pos >> 4) * gstride + (pos & 15u) : pos; }

// corpus rows: 128 B, eight 16-B chunks, chunk c of row r at position c ^ ((r >> 1) & 7)   (byte offset in the A region)
__device__ __forceinline__ int haswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// query rows: 64 B, four 16-B chunks, chunk c of row r at position c ^ ((r >> 2) & 3)      (byte offset in the B region)
__device__ __forceinline__ int hbswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// requests behind ONE M0 write; the instruction offset moves the LDS destination and the global address alike (the callers
// pre-compensate the sources)
template <bool NT>
__device__ __forceinline__ void h_dma16x4(const void *g0, const void *g1, const void *g2, const void *g3, uint32_t lds_addr)
{
    uint32_t save;
    if (NT) // non-temporal: a corpus line that only this workgroup will read (one or two query tiles per corpus tile)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off nt\n\t"
                     "global_load_lds_dwordx4 %2, off offset:1024 nt\n\t"
                     "global_load_lds_dwordx4 %3, off offset:2048 nt\n\t"
                     "global_load_lds_dwordx4 %4, off offset:3072 nt\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\t"
                     "global_load_lds_dwordx4 %2, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %3, off offset:2048\n\t"
                     "global_load_lds_dwordx4 %4, off offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void h_dma16x2(const void *g0, const void *g1, uint32_t lds_addr)
{
    uint32_t save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(g0), "v"(g1), "s"(lds_addr) : "memory");
}
// ONE request (SPREAD: a stage's six requests are issued one at a time between the MFMAs of the step before)
template <bool NT>
__device__ __forceinline__ void h_dma16(const void *g, uint32_t lds_addr)
{
    uint32_t save;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off nt\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(g), "s"(lds_addr));
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(g), "s"(lds_addr));
}
template <int N>
__device__ __forceinline__ void h_wait_vmcnt()
{
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ f16x8 h_cvt8(const f32x4 x0, const f32x4 x1)
{
    f16x8 r; // round to nearest even (v_cvt_f16_f32 under the default rounding mode)
    r[0] = (_Float16)x0.x; r[1] = (_Float16)x0.y; r[2] = (_Float16)x0.z; r[3] = (_Float16)x0.w;
    r[4] = (_Float16)x1.x; r[5] = (_Float16)x1.y; r[6] = (_Float16)x1.z; r[7] = (_Float16)x1.w;
    return r;
}

// NT: the corpus requests carry the non-temporal policy (few query tiles per corpus tile: the line is not read again)
// SPREAD: the requests of stage k + 2 go out one by one between the MFMAs of step k instead of in one burst behind the barrier
// (in a burst all eight waves queue on the CU's one address unit while the matrix pipe idles)
// ABL (diagnostic build; the product instantiates 0 only): timing-only ablations 1 = no requests inside the loop, 2 = requests,
// waits and barriers only (no LDS reads, no MFMAs), 3 = corpus requests only; 5 = cycle stamps
template <int METRIC, bool NT, bool SPREAD, int ABL = 0>
__global__ __launch_bounds__(H_THREADS, 2) void gemm_filter_tall16_kernel(Tall16Args a)
{
    // XCD-aware order as in gemm_filter_kernel: the query tiles of one corpus tile run side by side on one XCD
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int in_xcd = b >> 3;
    const int qt = in_xcd % a.n_q_tiles;
    const int rt = (in_xcd / a.n_q_tiles) * 8 + xcd;
    if (rt >= a.n_row_tiles) return;

    unsigned long long pr_start = 0;
    if (ABL == 5) pr_start = __builtin_amdgcn_s_memtime();
    extern __shared__ __attribute__((aligned(16))) unsigned char hlds[];
    unsigned char *ring = hlds;                                                        // [H_NST][A 32 KB | B 16 KB]
    float *s_aux = reinterpret_cast<float *>(ring + H_NST * H_STAGE_BYTES);            // [H_BM]
    uint32_t *s_rowid = reinterpret_cast<uint32_t *>(s_aux + H_BM);                    // [H_BM]
    uint8_t *s_vis = reinterpret_cast<uint8_t *>(s_rowid + H_BM);                      // [H_BM]
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ring;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1; // 4 x 2 waves: rows 64 wr .. +63, queries 128 wc .. +127
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t row0 = a.row_begin + (int64_t)rt * H_BM;
    const int q0 = qt * H_BN;
    const int64_t last_row = a.row_end - 1;
    const int last_q = a.nq - 1;

    auto corpus_row = [&](int64_t pos) -> int64_t {
        if (pos > last_row) pos = last_row;
        return a.rowmap ? (int64_t)a.rowmap[pos] : pos;
    };
    const int64_t side_ri = corpus_row(row0 + (tid & (H_BM - 1))); // threads 0 .. H_BM-1 carry one side input each

    // DMA sources (byte pointers).  Corpus: request i < 4 of this wave fills rows 32 wave + 8 i .. +7 (lane l: row l / 8, chunk
    // position l % 8).  Queries: request j < 2 fills rows 32 wave + 16 j .. +15 (lane l: row l / 4, chunk position l % 4).
    // The instruction offsets of the grouped requests (i KiB) are taken off the sources here.
    const unsigned char *srcA[4], *srcB[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        srcA[i] = reinterpret_cast<const unsigned char *>(a.X + corpus_row(row0 + row) * (int64_t)a.D) + 16 * c - 1024 * i;
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int row = wave * 32 + j * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        int qr = q0 + row;
        if (qr > last_q) qr = last_q;
        srcB[j] = reinterpret_cast<const unsigned char *>(a.Qh + (int64_t)qr * H_BK) + 16 * c - 1024 * j;
    }
    const int64_t kb_stride = (int64_t)a.q_stride * (H_BK * 2); // bytes from one K-block of the query image to the next
    auto issue = [&](int kt) {
        const uint32_t A = ring_base + (uint32_t)(kt % H_NST) * H_STAGE_BYTES;
        const uint32_t B = A + H_A_BYTES;
        const int ka = kt * (H_BK * 4);                       // byte offset along a corpus row
        const int64_t kb = kt * kb_stride;                    // the query image's K-block
        h_dma16x4<NT>(srcA[0] + ka, srcA[1] + ka, srcA[2] + ka, srcA[3] + ka, A + (uint32_t)(wave * 32 * 128));
        h_dma16x2(srcB[0] + kb, srcB[1] + kb, B + (uint32_t)(wave * 32 * 64));
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nk = a.D / H_BK; // D % 32 == 0 (launcher)
    issue(0);
    if (nk > 1) issue(1);
    // one burst behind the first stages (not needed before the epilogue): side inputs, thresholds, query scales
    const float side_aux = METRIC == METRIC_L2 ? a.norm2[side_ri] : (METRIC == METRIC_COS ? a.rnorm[side_ri] : 0.f);
    uint8_t side_vis = 1;
    if (a.mask) side_vis = a.mask[side_ri];
    float tk[4], qs[4];
#pragma unroll
    for (int tn = 0; tn < 4; tn++) {
        const int qj = q0 + wc * 128 + tn * 32 + l31;
        const int qc = qj < a.nq ? qj : a.nq - 1;
        uint64_t tau = a.boot ? 0ull : a.cs.tau[qc];
        if (qj >= a.nq) tau = 0ull;
        tk[tn] = tau_key_of(tau);
        qs[tn] = a.qinv[qc];
    }
    if (tid < H_BM) {
        s_aux[tid] = side_aux;
        s_vis[tid] = (row0 + tid <= last_row && side_vis) ? (uint8_t)1 : (uint8_t)0;
        s_rowid[tid] = (uint32_t)side_ri;
    }
    // (the loads above were waited for by the compiler before their use; from here on only the DMA requests are in flight,
    // H_NI per wave and stage, and the compiler sees none of them)

    unsigned long long pr_wait = 0, pr_bar = 0, pr_t0 = 0, pr_r0 = 0;
    if (ABL == 5) { pr_t0 = __builtin_amdgcn_s_memtime(); pr_r0 = __builtin_amdgcn_s_memrealtime(); }
    // one K-step; ISSUE: the requests of stage kt + 2 are made during it
    auto step = [&](int kt, auto issue_c) {
        constexpr bool ISSUE = decltype(issue_c)::value && ABL != 1;
        unsigned long long s0 = 0, s1 = 0;
        if (ABL == 5) s0 = __builtin_amdgcn_s_memtime();
        // stage kt has landed once at most the requests of stage kt + 1 are outstanding
        if (kt + 1 < nk) h_wait_vmcnt<(ABL == 3 ? 4 : H_NI)>();
        else h_wait_vmcnt<0>();
        if (ABL == 5) s1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier(); // everyone's part of stage kt is in; everyone is done reading stage kt - 1
        asm volatile("" ::: "memory");
        if (ABL == 5) { pr_wait += s1 - s0; pr_bar += __builtin_amdgcn_s_memtime() - s1; }
        if (ISSUE && !SPREAD) issue(kt + 2); // into the slot read at step kt - 1
        const uint32_t A2 = ring_base + (uint32_t)((kt + 2) % H_NST) * H_STAGE_BYTES + (uint32_t)(wave * 32 * 128);
        const uint32_t B2 = ring_base + (uint32_t)((kt + 2) % H_NST) * H_STAGE_BYTES + H_A_BYTES + (uint32_t)(wave * 32 * 64);
        const int ka2 = (kt + 2) * (H_BK * 4);
        const int64_t kb2 = (kt + 2) * kb_stride;
        const unsigned char *As = ring + (kt % H_NST) * H_STAGE_BYTES;
        const unsigned char *Bs = As + H_A_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; kb++) { // the stage's two MFMA k-blocks of 16
            f16x8 af[2];
            if (ABL != 2) {
#pragma unroll
                for (int tm = 0; tm < 2; tm++) {
                    const int r = wr * 64 + tm * 32 + l31;
                    // lane half h supplies k = 16 kb + 8 h .. + 7: f32 chunks 4 kb + 2 h and the next
                    const f32x4 x0 = *reinterpret_cast<const f32x4 *>(As + haswz(r, 4 * kb + 2 * h));
                    const f32x4 x1 = *reinterpret_cast<const f32x4 *>(As + haswz(r, 4 * kb + 2 * h + 1));
                    af[tm] = h_cvt8(x0, x1);
                }
            }
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                if (ABL != 2) {
                    const int r = wc * 128 + tn * 32 + l31;
                    // 8 fp16 = one 16-B chunk: k = 16 kb + 8 h .. + 7 is chunk 2 kb + h of the 64-B row
                    const f16x8 bf = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(Bs + hbswz(r, 2 * kb + h)));
#pragma unroll
                    for (int tm = 0; tm < 2; tm++)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tm], bf, acc[tm][tn], 0, 0, 0);
                }
                if (ISSUE && SPREAD) { // slots 0 1 2 . 4 5 6 . of the step's eight MFMA pairs: corpus 0..3, queries 0..1
                    const int slot = kb * 4 + tn;
                    if (slot < 3) h_dma16<NT>(srcA[slot] + 1024 * slot + ka2, A2 + 1024u * slot);
                    else if (slot == 4) h_dma16<NT>(srcA[3] + 1024 * 3 + ka2, A2 + 1024u * 3);
                    else if (slot == 5 && ABL != 3) h_dma16<false>(srcB[0] + kb2, B2);
                    else if (slot == 6 && ABL != 3) h_dma16<false>(srcB[1] + 1024 + kb2, B2 + 1024u);
                }
            }
        }
    };
    int kt = 0;
    for (; kt + 2 < nk; kt++) step(kt, std::true_type{});
    for (; kt < nk; kt++) step(kt, std::false_type{});
#ifdef LB_DIAG
    if (ABL == 5 && lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&g_tall16_probe[0], t1 - pr_t0); // loop cycles
        atomicAdd(&g_tall16_probe[1], r1 - pr_r0); // the same in 100 MHz ticks
        atomicAdd(&g_tall16_probe[2], 1ull);       // waves
        atomicAdd(&g_tall16_probe[3], pr_wait);    // in s_waitcnt vmcnt
        atomicAdd(&g_tall16_probe[4], pr_bar);     // in s_barrier
        atomicAdd(&g_tall16_probe[5], pr_t0 - pr_start); // prologue (kernel start -> loop start)
    }
    const unsigned long long pr_loop_end = ABL == 5 ? __builtin_amdgcn_s_memtime() : 0ull;
#endif

    // ---- epilogue: key + admission, one MFMA row tile (this lane's 16 rows of it) at a time ----------------
    // C layout (32x32): col = lane & 31 (query), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    // dot = acc * qs (the query's scale taken back out: exact, a power of two)
    auto key_of = [&](float acc_v, float qsc, float ax) -> float {
        const float dot = acc_v * qsc;
        if (METRIC == METRIC_L2) return fmaf(-2.0f, dot, ax);
        if (METRIC == METRIC_COS) return -dot * ax;
        return -dot;
    };
    // workgroup-local admission list, carved from the ring (free: every wave is past the last K-step's reads once the
    // barrier below is behind it)
    constexpr int FL_CAP = 4096;
    float *ringf = reinterpret_cast<float *>(ring);
    uint32_t *s_lcnt = reinterpret_cast<uint32_t *>(ringf);             // entries in the list
    uint32_t *s_qcnt = s_lcnt + 1;                                      // [H_BN] of them per query of the tile ...
    uint32_t *s_qbase = s_qcnt + H_BN;                                  // [H_BN] ... and where they start in the query's list
    uint64_t *s_lent = reinterpret_cast<uint64_t *>(ringf + 1024);
    uint16_t *s_lq = reinterpret_cast<uint16_t *>(ringf + 1024 + 2 * FL_CAP);
    uint16_t *s_lr = s_lq + FL_CAP;                                     // rank of the entry among its query's
    __syncthreads();
    if (tid == 0) *s_lcnt = 0;
    if (tid < H_BN) s_qcnt[tid] = 0;
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
                                    ((vbits >> (g * 4 + e)) & 1u)
                                        ? pack_entry(key_of(acc[tm][tn][4 * g + e], qs[tn], aux[g][e]), rid[g][e])
                                        : kEntryMax;
                    }
                }
                continue;
            }
            // Admission test, two VALU operations per element beside the key: the SIGN of (tau_key - key), shifted into a mask.
            // It admits key <= tau_key -- the exact rule (key < tau_key, or equal keys and a lower row) plus the ties with a
            // higher row: a superset, which the select behind this launch orders exactly as it orders every other entry
            // (tau only bounds the list; an entry at tau displaces nothing).  Both operands of the subtraction are the
            // rounded f32 values the exact rule compares, so its sign is theirs.
            uint32_t rej = 0; // bit i: element i is beyond the threshold (or a padded query: tk NaN with the sign set)
#pragma unroll
            for (int i = 15; i >= 0; i--) { // element 15 first: each step shifts the mask left, element i ends in bit i
                const float key = key_of(acc[tm][tn][i], qs[tn], aux[i >> 2][i & 3]);
                const float d = tk[tn] - key;
                rej = __builtin_amdgcn_alignbit(rej, __builtin_bit_cast(uint32_t, d), 31); // (rej << 1) | sign(d)
            }
            uint32_t bits = ~rej & 0xffffu;
            if (!qok) bits = 0;
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
                                s_lent[lp] = pack_entry(key_of(acc[tm][tn][4 * g + e], qs[tn], aux[g][e]), rid[g][e]);
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
                            if (pos < a.cs.cap)
                                list[pos] = pack_entry(key_of(acc[tm][tn][4 * g + e], qs[tn], aux[g][e]), rid[g][e]);
                            pos++;
                        }
            }
        }
    }
    if (!a.boot) { // flush the workgroup-local admissions: ONE returning global atomic per query of the tile, all in flight
        __syncthreads();
        if (tid < H_BN) {
            const uint32_t n = s_qcnt[tid];
            s_qbase[tid] = n ? atomicAdd(&a.cs.cnt[q0 + tid], n) : 0u; // (n != 0 implies a real query)
        }
        __syncthreads();
        const uint32_t total = *s_lcnt < (uint32_t)FL_CAP ? *s_lcnt : (uint32_t)FL_CAP;
        for (uint32_t i = tid; i < total; i += H_THREADS) {
            const uint64_t ent = s_lent[i];
            if (ent == kEntryMax) continue;
            const int ql = (int)s_lq[i];
            const uint32_t pos = s_qbase[ql] + (uint32_t)s_lr[i];
            if (pos < a.cs.cap) a.cs.lists[(size_t)(q0 + ql) * a.cs.cap + pos] = ent;
        }
    }
#ifdef LB_DIAG
    if (ABL == 5 && lane == 0) atomicAdd(&g_tall16_probe[6], __builtin_amdgcn_s_memtime() - pr_loop_end); // epilogue
#endif
}

// ---- persistent form: one workgroup per CU walks its corpus tiles as ONE flat pipeline ------------------------------------
// (unfiltered searches: no row map, no mask).  What it removes from every tile of the kernel above: the prologue (a round trip
// for the side inputs before the first K-step), the pipeline refill (the first stages of the NEXT tile are requested during
// the last K-steps and land under the epilogue) and the reload of the per-query thresholds and scales (a workgroup keeps its
// query tile).  Workgroup b sits in slot b >> 3 of XCD b & 7; the slots of an XCD form groups of n_q_tiles workgroups that walk
// the same corpus tiles side by side (they share the corpus lines through that XCD's L2), group g taking the tiles
// 8 (g + groups i) + xcd, i = 0, 1, ...  A stage beyond the last tile re-reads the last stage (its ring slot is free by
// then), so every K-step issues the same requests and every wait is the same s_waitcnt vmcnt(N).
// AIMG: the corpus comes from its fp16 image (the index keeps one while memory allows): a stage is 16 KB of corpus + 16 KB of
// queries, the ring FOUR stages deep and a stage is requested three K-steps ahead -- half the corpus bytes to stage, and
// a third more time for a line that four CUs ask for at once to arrive (LABNOTES.md 4.2).  The image is K-blocked like the
// query image: the 64 B a K-step needs of 16 consecutive rows are one KiB of whole lines.
// BOOT: the launch stores every position's entry (bootstrap chunk of the classic schedule, or -- gstride != 0 -- the sample pass)
// MAPPED: a filtered view -- the positions index a.rowmap; see the one-tile form below for how the kernel gathers
template <int METRIC, bool NT, bool AIMG, bool BOOT, bool MAPPED = false>
__global__ __launch_bounds__(H_THREADS, 2) void gemm_filter_tall16p_kernel(Tall16Args a, int spx)
{
    constexpr int NST = AIMG ? 4 : 3;                       // ring stages
    constexpr int A_BYTES = AIMG ? H_BM * H_BK * 2 : H_A_BYTES;
    constexpr int STAGE = A_BYTES + H_B_BYTES;
    constexpr int NPA = AIMG ? 2 : 4;                       // corpus requests per wave and stage
    constexpr int NPS = NPA + 2;                            // requests per wave and stage
    constexpr int DIST = NST - 1;                           // a stage is requested DIST K-steps before its use
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int nqt = a.n_q_tiles;
    const int gpx = spx / nqt; // >= 1 (launcher)
    const int group = slot / nqt, qt = slot - group * nqt;
    if (group >= gpx) return;
    const int jtop = a.n_row_tiles - 1 - xcd;
    if (jtop < 0 || group > (jtop >> 3)) return;
    const int n_my = ((jtop >> 3) - group) / gpx + 1; // corpus tiles of this workgroup

    extern __shared__ __attribute__((aligned(16))) unsigned char hlds[];
    unsigned char *ring = hlds;                                                     // [NST][A | B 16 KB]
    float *s_auxp = reinterpret_cast<float *>(ring + NST * STAGE);                  // [2][512]: side input of the tile, double-buffered
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ring;
    const uint32_t aux_base = ring_base + (uint32_t)(NST * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int q0 = qt * H_BN;
    const int last_q = a.nq - 1;
    const uint32_t last_pos = (uint32_t)(a.row_end - a.row_begin - 1); // positions of this launch: corpus row = row_begin + position (sample pass: h_rowof)
    const uint32_t gstride = BOOT ? a.gstride : 0u;
    const bool boot_vec = BOOT && (a.cs.cap & 3u) == 0u && (reinterpret_cast<uintptr_t>(a.cs.lists) & 15) == 0; // (h_store_entries4)
    // dot product: |x| of the lower-bound key, padded for the f32 roundings of the key's own fma (they are relative to the first
    // term, up to 1 / (gamma_a + gamma_o) times |x|) and of the stored norm
    const float dot_pad = 1.0f + (a.gsum > 0.f ? 6.0e-7f / a.gsum : 0.f) + 1.0e-5f + 1.05f * (float)(a.D + 8) * 5.9604645e-8f;
    const unsigned char *Xb = reinterpret_cast<const unsigned char *>(a.X + a.row_begin * (int64_t)a.D);
    const int64_t row_bytes = (int64_t)a.D * 4;
    const float *auxg = METRIC == METRIC_COS ? a.rnorm + (MAPPED ? 0 : a.row_begin) : a.norm2 + (MAPPED ? 0 : a.row_begin); // (dot: |x| for the lower-bound key)
    auto rt_of = [&](int i) { return (uint32_t)((group + gpx * i) * 8 + xcd) * H_BM; }; // first position of tile i
    // MAPPED: row ids of three tiles, [3][512] (every wave asks for 64: entries 256 .. 511 repeat 0 .. 255, as the side inputs do)
    const uint32_t *s_rowid = reinterpret_cast<const uint32_t *>(s_auxp + 2 * 512);
    const uint32_t rowid_base = aux_base + 4096u;
    auto rowid_request = [&](int t) {
        uint32_t pos = rt_of(t) + (uint32_t)((wave & 3) * 64 + lane);
        if (pos > last_pos) pos = last_pos;
        uint32_t save;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(a.rowmap + a.row_begin + h_rowof(pos, gstride)),
                       "s"(__builtin_amdgcn_readfirstlane((int)(rowid_base + (uint32_t)(t % 3) * 2048u + (uint32_t)wave * 256u))));
    };

    // request sources.  Corpus: request i < 4 of this wave fills local rows 32 wave + 8 i .. + 7 (lane l: row l / 8, chunk
    // position l % 8) -- from the image: request i < 2 fills rows 32 wave + 16 i .. + 15 (lane l: row l / 4, chunk position
    // l % 4, the query tile's layout); recomputed when the request cursor enters a tile.  Queries: fixed for the workgroup.
    const unsigned char *srcA[4], *srcB[2];
    const unsigned char *Xhb = reinterpret_cast<const unsigned char *>(a.Xh) + a.row_begin * (int64_t)H_XROW;
    const int64_t plane_bytes = a.xh_cap * (int64_t)H_XROW;
    auto set_srcA = [&](int t) { // request sources of tile t
        const uint32_t rt = rt_of(t);
        if (MAPPED) asm volatile("" ::: "memory"); // (the ids below were written by LDS-DMA: no read of them may be hoisted)
#pragma unroll
        for (int i = 0; i < NPA; i++) {
            const int row = AIMG ? wave * 32 + i * 16 + (lane >> 2) : wave * 32 + i * 8 + (lane >> 3);
            const int c = AIMG ? (lane & 3) ^ ((row >> 2) & 3) : (lane & 7) ^ ((row >> 1) & 7);
            if (MAPPED) {
                const int64_t rid = (int64_t)s_rowid[(t % 3) * 512 + row];
                srcA[i] = AIMG ? reinterpret_cast<const unsigned char *>(a.Xh) + rid * H_XROW + 16 * c
                               : reinterpret_cast<const unsigned char *>(a.X) + rid * row_bytes + 16 * c;
            } else {
                uint32_t pos = rt + (uint32_t)row;
                if (pos > last_pos) pos = last_pos;
                pos = h_rowof(pos, gstride);
                srcA[i] = AIMG ? Xhb + (int64_t)pos * H_XROW + 16 * c : Xb + (int64_t)pos * row_bytes + 16 * c;
            }
        }
    };
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int row = wave * 32 + j * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        int qr = q0 + row;
        if (qr > last_q) qr = last_q;
        srcB[j] = reinterpret_cast<const unsigned char *>(a.Qh + (int64_t)qr * H_BK) + 16 * c;
    }
    const int64_t kb_stride = (int64_t)a.q_stride * (H_BK * 2);
    const int nk = AIMG ? (a.D + H_BK - 1) / H_BK : a.D / H_BK; // (both images are zero-padded to a multiple of 32 dimensions)

    const int krot = (qt * a.rot) % nk;
    // request cursor: the stage to be requested next = K-step ik of this workgroup's tile it, into ring slot islot
    int it = 0, ik = 0, islot = 0;
    bool cursor_new_tile = false;
    auto piece = [&](int p) { // request p of the cursor's stage: 0 .. NPA - 1 corpus, then two of queries
        const uint32_t A = ring_base + (uint32_t)islot * STAGE + (uint32_t)(wave * 32 * (AIMG ? 64 : 128));
        const uint32_t B = ring_base + (uint32_t)islot * STAGE + A_BYTES + (uint32_t)(wave * 32 * 64);
        int ke = ik + krot; // (the sum over k does not care where it starts)
        if (ke >= nk) ke -= nk;
        if (p < NPA) h_dma16<NT>(srcA[p] + (AIMG ? h_xoff(ke, plane_bytes) : (int64_t)ke * (H_BK * 4)), A + 1024u * p);
        else h_dma16<false>(srcB[p - NPA] + ke * kb_stride, B + 1024u * (p - NPA));
    };
    auto advance = [&]() { // (beyond the last stage the cursor stays on it)
        islot = islot == NST - 1 ? 0 : islot + 1;
        if (ik + 1 < nk) ik++;
        else if (it + 1 < n_my) { it++; ik = 0; cursor_new_tile = true; }
    };
    auto aux_request = [&](int i) { // side input of tile i -> buffer i & 1 (every wave asks for 64 rows: entries 256 .. 511 repeat 0 .. 255)
        uint32_t pos = rt_of(i) + (uint32_t)((wave & 3) * 64 + lane);
        if (pos > last_pos) pos = last_pos;
        uint32_t src_row = h_rowof(pos, gstride);
        if (MAPPED) {
            asm volatile("" ::: "memory");
            src_row = s_rowid[(i % 3) * 512 + (wave & 3) * 64 + lane];
        }
        uint32_t save;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(auxg + src_row), "s"(__builtin_amdgcn_readfirstlane((int)(aux_base + (uint32_t)(i & 1) * 2048u + (uint32_t)wave * 256u))));
    };
    // the cursor has entered tile `it`: its request sources, and (MAPPED) the ids of the tile after it
    auto enter_tile = [&]() {
        set_srcA(it);
        if (MAPPED && it + 1 < n_my) rowid_request(it + 1);
        cursor_new_tile = false;
    };

    if (MAPPED) { // the ids of the first two tiles, once, before anything depends on them
        rowid_request(0);
        if (n_my > 1) rowid_request(1);
        h_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    set_srcA(0);
    aux_request(0);
#pragma unroll
    for (int st = 0; st < DIST; st++) { // the first DIST stages
        if (cursor_new_tile) enter_tile();
#pragma unroll
        for (int p = 0; p < NPS; p++) piece(p);
        advance();
    }

    // thresholds and scales of this workgroup's queries (kept for all its tiles).  tkc: the threshold in the form the epilogue
    // compares -- tau_key itself for L2, tau_key / qs for cosine and dot (qs = 2^-sh exactly, so 1 / qs is the float whose
    // exponent field is 254 minus that of qs, and the product is exact unless it leaves the normal range: an overflow admits
    // everything or nothing exactly as the unscaled comparison would, an underflow only moves ties -- see the epilogue)
    float tkc[4], qs[4], m2qs[4];
#pragma unroll
    for (int tn = 0; tn < 4; tn++) {
        const int qj = q0 + wc * 128 + tn * 32 + l31;
        const int qc = qj < a.nq ? qj : a.nq - 1;
        uint64_t tau = BOOT ? 0ull : a.cs.tau[qc];
        if (qj >= a.nq) tau = 0ull;
        const float tk = tau_key_of(tau);
        qs[tn] = a.qinv[qc];
        m2qs[tn] = -2.0f * qs[tn];
        if (METRIC == METRIC_DOT) { // key' = -(q.x)~ / G - |x|, G = (gamma_a + gamma_o) |q|: a LOWER bound of -q.x in units of G
            const float G = a.gsum * a.qnrm[qc];
            m2qs[tn] = G > 0.f ? -qs[tn] / G : 0.f;
        }
        const float scale = __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(uint32_t, qs[tn]));
        tkc[tn] = METRIC != METRIC_COS ? tk : tk * scale;
    }
    // admission segments: what is left of the LDS beside the ring (3.2 KB per wave with the fp16 copy, 1.2 KB without)
    // (MAPPED: the row-id ring takes 6 KB of it -- only built over the image)
    constexpr uint32_t WCAP = AIMG ? (MAPPED ? 208 : 272) : 104, WFLUSH = AIMG ? (MAPPED ? 136 : 176) : 60, SEG_BYTES = WCAP * 12;
    uint32_t *s_flag = reinterpret_cast<uint32_t *>(s_auxp + 2 * 512 + (MAPPED ? 3 * 512 : 0)); // [3] "somebody wants a flush", by tile % 3
    uint32_t *s_qn = s_flag + 4;                                                    // [H_BN] entries per query in the flush ...
    uint32_t *s_qb = s_qn + H_BN;                                                   // [H_BN] ... and where they start in its list
    unsigned char *seg = reinterpret_cast<unsigned char *>(s_qb + H_BN) + wave * SEG_BYTES;
    float *s_key = reinterpret_cast<float *>(seg);                // [WCAP]
    uint32_t *s_rid = reinterpret_cast<uint32_t *>(s_key + WCAP); // [WCAP]
    uint16_t *s_q = reinterpret_cast<uint16_t *>(s_rid + WCAP);   // [WCAP] query of the entry (in the tile)
    uint16_t *s_rk = s_q + WCAP;                                  // [WCAP] its rank among the flush's entries of that query
    uint32_t wcnt = 0;                                            // entries in the segment (wave-uniform)
    if (tid < H_BN) s_qn[tid] = 0;
    if (tid < 3) s_flag[tid] = 0; // (the first K-step's barrier is between this and the first use)

    // Fragments of a K-step's two MFMA k-blocks: block 0's are read HALF A STEP EARLY -- the wait + barrier that make stage
    // kt + 1 visible sit in the middle of step kt, and its first fragments are read while step kt's second k-block runs on
    // the matrix pipe.  (With the barrier at the top of the step all eight waves read LDS at the same moment and the pipe
    // idled for the latency, every step.)
    auto load_frag = [&](int slot_, int kb, f16x8(&af)[2], f16x8(&bf)[4]) {
        const unsigned char *As = ring + slot_ * STAGE;
        const unsigned char *Bs = As + A_BYTES;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
            const int r = wr * 64 + tm * 32 + l31;
            if (AIMG) {
                af[tm] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(As + hbswz(r, 2 * kb + h)));
            } else {
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(As + haswz(r, 4 * kb + 2 * h));
                const f32x4 x1 = *reinterpret_cast<const f32x4 *>(As + haswz(r, 4 * kb + 2 * h + 1));
                af[tm] = h_cvt8(x0, x1);
            }
        }
#pragma unroll
        for (int tn = 0; tn < 4; tn++) {
            const int r = wc * 128 + tn * 32 + l31;
            bf[tn] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(Bs + hbswz(r, 2 * kb + h)));
        }
    };
    auto spread_piece = [&](int sl) { // the requests of the stage DIST steps ahead, between the MFMA pairs (slot sl of 8)
        if (AIMG) {
            if ((sl & 1) == 0) piece(sl >> 1);
        } else {
            if (sl < 3) piece(sl);
            else if (sl == 4) piece(3);
            else if (sl == 5) piece(4);
            else if (sl == 6) piece(5);
        }
    };
    constexpr int H1 = AIMG ? 2 : 3; // requests of a stage that go out in the first half of a step
    constexpr bool PIPE = AIMG;      // (the f32 form has no registers for a third set of fragments: barrier at the top of the step)

    int cslot = 0; // ring slot of the stage being computed
    f16x8 a0[2], b0[4];
    if (PIPE) {
        h_wait_vmcnt<NPS *(DIST - 1)>(); // stage 0 has landed: at most the requests of the DIST - 1 younger stages are out
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_frag(0, 0, a0, b0);
    }
    for (int i = 0; i < n_my; i++) {
        const uint32_t rt = rt_of(i);
        f32x16 acc[2][4];
#pragma unroll
        for (int x = 0; x < 2; x++)
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[x][y][r] = 0.f;

        for (int kt = 0; kt < nk; kt++) {
            if (!PIPE) {
                h_wait_vmcnt<NPS *(DIST - 1)>(); // this stage has landed: at most the requests of the DIST - 1 younger ones are out
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                load_frag(cslot, 0, a0, b0);
            }
            f16x8 a1[2], b1[4];
            if (PIPE) load_frag(cslot, 1, a1, b1); // (visible since the barrier in the middle of the step before)
            if (cursor_new_tile) enter_tile();
            // side input of the next tile: its buffer was last read in the epilogue a tile ago, and a barrier of THIS tile lies
            // between that and the request (one K-step per tile: only the barrier in the middle of the step does)
            const bool aux_now = i + 1 < n_my && kt == (nk > 1 ? 1 : 0);
            if (aux_now && !(PIPE && nk == 1)) aux_request(i + 1);
#pragma unroll
            for (int tn = 0; tn < 4; tn++) { // k-block 0
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[tm], b0[tn], acc[tm][tn], 0, 0, 0);
                spread_piece(tn);
            }
            // middle of the step: this wave's reads of stage kt are complete (so the slot may be refilled once everybody is
            // here), stage kt + 1 has landed for this wave -- and, behind the barrier, for all
            const int nslot = cslot == NST - 1 ? 0 : cslot + 1;
            if (!PIPE) load_frag(cslot, 1, a1, b1);
            if (PIPE) {
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0), vmcnt / expcnt untouched
                h_wait_vmcnt<NPS *(DIST - 2) + H1>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (aux_now && nk == 1) aux_request(i + 1);
                load_frag(nslot, 0, a0, b0);
            }
#pragma unroll
            for (int tn = 0; tn < 4; tn++) { // k-block 1
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[tm], b1[tn], acc[tm][tn], 0, 0, 0);
                spread_piece(4 + tn);
            }
            advance();
            cslot = nslot;
        }

#ifdef LB_DIAG
        if (a.abl == 6) continue; // timing only: no epilogue
#endif
        // One K-step per tile (D <= 32): this tile's side input was asked for in the middle of the step before, which the
        // K-steps' waits do not cover (they cover a stage asked for DIST steps earlier, and with it everything older).  By now
        // 2 NPS - H1 (+ 1) requests are newer than it; vmcnt retires in order, the barrier publishes the other waves' parts.
        if (PIPE && nk == 1) {
            h_wait_vmcnt<2 * NPS - H1>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // ---- epilogue of the tile (the next tile's first two stages are landing meanwhile) --------------------------------
        // Two VALU operations per element: the candidate key in the form that needs ONE operation (cosine / dot: acc * (-ax),
        // compared with tau_key / qs -- qs is a power of two, so this is the comparison of the keys themselves; L2:
        // fma(acc, -2 qs, ax)) and v_cmp_le against the threshold.  It admits key <= tau_key: the exact rule (key < tau_key, or
        // equal keys and a lower row) plus the ties with a higher row -- a superset, which the select behind this launch orders
        // as it orders every other entry (tau only bounds the list).  Rows beyond the launch's range carry a NaN side input and
        // fail every comparison.  An admitted element costs three LDS stores; packing and the global atomic happen in the flush.
        // Every wave keeps its admissions in its own 100-entry LDS segment (beside the ring: nothing here waits for the ring
        // slot, so the epilogue has no barrier), its count in a scalar register (slots from the lane's rank among the
        // admitted lanes: no atomic, no wait), and flushes the segment itself once it is more than half full or the
        // workgroup is done -- every second or third tile: the flush's returning atomics cost a memory round trip, and under a
        // saturated corpus stream (one query tile) that was 15 % of the kernel when paid per tile.
        const float *s_aux = s_auxp + (i & 1) * 512;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
            float aux[4][4]; // cosine: -1/|x|, dot: -1, L2: |x|^2; NaN for a row beyond the range
            const uint32_t pos0 = rt + (uint32_t)(wr * 64 + tm * 32 + 4 * h); // position of element (g, e): pos0 + 8 g + e
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int lr = wr * 64 + tm * 32 + 8 * g + 4 * h;
                f32x4 av = {1.f, 1.f, 1.f, 1.f};
                av = *reinterpret_cast<const f32x4 *>(&s_aux[lr]);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v = METRIC == METRIC_L2 ? av[e] : (METRIC == METRIC_COS ? -av[e] : -sqrtf(av[e]) * dot_pad);
                    aux[g][e] = pos0 + (uint32_t)(8 * g + e) <= last_pos ? v : __builtin_nanf("");
                }
            }
            const uint32_t rid0 = (uint32_t)a.row_begin + pos0;
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                const int qj = q0 + wc * 128 + tn * 32 + l31;
                const bool qok = qj < a.nq;
                uint64_t *list = a.cs.lists + (size_t)(qok ? qj : 0) * a.cs.cap;
                if (BOOT) { // sample pass: the entry of position p goes to list[p] (rows beyond the range: none)
                    if (qok) {
#pragma unroll
                        for (int g = 0; g < 4; g++) { // four consecutive positions a lane: 32 B of the query's list in two stores
                            uint64_t ent[4];
                            const uint32_t pg = pos0 + (uint32_t)(8 * g);
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int x = 4 * g + e;
                                const float kp = METRIC != METRIC_COS ? fmaf(acc[tm][tn][x], m2qs[tn], aux[g][e]) : acc[tm][tn][x] * aux[g][e];
                                ent[e] = pack_entry(METRIC != METRIC_COS ? kp : kp * qs[tn], (uint32_t)a.row_begin + h_rowof(pg + (uint32_t)e, gstride));
                            }
                            h_store_entries4(list, pg, last_pos, ent, boot_vec);
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int x = 0; x < 16; x++) {
                    const float kp = METRIC != METRIC_COS ? fmaf(acc[tm][tn][x], m2qs[tn], aux[x >> 2][x & 3])
                                                         : acc[tm][tn][x] * aux[x >> 2][x & 3];
                    const bool adm = kp <= tkc[tn]; // (a padded query's threshold is NaN)
                    const uint64_t am = __builtin_amdgcn_ballot_w64(adm);
                    if (am != 0) { // (wave-uniform; one element in six has an admitted lane)
                        if (adm) {
                            const uint32_t sl = wcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
                            const float key = METRIC != METRIC_COS ? kp : kp * qs[tn];
                            const uint32_t rid = rid0 + 8 * (x >> 2) + (x & 3);
                            if (sl < WCAP) {
                                s_key[sl] = key;
                                s_rid[sl] = rid;
                                s_q[sl] = (uint16_t)(qj - q0);
                            } else { // (segment full) straight to the query's list
                                const uint32_t pos = atomicAdd(&a.cs.cnt[qj], 1u);
                                if (pos < a.cs.cap) list[pos] = pack_entry(key, rid);
                            }
                        }
                        wcnt += (uint32_t)__builtin_popcountll(am);
                    }
                }
            }
        }
        // Flush.  The returned positions of the flush's atomics retire behind every request this wave has in flight (vmcnt is
        // in order): a flush drains the wave's part of the ring, and through the barriers everybody waits for it.  So the
        // waves flush TOGETHER, and rarely: a wave that is past WFLUSH says so in the tile's flag; a tile later (the K-steps'
        // barriers have made the flag visible to all) every wave flushes.  Three flags in rotation: this tile's is written,
            // the one before is read, the one after (read a tile ago by everybody) is cleared.  A flush then costs ONE global atomic
        // per query with entries (ranks from LDS atomics): the per-entry atomics of an earlier version were ~2,300 on each query's
        // counter per search, and their serialisation showed (0.1 ms at 128 queries).
        if (wcnt > WFLUSH && lane == 0) s_flag[i % 3] = 1;
        if (tid == 0) s_flag[(i + 1) % 3] = 0;
        bool flush = !BOOT && (i + 1 == n_my || (i > 0 && s_flag[(i + 2) % 3] != 0));
#ifdef LB_DIAG
        if (a.abl == 7) flush = false;
#endif
        if (flush) { // (all waves are here together) ONE returning global atomic per query with entries, then the stores
            const uint32_t total = wcnt < WCAP ? wcnt : WCAP;
            for (uint32_t z = lane; z < total; z += 64) s_rk[z] = (uint16_t)atomicAdd(&s_qn[s_q[z]], 1u);
            __syncthreads();
            if (tid < H_BN) {
                const uint32_t n = s_qn[tid];
                s_qb[tid] = n ? atomicAdd(&a.cs.cnt[q0 + tid], n) : 0u; // (n != 0 implies a real query)
                s_qn[tid] = 0;
            }
            __syncthreads();
            for (uint32_t z = lane; z < total; z += 64) {
                const int ql = (int)s_q[z];
                const uint32_t pos = s_qb[ql] + (uint32_t)s_rk[z];
                if (pos < a.cs.cap) a.cs.lists[(size_t)(q0 + ql) * a.cs.cap + pos] = pack_entry(s_key[z], s_rid[z]);
            }
            wcnt = 0;
        }
    }
    h_wait_vmcnt<0>(); // the re-read stages behind the last tile: nothing may land in LDS after the workgroup has gone
}

// ---- up to 256 queries (one query tile) over the corpus's fp16 image: the pass is the image's HBM stream ---------------------
// Same persistent pipeline, tile 256 rows x BN queries (64, 128 or 256): a wave owns 32 rows x BN queries (BN / 32 MFMA
// tiles), stages exactly the corpus rows it consumes (2 requests per stage) and an eighth of the query tile (half a request
// at BN = 64 -- lanes 0 .. 31 --, one at 128, two at 256); a stage is 16 KB + 4 / 8 / 16 KB, the ring SIX / FIVE / FOUR stages
// deep (48-80 KB of corpus in flight per CU), loads non-temporal (every line is read once).  1M x 768: the image is 1.5 GB,
// the pass 0.24-0.3 ms where the f32 rows take 0.49.
// MAPPED: the positions index a.rowmap (the visible rows of a filtered view, ascending) -- the kernel gathers: the row ids of a
// tile come in by their own LDS-DMA request one tile ahead of the request cursor (a ring of three tiles: a plain load inside
// the loop would make the compiler wait for vmcnt(0) and drain the stage ring), the corpus pieces and the side inputs are then
// requested at rowmap[position]; candidate entries carry POSITIONS (the finish launch maps them to rows: posmap).  Needs at
// least DIST + 3 K-steps per tile (the launcher checks), so that a tile's ids are older than everything the counted waits
// leave in flight by the time they are read.
// positions a duty is worth: ~5 us of selection (+ D dependent additions of ~4.5 ns for the exact norm) against 0.077 ns per
// dimension and row of the stream (6.7 TB/s over 256 workgroups, 2 bytes per element)
static uint32_t tin_duty_rows(int D, bool with_norm)
{
    static const int pct = lb_tunable("LB_TIN_DR_PCT", 100);
    const double ns = (5000.0 + (with_norm ? 4.5 * D : 0.0)) * 0.01 * pct;
    return ((uint32_t)(ns / (0.077 * D)) + 15u) & ~15u;
}

// can a one-tile launch over n_pos positions on ng workgroups compute its nq thresholds itself?  (every duty workgroup keeps
// at least a tile of rows; the sample fits the 512 x 16 keys of tin_duty)
static bool tin_plan(int D, int nq, int64_t n_pos, int ng, bool with_norm, uint32_t count, int m, uint32_t &dr)
{
    if (nq < 1 || nq > ng / 2 || nq > 128 || n_pos >= ((int64_t)1 << 30) || count == 0 || count > (uint32_t)H_THREADS * 16u || m < 1 || m > 64) return false;
    dr = tin_duty_rows(D, with_norm);
    return (n_pos + (int64_t)nq * dr) / ng >= (int64_t)dr + 256;
}

// TAUIN: the FIRST nd workgroups of the launch have a duty in front of their rows (a threshold each: ~5 us + an exact norm);
// their ranges are dr positions shorter than the others', so that everybody ends together.  The ranges are h_range's over
// V = n_pos + nd dr "virtual" positions, dr of which at the head of every duty workgroup's range are its duty.  (The first,
// not the last: workgroups are dispatched in index order, so whenever any workgroup of the launch runs, the duty workgroups
// have been dispatched before it and depend on nobody -- a launch that shares the GPU with another persistent launch and is
// only partly resident still gets its thresholds.)
__device__ __forceinline__ void h_range_duty(uint32_t n_pos, int gi, int ng, int nd, uint32_t dr, uint32_t &lo, uint32_t &hi)
{
    const uint32_t V = n_pos + (uint32_t)nd * dr, q = V / (uint32_t)ng, r = V % (uint32_t)ng;
    auto bound = [&](int i) { // first real position of workgroup i
        const uint32_t v = q * (uint32_t)i + (r * (uint32_t)i) / (uint32_t)ng;
        const int jd = i < nd ? i : nd; // duty workgroups in front of it
        return (v - (uint32_t)jd * dr) & ~15u;
    };
    lo = bound(gi);
    hi = gi + 1 < ng ? bound(gi + 1) : n_pos;
    lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
}

// The duty of workgroup j < nq of a TAUIN launch: tau[j] = the m-th smallest key of the sample entries
// lists[j][0 .. count) with the row bits saturated (kernels_scan.hip: sample_tau_body, here on 512 threads with 16 keys
// each), cnt[j] = 0, the exact ||q_j||^2 (cosine).  `scratch`: LDS nobody uses
// before the first epilogue (>= 1 KB + a query row).
__device__ __forceinline__ void tin_duty(const Tall16Args &a, int j, unsigned char *scratch)
{
    constexpr uint32_t NONE = 0xffffffffu;
    constexpr int PER = 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t(*wl)[20] = reinterpret_cast<uint32_t(*)[20]>(scratch); // [8][20]
    float *sq = reinterpret_cast<float *>(scratch + 1024);
    const uint64_t *list = a.cs.lists + (size_t)j * a.cs.cap;
    uint32_t e[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint32_t idx = (uint32_t)tid + (uint32_t)H_THREADS * i;
        e[i] = idx < a.tin_count ? (uint32_t)(list[idx] >> 32) : NONE;
    }
    if (a.tin_qna) { // (the query row, for the norm below: in flight beside the entries)
        const float *q = a.tin_Q + (int64_t)j * a.D;
        for (int i = tid; i < a.D; i += H_THREADS) sq[i] = q[i];
    }
    // two sorted runs of 8 per thread (19 compare-exchanges each); a pop takes the smaller head
#define LB_CE(i, j)                         \
    {                                       \
        const uint32_t x = e[i], y = e[j];  \
        e[i] = x < y ? x : y;               \
        e[j] = x < y ? y : x;               \
    }
#define LB_SORT8(o)                                                                                                              \
    LB_CE(o + 0, o + 1) LB_CE(o + 2, o + 3) LB_CE(o + 4, o + 5) LB_CE(o + 6, o + 7) LB_CE(o + 0, o + 2) LB_CE(o + 1, o + 3)         \
    LB_CE(o + 4, o + 6) LB_CE(o + 5, o + 7) LB_CE(o + 1, o + 2) LB_CE(o + 5, o + 6) LB_CE(o + 0, o + 4) LB_CE(o + 3, o + 7)         \
    LB_CE(o + 1, o + 5) LB_CE(o + 2, o + 6) LB_CE(o + 1, o + 4) LB_CE(o + 3, o + 6) LB_CE(o + 2, o + 4) LB_CE(o + 3, o + 5)         \
    LB_CE(o + 3, o + 4)
    LB_SORT8(0)
    LB_SORT8(8)
#undef LB_SORT8
#undef LB_CE
    const int m = a.tin_m;
    const int r1 = m < 4 + m / 4 ? m : 4 + m / 4; // (as sample_tau_body: a wave that held more of the m smallest gives a LOOSER threshold)
    for (int r = 0; r < r1; r++) {
        const uint32_t hd = e[0] < e[8] ? e[0] : e[8];
        const uint32_t v = wave_min_u32(hd);
        if (lane == 0) wl[wave][r] = v;
        const uint64_t holders = __builtin_amdgcn_ballot_w64(hd == v);
        if (v != NONE && lane == (int)__builtin_ctzll(holders)) { // exactly one lane pops a head
            if (e[0] <= e[8]) {
#pragma unroll
                for (int i = 0; i < 7; i++) e[i] = e[i + 1];
                e[7] = NONE;
            } else {
#pragma unroll
                for (int i = 8; i < 15; i++) e[i] = e[i + 1];
                e[15] = NONE;
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t kth = NONE;
        int ptr = 0; // lanes 0 .. 7: the head of wave `lane`'s run
        for (int r = 0; r < m; r++) {
            const uint32_t head = (lane < H_THREADS / 64 && ptr < r1) ? wl[lane][ptr] : NONE;
            kth = (uint32_t)__builtin_amdgcn_readfirstlane((int)row16_min_u32(head));
            if (kth == NONE) break;
            const uint64_t holders = __builtin_amdgcn_ballot_w64(head == kth);
            if (lane == (int)__builtin_ctzll(holders)) ptr++;
        }
        if (lane == 0) {
            const uint64_t tv = kth == NONE ? kEntryMax : (((uint64_t)kth << 32) | 0xffffffffull);
            // the count first, then the threshold with release semantics: whoever sees the threshold appends behind a zeroed count
            __hip_atomic_store(&a.cs.cnt[j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (the asm wait stays beside the fence: ROCm 7.2 can drop the fence's own vmcnt wait, MI355X_MICROARCH.md)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LB_DIAG
            if (a.abl != 12) // (12: the threshold never comes out -- every wave's wait gives up; tests/test_gpu_thresholds_in_launch.py)
#endif
            __hip_atomic_store(&a.cs.tau[j], tv, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (a.tin_qna) { // D dependent additions in the reference's order (nothing waits for them but this workgroup's rows)
#pragma clang fp contract(off)
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                const int D = a.D, dmain = D & ~3;
                if (a.tin_order == 1) { // ORDER_UNROLL4
                    for (int i = 0; i < dmain; i += 4) {
                        const float v0 = sq[i], v1 = sq[i + 1], v2 = sq[i + 2], v3 = sq[i + 3];
                        s0 = s0 + v0 * v0;
                        s1 = s1 + v1 * v1;
                        s2 = s2 + v2 * v2;
                        s3 = s3 + v3 * v3;
                    }
                    for (int i = dmain; i < D; i++) s0 = s0 + sq[i] * sq[i];
                    float t = s0 + s1;
                    t = t + s2;
                    t = t + s3;
                    a.tin_qna[j] = t;
                } else {
                    for (int i = 0; i < D; i++) s0 = s0 + sq[i] * sq[i];
                    a.tin_qna[j] = s0;
                }
            }
        }
    }
    __syncthreads();
}

template <int METRIC, int BN, bool BOOT, bool MAPPED, bool TAUIN = false>
__global__ __launch_bounds__(H_THREADS, 2) void gemm_filter_narrow16p_kernel(Tall16Args a, int spx)
{
    constexpr int TN = BN / 32;
    constexpr int NST = BN == 64 ? 6 : (BN == 128 ? 5 : 4), A_BYTES = H_BM * H_BK * 2, B_BYTES = BN * H_BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int NPB = BN == 256 ? 2 : 1; // query requests per wave and stage
    constexpr int NPS = 2 + NPB, DIST = NST - 1, H1 = 2;
    constexpr bool PIPE = BN != 256; // (at 256 queries there are no registers for a third set of fragments: barrier on top)
    (void)spx;
    extern __shared__ __attribute__((aligned(16))) unsigned char hlds[];
    int bi = (int)blockIdx.x, ng = (int)gridDim.x;
    if (BOOT && a.qsplit) {
        // The sample of a batch of more than 128 queries: the launch's workgroups form a.qsplit groups, each scoring ALL sampled
        // positions for its window of 128 queries (its part of the query image, the scales and the candidate state) -- 64
        // positions a workgroup on every CU at 256 queries, where the 256-query kernel walks the 8192 rows as 32 whole tiles
        // on 32 CUs (38 us at 768 dimensions, 60 at 1536 under a row list).
        ng /= a.qsplit;
        const int win = bi / ng;
        if (win >= a.qsplit) return; // (the grid is not a multiple of the window count)
        bi -= win * ng;
        const int qoff = win * 128;
        a.Qh += (int64_t)qoff * H_BK;
        a.qinv += qoff;
        if (a.qnrm) a.qnrm += qoff;
        a.cs.lists += (size_t)qoff * a.cs.cap;
        a.cs.cnt += qoff;
        a.cs.tau += qoff;
        a.nq = a.nq - qoff < 128 ? a.nq - qoff : 128;
    }
    uint32_t lo, hi; // this workgroup's positions
    if (TAUIN) {
        h_range_duty((uint32_t)(a.row_end - a.row_begin), bi, ng, a.nq, a.tin_dr, lo, hi);
        if (bi < a.nq) tin_duty(a, bi, hlds);
    } else {
        h_range((uint32_t)(a.row_end - a.row_begin), bi, ng, lo, hi);
    }
    if (hi <= lo) return;
    const int n_my = (int)((hi - lo + H_BM - 1) / H_BM); // its tiles: positions lo + 256 i ..

    unsigned char *ring = hlds;
    float *s_auxp = reinterpret_cast<float *>(ring + NST * STAGE); // [2][512]
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ring;
    const uint32_t aux_base = ring_base + (uint32_t)(NST * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int last_q = a.nq - 1;
    const uint32_t last_pos = hi - 1;
    const uint32_t gstride = BOOT ? a.gstride : 0u;
    const bool boot_vec = BOOT && (a.cs.cap & 3u) == 0u && (reinterpret_cast<uintptr_t>(a.cs.lists) & 15) == 0; // (h_store_entries4)
    // dot product: |x| of the lower-bound key, padded for the f32 roundings of the key's own fma (they are relative to the first
    // term, up to 1 / (gamma_a + gamma_o) times |x|) and of the stored norm
    const float dot_pad = 1.0f + (a.gsum > 0.f ? 6.0e-7f / a.gsum : 0.f) + 1.0e-5f + 1.05f * (float)(a.D + 8) * 5.9604645e-8f;
    const float *auxg = METRIC == METRIC_COS ? a.rnorm + (MAPPED ? 0 : a.row_begin) : a.norm2 + (MAPPED ? 0 : a.row_begin); // (dot: |x| for the lower-bound key)
    auto rt_of = [&](int i) { return lo + (uint32_t)i * H_BM; }; // first position of tile i
    // MAPPED: row ids of three tiles, [3][512] (every wave asks for 64: entries 256 .. 511 repeat 0 .. 255, as the side inputs do)
    const uint32_t *s_rowid = reinterpret_cast<const uint32_t *>(s_auxp + 2 * 512);
    const uint32_t rowid_base = aux_base + 4096u;
    auto rowid_request = [&](int t) {
        uint32_t pos = rt_of(t) + (uint32_t)((wave & 3) * 64 + lane);
        if (pos > last_pos) pos = last_pos;
        uint32_t save;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(a.rowmap + a.row_begin + h_rowof(pos, gstride)),
                       "s"(__builtin_amdgcn_readfirstlane((int)(rowid_base + (uint32_t)(t % 3) * 2048u + (uint32_t)wave * 256u))));
    };

    const unsigned char *srcA[2], *srcB[NPB];
    const unsigned char *Xhb = reinterpret_cast<const unsigned char *>(a.Xh) + a.row_begin * (int64_t)H_XROW;
    const int64_t plane_bytes = a.xh_cap * (int64_t)H_XROW;
    auto set_srcA = [&](int t) { // request sources of tile t
        const uint32_t rt = rt_of(t);
        if (MAPPED) asm volatile("" ::: "memory"); // (the ids below were written by LDS-DMA: no read of them may be hoisted)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = wave * 32 + i * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            if (MAPPED) {
                srcA[i] = reinterpret_cast<const unsigned char *>(a.Xh) + (int64_t)s_rowid[(t % 3) * 512 + row] * H_XROW + 16 * c;
            } else {
                uint32_t pos = rt + (uint32_t)row;
                if (pos > last_pos) pos = last_pos;
                srcA[i] = Xhb + (int64_t)h_rowof(pos, gstride) * H_XROW + 16 * c;
            }
        }
    };
#pragma unroll
    for (int j = 0; j < NPB; j++) {
        // BN = 64: 8 query rows per wave, lanes 32 .. 63 do not take part in the request; 128: 16 rows; 256: 2 x 16 rows
        const int row = BN == 64 ? wave * 8 + ((lane & 31) >> 2) : wave * (BN / 8) + j * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        const int qr = row > last_q ? last_q : row;
        srcB[j] = reinterpret_cast<const unsigned char *>(a.Qh + (int64_t)qr * H_BK) + 16 * c;
    }
    const int64_t kb_stride = (int64_t)a.q_stride * (H_BK * 2);
    const int nk = (a.D + H_BK - 1) / H_BK; // (both images are zero-padded to a multiple of 32 dimensions)

    int it = 0, ik = 0, islot = 0;
    bool cursor_new_tile = false;
    auto piece = [&](int p) {
        const uint32_t A = ring_base + (uint32_t)islot * STAGE + (uint32_t)(wave * 32 * 64);
        const uint32_t B = ring_base + (uint32_t)islot * STAGE + A_BYTES + (uint32_t)(wave * (BN * 8));
        if (p < 2) h_dma16<true>(srcA[p] + h_xoff(ik, plane_bytes), A + 1024u * p);
        else if (BN != 64 || lane < 32) h_dma16<false>(srcB[p - 2] + ik * kb_stride, B + 1024u * (p - 2));
    };
    auto advance = [&]() {
        islot = islot == NST - 1 ? 0 : islot + 1;
        if (ik + 1 < nk) ik++;
        else if (it + 1 < n_my) { it++; ik = 0; cursor_new_tile = true; }
    };
    auto aux_request = [&](int i) {
        uint32_t pos = rt_of(i) + (uint32_t)((wave & 3) * 64 + lane);
        if (pos > last_pos) pos = last_pos;
        uint32_t src_row = h_rowof(pos, gstride);
        if (MAPPED) {
            asm volatile("" ::: "memory");
            src_row = s_rowid[(i % 3) * 512 + (wave & 3) * 64 + lane];
        }
        uint32_t save;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(auxg + src_row), "s"(__builtin_amdgcn_readfirstlane((int)(aux_base + (uint32_t)(i & 1) * 2048u + (uint32_t)wave * 256u))));
    };
    // the cursor has entered tile `it`: its request sources, and (MAPPED) the ids of the tile after it
    auto enter_tile = [&]() {
        set_srcA(it);
        if (MAPPED && it + 1 < n_my) rowid_request(it + 1);
        cursor_new_tile = false;
    };

    if (MAPPED) { // the ids of the first two tiles, once, before anything depends on them
        rowid_request(0);
        if (n_my > 1) rowid_request(1);
        h_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    set_srcA(0);
    aux_request(0);
#pragma unroll
    for (int st = 0; st < DIST; st++) {
        if (cursor_new_tile) enter_tile();
#pragma unroll
        for (int p = 0; p < NPS; p++) piece(p);
        advance();
    }

    float tkc[TN], qs[TN], m2qs[TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int qj = tn * 32 + l31;
        const int qc = qj < a.nq ? qj : a.nq - 1;
        uint64_t tau = (BOOT || TAUIN) ? 0ull : a.cs.tau[qc];
        if (qj >= a.nq) tau = 0ull;
        const float tk = tau_key_of(tau);
        qs[tn] = a.qinv[qc];
        m2qs[tn] = -2.0f * qs[tn];
        if (METRIC == METRIC_DOT) { // key' = -(q.x)~ / G - |x|, G = (gamma_a + gamma_o) |q|: a LOWER bound of -q.x in units of G
            const float G = a.gsum * a.qnrm[qc];
            m2qs[tn] = G > 0.f ? -qs[tn] / G : 0.f;
        }
        // (TAUIN: nothing below consumes the scales before the first admission, so the compiler would leave their loads in
        // flight and put the "s_waitcnt vmcnt(0)" INTO the admission branch, where it drains the DMA ring at every admitted
        // element -- measured: 297 us instead of 238.  Consumed here they are waited for once, in front of the loop.)
        if (TAUIN) asm volatile("" : "+v"(qs[tn]), "+v"(m2qs[tn]));
        const float scale = __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(uint32_t, qs[tn]));
        tkc[tn] = METRIC != METRIC_COS ? tk : tk * scale;
#ifdef LB_DIAG
        if (a.abl == 8) tkc[tn] = -__builtin_huge_valf(); // timing only: nothing is admitted
#endif
    }
    // (what the ring leaves; MAPPED: 6 KB less, the row-id ring)
    constexpr uint32_t WCAP = BN == 256 ? 272 : (MAPPED ? 296 : 360), WFLUSH = BN == 256 ? 176 : (MAPPED ? 200 : 240), SEG_BYTES = WCAP * 12;
    uint32_t *s_flag = reinterpret_cast<uint32_t *>(s_auxp + 2 * 512 + (MAPPED ? 3 * 512 : 0)); // [3] "somebody wants a flush", by tile % 3 (see the 256-query form)
    uint32_t *s_qn = s_flag + 4, *s_qb = s_qn + 256;
    unsigned char *seg = reinterpret_cast<unsigned char *>(s_qb + 256) + wave * SEG_BYTES;
    float *s_key = reinterpret_cast<float *>(seg);
    uint32_t *s_rid = reinterpret_cast<uint32_t *>(s_key + WCAP);
    uint16_t *s_q = reinterpret_cast<uint16_t *>(s_rid + WCAP);
    uint16_t *s_rk = s_q + WCAP;
    uint32_t wcnt = 0;
    if (tid < 256) s_qn[tid] = 0;
    if (tid < 3) s_flag[tid] = 0;

    auto load_frag = [&](int slot_, int kb, f16x8 &af, f16x8(&bf)[TN]) {
        const unsigned char *As = ring + slot_ * STAGE;
        const unsigned char *Bs = As + A_BYTES;
        af = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(As + hbswz(wave * 32 + l31, 2 * kb + h)));
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
            bf[tn] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(Bs + hbswz(tn * 32 + l31, 2 * kb + h)));
    };

    int cslot = 0;
    f16x8 a0, b0[TN];
    if (PIPE) {
        h_wait_vmcnt<NPS *(DIST - 1)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_frag(0, 0, a0, b0);
    }
    for (int i = 0; i < n_my; i++) {
        const uint32_t rt = rt_of(i);
        f32x16 acc[TN];
#pragma unroll
        for (int y = 0; y < TN; y++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[y][r] = 0.f;

        for (int kt = 0; kt < nk; kt++) {
            if (!PIPE) {
                h_wait_vmcnt<NPS *(DIST - 1)>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                load_frag(cslot, 0, a0, b0);
            }
            f16x8 a1, b1[TN];
            if (PIPE) load_frag(cslot, 1, a1, b1);
            if (cursor_new_tile) enter_tile();
            const bool aux_now = i + 1 < n_my && kt == (nk > 1 ? 1 : 0); // (see the 256-query form)
            if (aux_now && !(PIPE && nk == 1)) aux_request(i + 1);
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0[tn], acc[tn], 0, 0, 0);
                if (tn < 2) piece(tn);
            }
            const int nslot = cslot == NST - 1 ? 0 : cslot + 1;
            if (!PIPE) load_frag(cslot, 1, a1, b1);
            if (PIPE) { // middle of the step (see the 256-query form): reads of stage kt complete, stage kt + 1 landed, barrier
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
                h_wait_vmcnt<NPS *(DIST - 2) + H1>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (aux_now && nk == 1) aux_request(i + 1);
                load_frag(nslot, 0, a0, b0);
            }
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1[tn], acc[tn], 0, 0, 0);
                if (tn < NPB) piece(2 + tn);
            }
            advance();
            cslot = nslot;
        }

        // One or two K-steps per tile (D <= 64): the side input of this tile was asked for fewer requests ago than the ring keeps
        // in flight (the K-steps' waits cover a stage asked for DIST steps earlier, and the side input only when it is older
        // than that stage: from three K-steps per tile on).  Requests newer than it by now: 2 NPS - H1 (+ 1) at one K-step per
        // tile (asked for in the middle of the step before), 3 NPS (+ 1) at two (at the top of the tile before's second step);
        // vmcnt retires in order, and the barrier publishes the other waves' parts.
        if (PIPE && nk <= 2) {
            if (nk == 1) h_wait_vmcnt<2 * NPS - H1>();
            else h_wait_vmcnt<3 * NPS>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (TAUIN && i == 0) { // the thresholds, published meanwhile by the launch's first nq workgroups
            // Every wave reads ITS queries' thresholds with device-coherent vector loads (each drains the wave's part of the DMA
            // ring once); a threshold of 0 is "not out yet" (the launch before left them so; a genuine one ends in 32 one-bits).
            // Usually the first look finds them all: the duty takes ~6 us, the first epilogue comes ~16 us into the launch.
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            uint64_t tauv[TN];
            bool ok = false;
#pragma nounroll
            for (uint32_t spin = 0; spin < 1500u && !ok; spin++) { // ~0.7 us a round: ~1 ms in all
                bool all = true;
#pragma unroll
                for (int tn = 0; tn < TN; tn++) {
                    const int qj = tn * 32 + l31;
                    u32x2 tv;
                    const uint64_t *tp = a.cs.tau + (qj < a.nq ? qj : 0);
                    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(tv) : "v"(tp) : "memory");
                    tauv[tn] = ((uint64_t)tv[1] << 32) | tv[0];
                    all = all && tauv[tn] != 0ull;
                }
                ok = __builtin_amdgcn_ballot_w64(!all) == 0ull;
                if (!ok) __builtin_amdgcn_s_sleep(8);
            }
            // (gave up: this WAVE admits nothing -- whichever wave it is must say so, the proof behind the pass relies on every
            // row below the threshold having been admitted -- and the host redoes the batch)
            if (!ok && lane == 0) *a.tin_fail = a.tin_tag;
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                const int qj = tn * 32 + l31;
                uint64_t tau = tauv[tn];
                if (!ok || qj >= a.nq) tau = 0ull;
                const float tk = tau_key_of(tau);
                const float scale = __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(uint32_t, qs[tn]));
                tkc[tn] = METRIC != METRIC_COS ? tk : tk * scale;
#ifdef LB_DIAG
                if (a.abl == 8) tkc[tn] = -__builtin_huge_valf();
#endif
            }
        }
        // ---- epilogue of the tile (as the 256-query form: 2 VALU + 1 scalar branch per element, per-wave segments) ----------
        const float *s_aux = s_auxp + (i & 1) * 512;
        float aux[4][4];
        const uint32_t pos0 = rt + (uint32_t)(wave * 32 + 4 * h);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int lr = wave * 32 + 8 * g + 4 * h;
            f32x4 av = {1.f, 1.f, 1.f, 1.f};
            av = *reinterpret_cast<const f32x4 *>(&s_aux[lr]);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float v = METRIC == METRIC_L2 ? av[e] : (METRIC == METRIC_COS ? -av[e] : -sqrtf(av[e]) * dot_pad);
                aux[g][e] = pos0 + (uint32_t)(8 * g + e) <= last_pos ? v : __builtin_nanf("");
            }
        }
        const uint32_t rid0 = (uint32_t)a.row_begin + pos0;
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
            const int qj = tn * 32 + l31;
            const bool qok = qj < a.nq;
            uint64_t *list = a.cs.lists + (size_t)(qok ? qj : 0) * a.cs.cap;
            if (BOOT) {
                if (qok) {
#pragma unroll
                    for (int g = 0; g < 4; g++) { // four consecutive positions a lane: 32 B of the query's list in two stores
                        uint64_t ent[4];
                        const uint32_t pg = pos0 + (uint32_t)(8 * g);
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int x = 4 * g + e;
                            const float kp = METRIC != METRIC_COS ? fmaf(acc[tn][x], m2qs[tn], aux[g][e]) : acc[tn][x] * aux[g][e];
                            ent[e] = pack_entry(METRIC != METRIC_COS ? kp : kp * qs[tn], (uint32_t)a.row_begin + h_rowof(pg + (uint32_t)e, gstride));
                        }
                        h_store_entries4(list, pg, last_pos, ent, boot_vec);
                    }
                }
                continue;
            }
#pragma unroll
            for (int x = 0; x < 16; x++) {
                const float kp = METRIC != METRIC_COS ? fmaf(acc[tn][x], m2qs[tn], aux[x >> 2][x & 3]) : acc[tn][x] * aux[x >> 2][x & 3];
                const bool adm = kp <= tkc[tn];
                const uint64_t am = __builtin_amdgcn_ballot_w64(adm);
                if (am != 0) {
                    if (adm) {
                        const uint32_t sl = wcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
                        const float key = METRIC != METRIC_COS ? kp : kp * qs[tn];
                        const uint32_t rid = rid0 + 8 * (x >> 2) + (x & 3);
                        if (sl < WCAP) {
                            s_key[sl] = key;
                            s_rid[sl] = rid;
                            s_q[sl] = (uint16_t)qj;
                        } else {
                            const uint32_t pos = atomicAdd(&a.cs.cnt[qj], 1u);
                            if (pos < a.cs.cap) list[pos] = pack_entry(key, rid);
                        }
                    }
                    wcnt += (uint32_t)__builtin_popcountll(am);
                }
            }
        }
        if (wcnt > WFLUSH && lane == 0) s_flag[i % 3] = 1;
        if (tid == 0) s_flag[(i + 1) % 3] = 0;
        if (!BOOT && (i + 1 == n_my || (i > 0 && s_flag[(i + 2) % 3] != 0))) { // (all waves together: see the 256-query form)
            const uint32_t total = wcnt < WCAP ? wcnt : WCAP;
            for (uint32_t z = lane; z < total; z += 64) s_rk[z] = (uint16_t)atomicAdd(&s_qn[s_q[z]], 1u);
            __syncthreads();
            if (tid < BN) {
                const uint32_t n = s_qn[tid];
                s_qb[tid] = n ? atomicAdd(&a.cs.cnt[tid], n) : 0u;
                s_qn[tid] = 0;
            }
            __syncthreads();
            for (uint32_t z = lane; z < total; z += 64) {
                const int qg = (int)s_q[z];
                const uint32_t pos = s_qb[qg] + (uint32_t)s_rk[z];
                if (pos < a.cs.cap) a.cs.lists[(size_t)qg * a.cs.cap + pos] = pack_entry(s_key[z], s_rid[z]);
            }
            wcnt = 0;
        }
    }
    h_wait_vmcnt<0>();
}

// f32 [nq][D] -> fp16 [D / 32][nq][32], each query scaled by the power of two that brings its norm into [1, 2); qinv[q] = 1 / scale.
// One wave per query.  (A zero or non-finite query keeps scale 1: its search is answered by the exact scan anyway.)
__global__ __launch_bounds__(256) void queries_to_f16_kernel(const float *Q, int nq, int D, _Float16 *Qh, float *qinv)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float *src = Q + (int64_t)q * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += src[i] * src[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    float scale = 1.f, inv = 1.f;
    if (s > 0.f && s < 3.0e38f) {
        const float nrm = sqrtf(s);
        int e;
        (void)frexpf(nrm, &e);       // nrm = m * 2^e, 0.5 <= m < 1  ->  nrm * 2^(1 - e) in [1, 2)
        int sh = 1 - e;
        sh = sh < -120 ? -120 : (sh > 120 ? 120 : sh);
        scale = ldexpf(1.f, sh);
        inv = ldexpf(1.f, -sh);
    }
    const int Dp = (D + 31) & ~31; // (dimensions beyond D: zero -- they add nothing to a product)
    for (int i = lane; i < Dp; i += 64) Qh[((int64_t)(i >> 5) * nq + q) * 32 + (i & 31)] = i < D ? (_Float16)(src[i] * scale) : (_Float16)0.f;
    if (lane == 0) qinv[q] = inv;
}

// f32 corpus rows [row_begin, row_end) -> the blocked fp16 image Xh[Dp / H_XP][cap][H_XP] (round to nearest even, the conversion
// the kernels above apply in registers: both forms of the route see the same fp16 values; dimensions beyond D are zero).
// One workgroup = 64 rows x 32 dimensions: whole 128-B lines in, 64-B pieces out.
__global__ __launch_bounds__(256) void corpus_to_f16_kernel(const float *X, int64_t row_begin, int64_t row_end, int D, _Float16 *Xh, int64_t cap,
                                                            const float *center)
{
    const int64_t row = row_begin + (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const int kb = blockIdx.y, q = threadIdx.x & 3; // kb: block of 32 dimensions
    if (row >= row_end) return;
    const int k0 = kb * 32 + q * 8;
    f16x8 v;
    if (center) { // the centred image (L2): fp16(x - center), the subtraction in f32
        const float *src = X + row * (int64_t)D;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = k0 + e < D ? (_Float16)(src[k0 + e] - center[k0 + e]) : (_Float16)0.f;
    } else if (k0 + 8 <= D && (D & 3) == 0) { // (rows are 16-B aligned when D % 4 == 0)
        const f32x4 *src = reinterpret_cast<const f32x4 *>(X + row * (int64_t)D + k0);
        v = h_cvt8(src[0], src[1]);
    } else { // the last block of a dimension that is not a multiple of 32 (zero beyond D), or unaligned rows
        const float *src = X + row * (int64_t)D;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = k0 + e < D ? (_Float16)src[k0 + e] : (_Float16)0.f;
    }
    *reinterpret_cast<f16x8 *>(Xh + ((int64_t)(k0 / H_XP) * cap + row) * H_XP + (k0 % H_XP)) = v;
}

// What the fp16 image loses, MEASURED: stat = max over rows of |x - fp16(x)|^2 / |x|^2 (float bits; x - center for the centred
// image), one wave a row.  |q.x - q~.x~| <= |q| |x - x~| + |q - q~| |x~| by Cauchy-Schwarz on the residual VECTORS: with the
// measured ratios (~0.3 x 2^-11 for data that fills the mantissa) the candidate keys' rigorous error bound is a third to a half
// of the per-element worst case 2^-11 + 2^-11 -- and with it the rows the finish launch has to score exactly.  (Elements that
// fall into fp16's subnormal range or flush to zero are in the residual like everything else.)
__global__ __launch_bounds__(256) void f16_residual_kernel(const float *X, int64_t row_begin, int64_t row_end, int D, const float *center,
                                                          uint32_t *stat)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = row_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= row_end) return;
    const float *src = X + row * (int64_t)D;
    float r2 = 0.f, n2 = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float v = center ? src[i] - center[i] : src[i];
        const float d = v - (float)(_Float16)v; // (exact: the two are within a factor of two of each other, or the second is 0)
        r2 = fmaf(d, d, r2);
        n2 = fmaf(v, v, n2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r2 += __shfl_xor(r2, off);
        n2 += __shfl_xor(n2, off);
    }
    if (lane == 0 && n2 > 0.f && r2 > 0.f) {
        // (f32 sums of D non-negative terms: relative error below (D + 8) 2^-24 each; the ratio is padded for both)
        const float ratio = (r2 / n2) * (1.0f + 4.0f * (float)(D + 8) * 5.9604645e-8f);
        atomicMax(stat, __builtin_bit_cast(uint32_t, ratio)); // (non-negative floats order as their bits; NaN / inf end up on top: the caller checks)
    }
}

} // namespace

void launch_f16_residual(const float *X, int64_t row_begin, int64_t row_end, int D, const float *center, uint32_t *stat, hipStream_t s)
{
    if (row_end <= row_begin) return;
    hipLaunchKernelGGL(f16_residual_kernel, dim3((unsigned)((row_end - row_begin + 3) / 4)), dim3(256), 0, s, X, row_begin, row_end, D, center, stat);
}

#ifdef LB_DIAG
void read_tall16_probe(unsigned long long out[8], bool reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tall16_probe), 8 * sizeof(unsigned long long));
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tall16_probe), z, sizeof z);
    }
}
#endif

int corpus_f16_plane_dims() { return H_XP; }

void launch_corpus_to_f16(const float *X, int64_t row_begin, int64_t row_end, int D, void *Xh, int64_t cap, hipStream_t s,
                          const float *center)
{
    if (row_end <= row_begin) return;
    dim3 grid((unsigned)((row_end - row_begin + 63) / 64), (unsigned)(((D + H_XP - 1) / H_XP) * (H_XP / 32))); // (whole planes: zero padding)
    hipLaunchKernelGGL(corpus_to_f16_kernel, grid, dim3(256), 0, s, X, row_begin, row_end, D, reinterpret_cast<_Float16 *>(Xh), cap, center);
}

void launch_queries_to_f16(const float *Q, int nq, int D, void *Qh, float *qinv, hipStream_t s)
{
    if (nq <= 0) return;
    hipLaunchKernelGGL(queries_to_f16_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, Q, nq, D,
                       reinterpret_cast<_Float16 *>(Qh), qinv);
}

// do the persistent kernels serve this launch (and, under a row list, leave POSITIONS in the candidate entries)?
static bool tall16_persistent_ok(int D, int nq, bool img, bool mapped, bool masked)
{
    static const int persist = lb_tunable("LB_F16_PERSIST", 1);
    static const int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    // (the persistent kernels take nearly all of a CU's 160 KB of LDS: on a device that offers a workgroup less, the
    // one-workgroup-per-tile kernel serves everything)
    static const int lds_max = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
        return n;
    }();
    static const int mapped_on = lb_tunable("LB_F16_MAPPED", 1);
    const int spx = cus / 8;
    if (!persist || masked || spx < 1 || (nq + H_BN - 1) / H_BN > spx || lds_max < 163840) return false;
    // a row list: gathered out of the image only, and with enough K-steps per tile for the row-id ring's hand-off (D >= 256)
    if (mapped && !(img && mapped_on && (D + H_BK - 1) / H_BK >= 8)) return false;
    return true;
}
bool tall16_runs_persistent(int D, int nq, bool img, bool mapped, bool masked) { return tall16_persistent_ok(D, nq, img, mapped, masked); }
bool tall16_tin_ok(int D, int nq, int64_t n_pos, bool img, bool mapped, bool masked, bool with_norm, uint32_t count, int m)
{
    static const int n16 = lb_tunable("LB_F16_NARROW", 1);
    static const int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    uint32_t dr;
    return img && n16 && nq <= 128 && tall16_persistent_ok(D, nq, img, mapped, masked) && tin_plan(D, nq, n_pos, (cus / 8) * 8, with_norm, count, m, dr);
}
bool tall16_entries_are_positions(int D, int nq, bool img, bool mapped, bool masked)
{
    return mapped && tall16_persistent_ok(D, nq, img, true, masked);
}

// Requires D % 32 == 0, 16-B aligned X / Qh; Qh / qinv from launch_queries_to_f16; X is the plain f32 corpus.
// One launch over a window of the batch: Qh, qinv and cs start at the window's first query, q_stride is the batch's row count
// (the K-block planes of the query image are that far apart).
static void tall16_window(int metric, const float *X, const float *norm2, const float *rnorm, int64_t row_begin, int64_t row_end,
                          int D, const void *Qh, const float *qinv, int nq, int q_stride, const uint8_t *mask,
                          const uint32_t *rowmap, CandState cs, bool boot, hipStream_t s, const void *Xh, int64_t xh_cap,
                          bool may_split, uint32_t gstride, const float *qnrm, float gsum, const Tall16Tin *tin = nullptr)
{
    if (row_end <= row_begin || nq <= 0) return;
    Tall16Args a;
    a.tin_fail = nullptr;
    a.qsplit = 0;
    a.gstride = gstride;
    a.qnrm = qnrm;
    a.gsum = gsum;
    a.rowmap = rowmap;
    a.X = X; a.norm2 = norm2; a.rnorm = rnorm; a.row_begin = row_begin; a.row_end = row_end; a.D = D;
    a.Qh = reinterpret_cast<const _Float16 *>(Qh); a.qinv = qinv; a.nq = nq; a.q_stride = q_stride; a.mask = mask; a.cs = cs; a.boot = boot ? 1 : 0;
    a.n_row_tiles = (int)((row_end - row_begin + H_BM - 1) / H_BM);
    a.n_q_tiles = (nq + H_BN - 1) / H_BN;
    a.abl = lb_tunable("LB_F16_ABL", 0);
    a.rot = lb_tunable("LB_F16_ROT", 0);
    a.Xh = reinterpret_cast<const _Float16 *>(Xh);
    a.xh_cap = xh_cap;
    const int groups = (a.n_row_tiles + 7) / 8;
    dim3 grid((unsigned)(groups * 8 * a.n_q_tiles));
    const size_t shmem = (size_t)H_NST * H_STAGE_BYTES + H_BM * 4 + H_BM * 4 + H_BM;
    // persistent form: one workgroup per CU (the ring leaves room for one), slots per XCD = CUs / 8; unfiltered searches, and
    // (over the image) searches over a row list
    static const int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    const int spx = cus / 8;
    const bool mapped = rowmap != nullptr;
    if (tall16_persistent_ok(D, nq, Xh != nullptr, mapped, mask != nullptr)) {
        const bool img = Xh != nullptr; // (sync_f16_image: in step with the corpus, K-blocked, xh_cap rows per plane)
        // A batch that ends 1 .. 64 queries into a 256-query tile: the whole tiles on the 256-wide kernel, the rest on the
        // one-tile kernel (a second pass over the image at its HBM rate, 0.24 ms per 1M x 768, instead of one more 256-wide
        // query tile that is mostly padding: 0.38 ms beside the others).  1M x 768: 257 queries 1.13 -> 0.98 ms, 320: 1.00 ->
        // 0.91, 640: 1.59 -> 1.53; a tail of 65 .. 128 goes to the 128-query tile (round 4: 352 queries 1.00 -> 0.95 ms, 384:
        // 0.90 -> 0.85, 640: 1.39 -> 1.34, 896 level).
        // Each launch sees its own window of the batch: the image, the scales and the candidate state from its first query
        // on; q_stride stays the batch's.
        const int tail = nq % H_BN;
        // (the sample of more than 128 queries: windows of 128 on the one-tile kernel in ONE launch, see the kernel)
        static const int qsplit_on = lb_tunable("LB_F16_SAMPLE_QSPLIT", 1);
        static const int qsplit_maxq = lb_tunable("LB_F16_SAMPLE_QSPLIT_MAXQ", 1024);
        static const int n16_on = lb_tunable("LB_F16_NARROW", 1);
        const bool qsplit = img && n16_on && boot && gstride != 0 && nq > 128 && nq <= qsplit_maxq && (nq + 127) / 128 <= spx * 8 && qsplit_on && tin == nullptr;
        a.qsplit = qsplit ? (nq + 127) / 128 : 0;
        static const int split_tail = lb_tunable("LB_F16_SPLIT_TAIL", 1);
        static const int split_tail_max = lb_tunable("LB_F16_SPLIT_TAIL_MAX", 128); // (65 .. 128 on the 128-query tile: 384 queries 0.90 -> 0.85 ms, 640: 1.39 -> 1.34)
        if (img && may_split && split_tail && nq > H_BN && tail >= 1 && tail <= split_tail_max && !qsplit) {
            const int head = nq - tail;
            tall16_window(metric, X, norm2, rnorm, row_begin, row_end, D, Qh, qinv, head, q_stride, mask, rowmap, cs, boot, s, Xh,
                          xh_cap, false, gstride, qnrm, gsum);
            CandState ct = cs;
            ct.lists += (size_t)head * cs.cap;
            ct.cnt += head;
            ct.tau += head;
            tall16_window(metric, X, norm2, rnorm, row_begin, row_end, D, reinterpret_cast<const _Float16 *>(Qh) + (size_t)head * H_BK,
                          qinv + head, tail, q_stride, mask, rowmap, ct, boot, s, Xh, xh_cap, false, gstride, qnrm ? qnrm + head : nullptr, gsum);
            return;
        }
        static const int n16 = lb_tunable("LB_F16_NARROW", 1);
        // (the 256-query instance of this kernel measured level with the 4 x 2-wave tile below -- 0.43 ms per pass, bound by MFMA +
        // LDS work either way -- and is not built)
        if (img && (nq <= 128 || qsplit) && n16) { // one query tile of 64 / 128: the pass is the image's HBM stream
            const int bn = (nq <= 64 && !qsplit) ? 64 : 128;
            const size_t ring_b = bn == 64 ? (size_t)6 * (H_BM * H_BK * 2 + 64 * H_BK * 2)
                                           : (bn == 128 ? (size_t)5 * (H_BM * H_BK * 2 + 128 * H_BK * 2) : (size_t)4 * (H_BM * H_BK * 2 + 256 * H_BK * 2));
            // ring, side inputs, flush flags + counters, admission segments
            const size_t nshmem = ring_b + 2 * 512 * sizeof(float) + 16 + 2 * 256 * 4 + 8 * (bn == 256 ? 272 : 360) * 12;
            dim3 ngrid((unsigned)(spx * 8));
            bool tauin = false;
            uint32_t dr = 0;
            if (tin && !boot && gstride == 0 && tin_plan(D, nq, row_end - row_begin, (int)ngrid.x, tin->qna != nullptr, tin->count, tin->m, dr)) {
                tauin = true; // thresholds inside the launch (see the kernel)
                a.tin_count = tin->count; a.tin_m = tin->m; a.tin_tag = tin->tag;
                a.tin_fail = tin->fail_host; a.tin_Q = tin->Q; a.tin_qna = tin->qna; a.tin_order = tin->order; a.tin_dr = dr;
            }
#define LB_NARROW16T(M, N, P)                                                                                                  \
    do {                                                                                                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_filter_narrow16p_kernel<M, N, false, P, true>),         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)nshmem);                                    \
        hipLaunchKernelGGL((gemm_filter_narrow16p_kernel<M, N, false, P, true>), ngrid, dim3(H_THREADS), nshmem, s, a, spx);   \
    } while (0)
#define LB_NARROW16(M, N, B, P)                                                                                        \
    do {                                                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_filter_narrow16p_kernel<M, N, B, P>),           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)nshmem);                            \
        hipLaunchKernelGGL((gemm_filter_narrow16p_kernel<M, N, B, P>), ngrid, dim3(H_THREADS), nshmem, s, a, spx);     \
    } while (0)
#define LB_NARROW16_B(M, B, P)                       \
    do {                                             \
        if (bn == 64) LB_NARROW16(M, 64, B, P);      \
        else LB_NARROW16(M, 128, B, P);              \
    } while (0)
#define LB_NARROW16_M(M)                                     \
    do {                                                     \
        if (tauin) {                                         \
            if (mapped) {                                    \
                if (bn == 64) LB_NARROW16T(M, 64, true);     \
                else LB_NARROW16T(M, 128, true);             \
            } else {                                         \
                if (bn == 64) LB_NARROW16T(M, 64, false);    \
                else LB_NARROW16T(M, 128, false);            \
            }                                                \
        } else if (mapped) {                                 \
            if (boot) LB_NARROW16_B(M, true, true);          \
            else LB_NARROW16_B(M, false, true);              \
        } else {                                             \
            if (boot) LB_NARROW16_B(M, true, false);         \
            else LB_NARROW16_B(M, false, false);             \
        }                                                    \
    } while (0)
            if (metric == METRIC_L2) LB_NARROW16_M(METRIC_L2);
            else if (metric == METRIC_COS) LB_NARROW16_M(METRIC_COS);
            else LB_NARROW16_M(METRIC_DOT);
#undef LB_NARROW16_M
#undef LB_NARROW16_B
#undef LB_NARROW16
#undef LB_NARROW16T
            return;
        }
        const size_t pshmem = (img ? (size_t)4 * (H_BM * H_BK * 2 + H_B_BYTES) : (size_t)H_NST * H_STAGE_BYTES) + 2 * 512 * sizeof(float) +
                              16 + 2 * H_BN * 4 + 8 * (img ? 272 : 104) * 12; // ring, side inputs, flush flags + counters, admission segments
        const bool pnt = a.n_q_tiles <= 1;
        static const int mapped_nt_on = lb_tunable("LB_F16_MAPPED_NT", 1); // (10 % visible, 256 queries: 243 -> 227 us; 50 %: 544 -> 535)
        const bool mapped_nt = mapped_nt_on && pnt;
        dim3 pgrid((unsigned)(spx * 8));
#define LB_TALL16P(M, N, I, B, P)                                                                                       \
    do {                                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_filter_tall16p_kernel<M, N, I, B, P>),           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)pshmem);                             \
        hipLaunchKernelGGL((gemm_filter_tall16p_kernel<M, N, I, B, P>), pgrid, dim3(H_THREADS), pshmem, s, a, spx);     \
    } while (0)
#define LB_TALL16P_M(M)                                        \
    do {                                                       \
        if (mapped) { /* (only over the image) */              \
            if (boot) LB_TALL16P(M, false, true, true, true);  \
            else if (mapped_nt) LB_TALL16P(M, true, true, false, true); \
            else LB_TALL16P(M, false, true, false, true);      \
        } else if (boot) { /* (a short launch: the cache policy does not matter) */ \
            if (img) LB_TALL16P(M, false, true, true, false);  \
            else LB_TALL16P(M, false, false, true, false);     \
        } else if (img) {                                      \
            if (pnt) LB_TALL16P(M, true, true, false, false);  \
            else LB_TALL16P(M, false, true, false, false);     \
        } else {                                               \
            if (pnt) LB_TALL16P(M, true, false, false, false); \
            else LB_TALL16P(M, false, false, false, false);    \
        }                                                      \
    } while (0)
        if (metric == METRIC_L2) LB_TALL16P_M(METRIC_L2);
        else if (metric == METRIC_COS) LB_TALL16P_M(METRIC_COS);
        else LB_TALL16P_M(METRIC_DOT);
#undef LB_TALL16P_M
#undef LB_TALL16P
        return;
    }
    static const int nt_max_tiles = lb_tunable("LB_F16_NT_MAXTILES", 1);
    const bool nt = a.n_q_tiles <= nt_max_tiles;
    static const int spread = lb_tunable("LB_F16_SPREAD", 1);
    static const int abl = lb_tunable("LB_F16_ABL", 0);
#define LB_TALL16(M, N, S, A)                                                                                    \
    do {                                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_filter_tall16_kernel<M, N, S, A>),        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); /* per device */      \
        hipLaunchKernelGGL((gemm_filter_tall16_kernel<M, N, S, A>), grid, dim3(H_THREADS), shmem, s, a);         \
    } while (0)
#ifdef LB_DIAG
    if (abl && metric == METRIC_COS) { // timing-only ablations / stamps (the bench metric only)
        if (nt) {
            if (abl == 1) LB_TALL16(METRIC_COS, true, true, 1);
            else if (abl == 2) LB_TALL16(METRIC_COS, true, true, 2);
            else if (abl == 3) LB_TALL16(METRIC_COS, true, true, 3);
            else LB_TALL16(METRIC_COS, true, true, 5);
        } else {
            if (abl == 1) LB_TALL16(METRIC_COS, false, true, 1);
            else if (abl == 2) LB_TALL16(METRIC_COS, false, true, 2);
            else if (abl == 3) LB_TALL16(METRIC_COS, false, true, 3);
            else LB_TALL16(METRIC_COS, false, true, 5);
        }
        return;
    }
#define LB_TALL16_M(M)                                   \
    do {                                                 \
        if (spread) {                                    \
            if (nt) LB_TALL16(M, true, true, 0);         \
            else LB_TALL16(M, false, true, 0);           \
        } else {                                         \
            if (nt) LB_TALL16(M, true, false, 0);        \
            else LB_TALL16(M, false, false, 0);          \
        }                                                \
    } while (0)
#else
#define LB_TALL16_M(M)                        \
    do {                                      \
        (void)spread;                         \
        (void)abl;                            \
        if (nt) LB_TALL16(M, true, true, 0);  \
        else LB_TALL16(M, false, true, 0);    \
    } while (0)
#endif
    if (metric == METRIC_L2) LB_TALL16_M(METRIC_L2);
    else if (metric == METRIC_COS) LB_TALL16_M(METRIC_COS);
    else LB_TALL16_M(METRIC_DOT);
#undef LB_TALL16_M
#undef LB_TALL16
}

void launch_gemm_filter_tall16(int metric, const float *X, const float *norm2, const float *rnorm, int64_t row_begin,
                               int64_t row_end, int D, const void *Qh, const float *qinv, int nq, const uint8_t *mask,
                               const uint32_t *rowmap, CandState cs, bool boot, hipStream_t s, const void *Xh, int64_t xh_cap,
                               uint32_t gstride, const float *qnrm, float gsum, const Tall16Tin *tin)
{
    // (the granule-strided sample view exists in the persistent forms only)
    if (gstride != 0 && (!boot || !tall16_persistent_ok(D, nq, Xh != nullptr, rowmap != nullptr, mask != nullptr))) return;
    tall16_window(metric, X, norm2, rnorm, row_begin, row_end, D, Qh, qinv, nq, nq, mask, rowmap, cs, boot, s, Xh, xh_cap, true, gstride, qnrm, gsum,
                  tin);
}

} // namespace lb
