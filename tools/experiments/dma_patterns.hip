// dma_patterns.hip -- what does the lane -> address pattern of a global -> LDS staging request cost on MI355X?
//
// One 512-thread workgroup per CU stages "corpus tiles" of 256 rows x 128 B per K-step (row stride 3072 B = 768 floats, 24
// K-steps per tile) exactly as the candidate kernels do, and nothing else: no LDS reads, no MFMAs.  Modes:
//   0  LDS-DMA, piece = 8 rows x 128 B, 16-B chunks XOR-swizzled by (row >> 1) & 7         (what the kernels do today)
//   1  LDS-DMA, same piece, chunks in address order                                        (bank-conflicted image; reference)
//   2  LDS-DMA, piece = 1 KiB contiguous                                                   (reference: best case for the TA)
//   3  global_load_dwordx4 into registers, address order, no LDS write                     (register path, loads only)
//   4  global_load_dwordx4 -> ds_write_b128 at the swizzled position                       (register staging, f32 image)
//   5  LDS-DMA, XOR by ((row >> 1) & 1) << 2 only                                          (quads of lanes stay in order)
//   6  LDS-DMA, XOR by ((row >> 1) & 3) << 1 only                                          (pairs of lanes stay in order)
//   7  global_load_dwordx4 -> v_cvt_pk_f16_f32 -> ds_write_b64 at a swizzled fp16 position (register staging, fp16 image)
// source = 0: every workgroup of an XCD reads the same tile over and over (L2 hits); 1: the tiles of a 3 GB buffer, once each.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_patterns tools/experiments/dma_patterns.hip ; run: /tmp/dma_patterns
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int ROWS = 256, D = 768, NK = D / 32, THREADS = 512;
constexpr int STAGE = ROWS * 128; // 32 KB
constexpr int NST = 3;

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void dma16(const void *g, uint32_t lds_addr)
{
    uint32_t save;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(g), "s"(lds_addr));
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(g), "s"(lds_addr));
}
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

template <int MODE, bool NT>
__global__ __launch_bounds__(THREADS, 2) void stage_kernel(const float *X, int64_t n_tiles, int tiles_per_wg, int shared_src, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x4 accv = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < tiles_per_wg; t++) {
        int64_t tile = shared_src ? (int64_t)(blockIdx.x & 7) : ((int64_t)blockIdx.x + (int64_t)t * gridDim.x) % n_tiles;
        const float *T = X + tile * (int64_t)ROWS * D;
        // this wave's four requests per K-step: rows 32 wave + 8 i .. + 7
        const unsigned char *src[4];
        uint32_t dst_off[4]; // (register modes) LDS byte offset of this lane's 16 B inside a stage
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = wave * 32 + i * 8 + (lane >> 3);
            int c = lane & 7;
            if (MODE == 0) c ^= (row >> 1) & 7;
            if (MODE == 5) c ^= ((row >> 1) & 1) << 2;
            if (MODE == 6) c ^= ((row >> 1) & 3) << 1;
            if (MODE == 2) // 1 KiB contiguous: "row" = 1 KiB segment of the tile's first rows
                src[i] = reinterpret_cast<const unsigned char *>(T) + (size_t)(wave * 4 + i) * 3072 * 8 + lane * 16;
            else
                src[i] = reinterpret_cast<const unsigned char *>(T + (int64_t)row * D) + 16 * c;
            dst_off[i] = (uint32_t)(row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
            if (MODE == 7) dst_off[i] = (uint32_t)(row * 64 + (((lane & 7) ^ ((row >> 2) & 7)) << 3)); // fp16 image: 64-B rows
        }
        if (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 5 || MODE == 6) {
            for (int kt = 0; kt < NK; kt++) {
                const uint32_t A = ring + (uint32_t)(kt % NST) * STAGE + (uint32_t)(wave * 32 * 128);
                const int kb = MODE == 2 ? kt * 1024 % 3072 : kt * 128;
#pragma unroll
                for (int i = 0; i < 4; i++) dma16<NT>(src[i] + kb, A + 1024u * i);
                wait_vmcnt<8>(); // two stages of this wave's requests stay in flight
                if (MODE != 2) __builtin_amdgcn_s_barrier();
            }
            wait_vmcnt<0>();
        } else {
            f32x4 v[2][4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[0][i] = *reinterpret_cast<const f32x4 *>(src[i]);
#pragma unroll
            for (int kt = 0; kt < NK; kt++) { // (fully unrolled: `cur` is a constant in every copy)
                const int cur = kt & 1;
                if (kt + 1 < NK) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const f32x4 *p = reinterpret_cast<const f32x4 *>(src[i] + (kt + 1) * 128);
                        if (cur == 0) v[1][i] = NT ? __builtin_nontemporal_load(p) : *p;
                        else v[0][i] = NT ? __builtin_nontemporal_load(p) : *p;
                    }
                }
                unsigned char *S = lds + (kt % NST) * STAGE;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const f32x4 x = cur == 0 ? v[0][i] : v[1][i];
                    if (MODE == 3) accv += x;
                    if (MODE == 4) *reinterpret_cast<f32x4 *>(S + dst_off[i]) = x;
                    if (MODE == 7) {
                        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                        h4 hv;
                        hv[0] = (_Float16)x.x; hv[1] = (_Float16)x.y; hv[2] = (_Float16)x.z; hv[3] = (_Float16)x.w;
                        *reinterpret_cast<h4 *>(S + dst_off[i]) = hv;
                    }
                }
                if (MODE != 3) __builtin_amdgcn_s_barrier();
            }
        }
    }
    if (accv.x + accv.y + accv.z + accv.w == 12345.678f) sink[tid] = accv.x; // keep the loads
    if (tid == 0 && lds[blockIdx.x & 1023] == 77 && sink[0] == 1.5f) sink[1] = 1.f; // keep the LDS writes
}


// ---- the fp16 candidate kernel's main loop rebuilt piece by piece (same grid, same tile walk, same requests) ---------------
// PARTS bit 0: corpus requests (4 per wave and K-step)   bit 1: query requests (2 per wave and K-step: 16 rows x 64 B)
//       bit 2: the 16 LDS fragment reads of a K-step     bit 3: the 16 MFMAs (on whatever the registers hold)
//       bit 4: requests spread between the MFMA pairs instead of one burst behind the barrier
//       bit 5: query image K-blocked ([k-block][query][32 halfs]: a K-step's 256 x 64 B are contiguous)
//       bit 6: L2 prefetch: one dword per 128-B line, `pf` K-steps ahead (into the successor block's tile near the end), by LDS-DMA into a junk
//              area (no register is written later); the sharers of a corpus tile take turns
// rot: the query tiles of one corpus tile walk K rotated by rot K-steps against each other (they do not ask for the same line at once)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int T_STAGE = 48 * 1024;

template <int PARTS, bool NT>
__global__ __launch_bounds__(THREADS, 2) void tile_kernel(const float *X, const _Float16 *Qh, int n_row_tiles, int n_q_tiles, float *sink, int rot, int pf)
{
    const int b = blockIdx.x;
    const int xcd = b & 7, in_xcd = b >> 3;
    const int qt = in_xcd % n_q_tiles;
    const int rt = (in_xcd / n_q_tiles) * 8 + xcd;
    if (rt >= n_row_tiles) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
    const unsigned char *srcA[4], *srcB[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        srcA[i] = reinterpret_cast<const unsigned char *>(X + ((int64_t)rt * ROWS + row) * D) + 16 * c;
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int row = wave * 32 + j * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        srcB[j] = (PARTS & 32) ? reinterpret_cast<const unsigned char *>(Qh) + ((int64_t)qt * 256 + row) * 64 + 16 * c
                               : reinterpret_cast<const unsigned char *>(Qh + ((int64_t)qt * 256 + row) * D) + 16 * c;
    }
    const int krot = (qt * rot) % NK;
    const int64_t kb_stride = (PARTS & 32) ? (int64_t)n_q_tiles * 256 * 64 : 64;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    constexpr int NI = ((PARTS & 1) ? 4 : 0) + ((PARTS & 2) ? 2 : 0);
    auto piece = [&](int kt, int p) { // request p (0..3 corpus, 4..5 queries) of stage kt
        const uint32_t A = ring + (uint32_t)(kt % 3) * T_STAGE + (uint32_t)(wave * 32 * 128);
        const uint32_t B = ring + (uint32_t)(kt % 3) * T_STAGE + 32768u + (uint32_t)(wave * 32 * 64);
        int ke = kt + krot;
        if (ke >= NK) ke -= NK;
        if (p < 4) { if (PARTS & 1) dma16<NT>(srcA[p] + ke * 128, A + 1024u * p); }
        else if (PARTS & 2) dma16<false>(srcB[p - 4] + ke * kb_stride, B + 1024u * (p - 4));
    };
    auto issue = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 6; p++) piece(kt, p);
    };
    // prefetch source: lane l < 32: row 32 wave + l at K-step kt + pf; lanes 32..63: the same rows one K-step further
    const int64_t succ_rows = (int64_t)(32 / n_q_tiles) * 8 * ROWS; // the tile of block b + 256 (same XCD, same query tile)
    const unsigned char *pfsrc = reinterpret_cast<const unsigned char *>(X + ((int64_t)rt * ROWS + wave * 32 + (lane & 31)) * D) + (lane >> 5) * 128;
    const uint32_t junk = ring + 3 * T_STAGE + (uint32_t)wave * 256u;
    auto prefetch = [&](int kt) {
        int ke = kt + pf;
        const unsigned char *g = pfsrc;
        if (ke >= NK) { ke -= NK; g += succ_rows * D * 4; if ((int64_t)rt * ROWS + succ_rows >= (int64_t)n_row_tiles * ROWS) return; }
        uint32_t save;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(save) : "v"(g + ke * 128), "s"(junk));
    };
    issue(0);
    issue(1);
#pragma unroll 1
    for (int kt = 0; kt < NK; kt++) {
        if (kt + 1 < NK) wait_vmcnt<NI>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool pre = kt + 2 < NK;
        if (pre && !(PARTS & 16)) issue(kt + 2);
        if ((PARTS & 64) && !(kt & 1) && ((kt >> 1) % n_q_tiles) == qt) prefetch(kt);
        const unsigned char *As = lds + (kt % 3) * T_STAGE;
        const unsigned char *Bs = As + 32768;
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            f16x8 af[2];
#pragma unroll
            for (int tm = 0; tm < 2; tm++) {
                if (PARTS & 4) {
                    const int r = wr * 64 + tm * 32 + l31;
                    const f32x4 x0 = *reinterpret_cast<const f32x4 *>(As + r * 128 + (((4 * kb + 2 * h) ^ ((r >> 1) & 7)) << 4));
                    const f32x4 x1 = *reinterpret_cast<const f32x4 *>(As + r * 128 + (((4 * kb + 2 * h + 1) ^ ((r >> 1) & 7)) << 4));
                    f16x8 t;
                    t[0] = (_Float16)x0.x; t[1] = (_Float16)x0.y; t[2] = (_Float16)x0.z; t[3] = (_Float16)x0.w;
                    t[4] = (_Float16)x1.x; t[5] = (_Float16)x1.y; t[6] = (_Float16)x1.z; t[7] = (_Float16)x1.w;
                    af[tm] = t;
                } else {
                    for (int e = 0; e < 8; e++) af[tm][e] = (_Float16)(float)(lane + e);
                }
            }
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                f16x8 bf;
                if (PARTS & 4) {
                    const int r = wc * 128 + tn * 32 + l31;
                    bf = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(Bs + r * 64 + (((2 * kb + h) ^ ((r >> 2) & 3)) << 4)));
                } else {
                    for (int e = 0; e < 8; e++) bf[e] = (_Float16)(float)(lane - e);
                }
                if (PARTS & 8) {
#pragma unroll
                    for (int tm = 0; tm < 2; tm++)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tm], bf, acc[tm][tn], 0, 0, 0);
                } else if (PARTS & 4) {
                    acc[0][tn][0] += (float)af[0][0] + (float)af[1][0] + (float)bf[0]; // keep the reads
                }
                if (pre && (PARTS & 16)) {
                    const int slot = kb * 4 + tn;
                    if (slot < 3) piece(kt + 2, slot);
                    else if (slot == 4) piece(kt + 2, 3);
                    else if (slot == 5) piece(kt + 2, 4);
                    else if (slot == 6) piece(kt + 2, 5);
                }
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) sum += acc[i][j][r];
    if (sum == 12345.678f) sink[tid] = sum;
}

template <int PARTS, bool NT>
static double run_tile(const float *X, const _Float16 *Qh, int n_row_tiles, int n_q_tiles, float *sink, int rot = 0, int pf = 0)
{
    const size_t shmem = 3 * T_STAGE + 2048;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(tile_kernel<PARTS, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const int grid = ((n_row_tiles + 7) / 8) * 8 * n_q_tiles;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((tile_kernel<PARTS, NT>), dim3(grid), dim3(THREADS), shmem, 0, X, Qh, n_row_tiles, n_q_tiles, sink, rot, pf);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((tile_kernel<PARTS, NT>), dim3(grid), dim3(THREADS), shmem, 0, X, Qh, n_row_tiles, n_q_tiles, sink, rot, pf);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int MODE, bool NT>
static double run(const float *X, int64_t n_tiles, int tiles_per_wg, int shared_src, float *sink, int grid)
{
    const size_t shmem = (size_t)NST * STAGE;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stage_kernel<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((stage_kernel<MODE, NT>), dim3(grid), dim3(THREADS), shmem, 0, X, n_tiles, tiles_per_wg, shared_src, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((stage_kernel<MODE, NT>), dim3(grid), dim3(THREADS), shmem, 0, X, n_tiles, tiles_per_wg, shared_src, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / reps;
}

int main()
{
    const int64_t n_tiles = 3906; // 1M rows x 768 f32 = 3.07 GB
    float *X, *sink;
    CK(hipMalloc(&X, (size_t)n_tiles * ROWS * D * 4));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(X, 0, (size_t)n_tiles * ROWS * D * 4));
    CK(hipMemset(sink, 0, 4096));
    const int grid = 256;
    const char *names[8] = {"DMA 8x128B xor-swizzled (today)", "DMA 8x128B address order", "DMA 1 KiB contiguous", "regs, loads only",
                            "regs -> ds_write_b128 swizzled", "DMA xor bit 2 only (quads in order)", "DMA xor bits 1-2 (pairs in order)",
                            "regs -> cvt f16 -> ds_write_b64"};
    for (int shared = 1; shared >= 0; shared--) {
        const int tiles_per_wg = shared ? 64 : 15; // 15 x 256 = 3840 of the 3906 tiles, each once
        const double bytes = (double)grid * tiles_per_wg * NK * STAGE;
        printf("source: %s   (%.2f GB per launch, one 512-thread workgroup per CU, 32 KB per K-step)\n",
               shared ? "one tile per XCD, re-read (L2 hits)" : "3 GB buffer, every tile once (HBM)", bytes / 1e9);
#define ROW(M, N)                                                                                                     \
    {                                                                                                                 \
        const double ms = run<M, N>(X, n_tiles, tiles_per_wg, shared, sink, grid);                                     \
        printf("  mode %d %-40s %s  %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip\n", M, names[M], N ? "nt" : "  ", ms, \
               bytes / ms / 1e6 / grid, bytes / ms / 1e9);                                                            \
    }
        ROW(0, false) ROW(1, false) ROW(2, false) ROW(5, false) ROW(6, false) ROW(3, false) ROW(4, false) ROW(7, false)
        if (!shared) { ROW(0, true) ROW(1, true) ROW(3, true) ROW(4, true) ROW(7, true) }
    }
    // ---- the fp16 candidate kernel's loop, part by part, at its real grid (1M rows x 1024 / 256 queries)
    _Float16 *Qh;
    CK(hipMalloc(&Qh, (size_t)1024 * D * 2));
    CK(hipMemset(Qh, 0, (size_t)1024 * D * 2));
    for (int nq = 4; nq >= 1; nq -= 3) {
        printf("tile loop at the kernel's grid: 3906 corpus tiles x %d query tiles of 256 (MFMA floor %.3f ms at 2.5 PF)\n", nq,
               2.0 * 3906 * 256 * 256.0 * nq * D / 2.5e15 * 1e3);
#define TROW(P, N, what) printf("  parts %2d %-64s %s %8.3f ms\n", P, what, N ? "nt" : "  ", run_tile<P, N>(X, Qh, (int)n_tiles, nq, sink));
        TROW(1, false, "corpus requests only (burst)")
        TROW(2, false, "query requests only (burst)")
        TROW(3, false, "corpus + query requests (burst)")
        TROW(19, false, "corpus + query requests (spread slots, nothing between)")
        TROW(7, false, "requests (burst) + LDS fragment reads")
        TROW(8, false, "MFMAs only (registers)")
        TROW(12, false, "LDS reads + MFMAs, no requests")
        TROW(15, false, "everything, burst")
        TROW(31, false, "everything, spread")
        if (nq == 1) { TROW(3, true, "corpus + query requests (burst)") TROW(31, true, "everything, spread") }
        TROW(34, false, "query requests only, K-blocked image")
        TROW(35, false, "corpus + query requests, K-blocked query image")
        TROW(63, false, "everything, spread, K-blocked query image")
        if (nq == 4)
            for (int pf = 4; pf <= 12; pf += 2)
                printf("  L2 prefetch %2d K-steps ahead: corpus only %8.3f ms   corpus + K-blocked queries %8.3f   everything spread, K-blocked Q %8.3f\n", pf,
                       run_tile<65, false>(X, Qh, (int)n_tiles, nq, sink, 0, pf), run_tile<99, false>(X, Qh, (int)n_tiles, nq, sink, 0, pf),
                       run_tile<127, false>(X, Qh, (int)n_tiles, nq, sink, 0, pf));
        if (nq == 4)
            for (int rot = 1; rot <= 6; rot += (rot < 3 ? 1 : 3)) {
                printf("  rot %d: corpus only %8.3f ms   corpus + queries %8.3f   everything spread %8.3f   everything, K-blocked Q %8.3f\n", rot,
                       run_tile<1, false>(X, Qh, (int)n_tiles, nq, sink, rot), run_tile<3, false>(X, Qh, (int)n_tiles, nq, sink, rot),
                       run_tile<31, false>(X, Qh, (int)n_tiles, nq, sink, rot), run_tile<63, false>(X, Qh, (int)n_tiles, nq, sink, rot));
            }
    }
    return 0;
}
