// kernels_pq.hip -- product-quantisation asymmetric distance (ADC) on gfx950.
//
// build_adc_table: pq.BuildADCTable (internal/pq/adc_table.go:15-51):
//     table[i*K + j] = L2SquaredFloat32(q_sub_i, centroid_ij)   (4-accumulator order,
//     internal/simd/distance_functions.go:195-227), squared, no sqrt.
// adc_scan: simd.adcBatchGeneric (internal/simd/simd.go:345-355):
//     out[r] = float32(sqrt(float64(sum_j table[j*256 + codes[r*M + j]]))), f32 sum, j ascending.
// The whole M x 256 f32 table (98,304 B at M = 96) lives in LDS for the duration of the
// scan; each lane streams one code row (M bytes) from HBM and does M LDS gathers.
#include "lb_device.h"

#pragma clang fp contract(off)

namespace lb {

constexpr int ADC_THREADS = 1024;

__global__ __launch_bounds__(256) void build_adc_table_kernel(const float *codebooks, int M, int K, int sub,
                                                              const float *Q, float *tables, float *minrng,
                                                              uint32_t *zero_a, uint32_t *zero_b)
{
    const int q = blockIdx.y;
    const int i = blockIdx.x;
    // per-query status words of the search that follows (saves two memset launches on its critical path)
    if (i == 0 && threadIdx.x == 0) {
        if (zero_a) zero_a[q] = 0;
        if (zero_b) zero_b[q] = 0;
    }
    float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
    int bad = 0;
    const float *qs = Q + ((int64_t)q * M + i) * sub;
    const float *cb = codebooks + (int64_t)i * K * sub;
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        const float *c = cb + (int64_t)j * sub;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int t = 0;
        for (; t <= sub - 4; t += 4) {
            const float d0 = qs[t] - c[t], d1 = qs[t + 1] - c[t + 1];
            const float d2 = qs[t + 2] - c[t + 2], d3 = qs[t + 3] - c[t + 3];
            s0 = s0 + d0 * d0;
            s1 = s1 + d1 * d1;
            s2 = s2 + d2 * d2;
            s3 = s3 + d3 * d3;
        }
        for (; t < sub; t++) {
            const float d = qs[t] - c[t];
            s0 = s0 + d * d;
        }
        float r = s0 + s1;
        r = r + s2;
        r = r + s3;
        tables[((int64_t)q * M + i) * K + j] = r;
        mn = fminf(mn, r);
        mx = fmaxf(mx, r);
        if (!(r >= 0.f) || r > 3.0e38f) bad = 1;
    }
    if (minrng) { // subtable minimum / range for the byte-table prefilter (kernels_pq2.hip)
        __shared__ float s_mn[4], s_mx[4];
        __shared__ int s_bad[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off));
            mx = fmaxf(mx, __shfl_xor(mx, off));
            bad |= __shfl_xor(bad, off);
        }
        if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; s_bad[threadIdx.x >> 6] = bad; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float a = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
            const float b = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
            float *o = minrng + ((int64_t)q * M + i) * 4;
            o[0] = a;
            o[1] = b - a;
            o[2] = (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) ? 1.0f : 0.0f;
            o[3] = 0.f;
        }
    }
}

void launch_build_adc_table(const float *codebooks, int M, int K, int sub, const float *Q, int nq,
                            float *tables, hipStream_t s, float *minrng, uint32_t *zero_a, uint32_t *zero_b)
{
    if (nq <= 0) return;
    hipLaunchKernelGGL(build_adc_table_kernel, dim3(M, nq), dim3(256), 0, s, codebooks, M, K, sub, Q, tables, minrng,
                       zero_a, zero_b);
}

struct AdcArgs {
    const float *table; // [M*256] for this query
    int M;
    const uint8_t *codes;
    int64_t row_begin, row_end;
    int slot;
    const uint8_t *mask;
    CandState cs;
    float *all_out; // indexed by absolute row - out_base
    int64_t out_base;
    int boot;
    const int *skip_if_ok; // byte-table prefilter params {s_tau, ok}: ok != 0 -> nothing to do here
};

// ABL (profiling aid, wrong results): 1 = no LDS gathers, 2 = no global code loads
template <bool VEC16, int ABL = 0>
__global__ __launch_bounds__(ADC_THREADS) void adc_scan_kernel(AdcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float tab[];
    const int M = a.M;
    if (a.skip_if_ok && a.skip_if_ok[1] != 0) return;
    for (int i = threadIdx.x; i < M * 256; i += ADC_THREADS) tab[i] = a.table[i];
    __syncthreads();
    const uint64_t tau = a.all_out ? 0ull : a.cs.tau[a.slot];
    for (int64_t row = a.row_begin + (int64_t)blockIdx.x * ADC_THREADS + threadIdx.x; row < a.row_end;
         row += (int64_t)gridDim.x * ADC_THREADS) {
        const uint8_t *c = a.codes + row * (int64_t)M;
        float sum = 0.f;
        if (VEC16) {
            const uint4 *c4 = reinterpret_cast<const uint4 *>(c);
            for (int g = 0; g < M / 16; g++) {
                uint4 v;
                if (ABL == 2) v = make_uint4((uint32_t)row * 2654435761u, (uint32_t)row * 40503u + g, (uint32_t)(row >> 3) * 97u, (uint32_t)row ^ (g * 0x9e3779b9u));
                else v = c4[g];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int j = g * 16 + t * 4 + b;
                        if (ABL == 1) sum = sum + (float)((w[t] >> (8 * b)) & 0xffu);
                        else sum = sum + tab[j * 256 + ((w[t] >> (8 * b)) & 0xffu)];
                    }
                }
            }
        } else {
            for (int j = 0; j < M; j++) sum = sum + tab[j * 256 + c[j]];
        }
        const float dist = (float)sqrt((double)sum);
        if (a.all_out) {
            a.all_out[row - a.out_base] = dist;
        } else {
            const bool masked = a.mask && !a.mask[row];
            const uint64_t ent = pack_entry(dist, (uint32_t)row);
            if (a.boot) {
                a.cs.lists[(size_t)a.slot * a.cs.cap + (row - a.row_begin)] = masked ? kEntryMax : ent;
            } else if (!masked && ent < tau) {
                uint32_t pos = atomicAdd(&a.cs.cnt[a.slot], 1u);
                if (pos < a.cs.cap) a.cs.lists[(size_t)a.slot * a.cs.cap + pos] = ent;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Coalesced variant (M % 16 == 0): per-lane 16-B loads at an M-byte stride touch ~48 cache lines per
// wave instruction and cap the code stream at ~3.9 TB/s.  Here each wave pulls its 64 rows
// (64*M contiguous bytes) with M/16 direct-to-LDS DMA instructions of 1 KiB each
// (global_load_lds_dwordx4: full lines, no VGPR round trip) into a private staging slot, then every
// lane reads back its own row (M/16 x ds_read_b128) and does the M table gathers.  The DMA of the
// next 64 rows is issued as soon as the slot has been read, so it runs under the gather phase.
// LDS: table M*256*4 B + 10 waves * 64*M B  (98,304 + 61,440 B at M = 96); one workgroup per CU.
constexpr int ADC_DMA_WAVES = 8;
constexpr int ADC_DMA_SLOTS = 1; // staged 64-row tiles per wave (measured: 8x1 2.21 ms, 10x1 2.33, 4x2 3.03 at C4)

template <int MCH> // MCH = M / 16
__global__ __launch_bounds__(ADC_DMA_WAVES * 64) void adc_scan_dma_kernel(AdcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int M = MCH * 16;
    float *tab = smem_f;
    unsigned char *stage_all = reinterpret_cast<unsigned char *>(smem_f + M * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.skip_if_ok && a.skip_if_ok[1] != 0) return;
    for (int i = tid; i < M * 256; i += ADC_DMA_WAVES * 64) tab[i] = a.table[i];
    __syncthreads();
    unsigned char *stage = stage_all + wave * (ADC_DMA_SLOTS * 64 * M);
    const uint64_t tau = a.all_out ? 0ull : a.cs.tau[a.slot];
    const int64_t nrows = a.row_end - a.row_begin;
    const int64_t ntiles = (nrows + 63) / 64;
    const int64_t tstride = (int64_t)gridDim.x * ADC_DMA_WAVES;
    const unsigned char *last16 = a.codes + a.row_end * (int64_t)M - 16;

    auto issue = [&](int64_t tile, int slot) {
        const unsigned char *src0 = a.codes + (a.row_begin + tile * 64) * (int64_t)M + lane * 16;
#pragma unroll
        for (int i = 0; i < MCH; i++) {
            const unsigned char *src = src0 + i * 1024;
            if (src > last16) src = last16; // tail tile: stay inside the buffer (rows past the end are discarded)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(stage + slot * (64 * M) + i * 1024),
                                             16, 0, 2 /* nt: the codes stream through once */);
        }
    };

    int64_t tile = (int64_t)blockIdx.x * ADC_DMA_WAVES + wave;
    if (tile < ntiles) issue(tile, 0);
    if (ADC_DMA_SLOTS == 2 && tile + tstride < ntiles) issue(tile + tstride, 1);
    int slot = 0;
    for (; tile < ntiles; tile += tstride, slot = (ADC_DMA_SLOTS == 2) ? (slot ^ 1) : 0) {
        // two slots: the older of the (up to) two tiles in flight has landed once at most MCH DMAs remain
        if (ADC_DMA_SLOTS == 2 && tile + tstride < ntiles) {
            if (MCH == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (MCH == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (MCH == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (MCH == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (MCH == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char *sl = stage + slot * (64 * M);
        uint4 c[MCH];
#pragma unroll
        for (int i = 0; i < MCH; i++) c[i] = *reinterpret_cast<const uint4 *>(sl + lane * M + i * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // slot fully read before it is refilled
        if (tile + ADC_DMA_SLOTS * tstride < ntiles) issue(tile + ADC_DMA_SLOTS * tstride, slot);

        const int64_t row = a.row_begin + tile * 64 + lane;
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < MCH; g++) {
            const uint32_t w[4] = {c[g].x, c[g].y, c[g].z, c[g].w};
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int b = 0; b < 4; b++)
                    sum = sum + tab[(g * 16 + t * 4 + b) * 256 + ((w[t] >> (8 * b)) & 0xffu)];
        }
        if (row < a.row_end) {
            const float dist = (float)sqrt((double)sum);
            if (a.all_out) {
                a.all_out[row - a.out_base] = dist;
            } else {
                const bool masked = a.mask && !a.mask[row];
                const uint64_t ent = pack_entry(dist, (uint32_t)row);
                if (a.boot) {
                    a.cs.lists[(size_t)a.slot * a.cs.cap + (row - a.row_begin)] = masked ? kEntryMax : ent;
                } else if (!masked && ent < tau) {
                    uint32_t pos = atomicAdd(&a.cs.cnt[a.slot], 1u);
                    if (pos < a.cs.cap) a.cs.lists[(size_t)a.slot * a.cs.cap + pos] = ent;
                }
            }
        }
    }
}

template <int MCH>
static bool try_launch_adc_dma(const AdcArgs &a, hipStream_t s)
{
    constexpr int M = MCH * 16;
    const size_t shmem = (size_t)M * 256 * 4 + (size_t)ADC_DMA_WAVES * ADC_DMA_SLOTS * 64 * M;
    if (shmem > 160 * 1024) return false;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_scan_dma_kernel<MCH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    const int64_t ntiles = (a.row_end - a.row_begin + 63) / 64;
    int64_t blocks = (ntiles + ADC_DMA_WAVES - 1) / ADC_DMA_WAVES;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL((adc_scan_dma_kernel<MCH>), dim3((unsigned)blocks), dim3(ADC_DMA_WAVES * 64), shmem, s, a);
    return true;
}

// Exact ADC distances of `count` evenly spaced rows of [0, n) (sampled admission threshold, see
// index.hip: sample_plan): the table sits in LDS as in the scan kernels, one lane per sampled row.
__global__ __launch_bounds__(ADC_THREADS) void adc_sample_kernel(const float *table, int M, const uint8_t *codes, int64_t n,
                                                                 uint32_t count, uint64_t *out, int vec16)
{
    extern __shared__ __attribute__((aligned(16))) float tab[];
    for (int i = threadIdx.x; i < M * 256; i += ADC_THREADS) tab[i] = table[i];
    __syncthreads();
    for (uint32_t i = blockIdx.x * ADC_THREADS + threadIdx.x; i < count; i += gridDim.x * ADC_THREADS) {
        const int64_t row = (int64_t)(((uint64_t)i * (uint64_t)n) / count); // i, n < 2^32
        const uint8_t *c = codes + row * (int64_t)M;
        float sum = 0.f;
        if (vec16) {
            for (int g = 0; g < M / 16; g++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(c + g * 16);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int b2 = 0; b2 < 4; b2++)
                        sum = sum + tab[(g * 16 + t * 4 + b2) * 256 + ((w[t] >> (8 * b2)) & 0xffu)];
            }
        } else {
            for (int j = 0; j < M; j++) sum = sum + tab[j * 256 + c[j]];
        }
        out[i] = pack_entry((float)sqrt((double)sum), (uint32_t)row);
    }
}

void launch_adc_sample(const float *table, int M, const uint8_t *codes, int64_t n, uint32_t count, uint64_t *out,
                       hipStream_t s)
{
    if (count == 0) return;
    const int vec16 = (M % 16 == 0) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0);
    const size_t shmem = (size_t)M * 256 * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)shmem);
    uint32_t blocks = (count + ADC_THREADS - 1) / ADC_THREADS;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(adc_sample_kernel, dim3(blocks), dim3(ADC_THREADS), shmem, s, table, M, codes, n, count, out, vec16);
}

int g_adc_ablation = 0; // profiling aid (diagnostic build only; always 0 in the product build)

void launch_adc_scan(const float *table, int M, const uint8_t *codes, int64_t row_begin, int64_t row_end,
                     int slot, const uint8_t *mask, CandState cs, bool boot, float *all_out,
                     int64_t out_base, hipStream_t s, const int *skip_if_ok)
{
    if (row_end <= row_begin) return;
    AdcArgs a;
    a.skip_if_ok = skip_if_ok;
    a.boot = boot ? 1 : 0;
    a.table = table; a.M = M; a.codes = codes; a.row_begin = row_begin; a.row_end = row_end;
    a.slot = slot; a.mask = mask; a.cs = cs; a.all_out = all_out; a.out_base = out_base;
    const size_t shmem = (size_t)M * 256 * sizeof(float);
    const int64_t nrows = row_end - row_begin;
    int64_t blocks = (nrows + ADC_THREADS - 1) / ADC_THREADS;
    if (blocks > 256) blocks = 256; // one 1024-thread workgroup per CU holds the table once
    const bool vec = (M % 16 == 0) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0);
    if (vec && !g_adc_ablation && (a.row_end - a.row_begin) >= 4096) {
        bool ok = false;
        switch (M / 16) {
        case 1: ok = try_launch_adc_dma<1>(a, s); break;
        case 2: ok = try_launch_adc_dma<2>(a, s); break;
        case 3: ok = try_launch_adc_dma<3>(a, s); break;
        case 4: ok = try_launch_adc_dma<4>(a, s); break;
        case 6: ok = try_launch_adc_dma<6>(a, s); break;
        case 8: ok = try_launch_adc_dma<8>(a, s); break;
        default: break;
        }
        if (ok) return;
    }
#ifdef LB_DIAG
    if (vec && g_adc_ablation) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_scan_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_scan_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (g_adc_ablation == 1) hipLaunchKernelGGL((adc_scan_kernel<true, 1>), dim3((unsigned)blocks), dim3(ADC_THREADS), shmem, s, a);
        else hipLaunchKernelGGL((adc_scan_kernel<true, 2>), dim3((unsigned)blocks), dim3(ADC_THREADS), shmem, s, a);
        return;
    }
#endif
    if (vec) {
        if (shmem > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_scan_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(adc_scan_kernel<true>, dim3((unsigned)blocks), dim3(ADC_THREADS), shmem, s, a);
    } else {
        if (shmem > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_scan_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(adc_scan_kernel<false>, dim3((unsigned)blocks), dim3(ADC_THREADS), shmem, s, a);
    }
}

} // namespace lb
