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
                                                              const float *Q, float *tables)
{
    const int q = blockIdx.y;
    const int i = blockIdx.x;
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
    }
}

void launch_build_adc_table(const float *codebooks, int M, int K, int sub, const float *Q, int nq,
                            float *tables, hipStream_t s)
{
    if (nq <= 0) return;
    hipLaunchKernelGGL(build_adc_table_kernel, dim3(M, nq), dim3(256), 0, s, codebooks, M, K, sub, Q, tables);
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
};

template <bool VEC16>
__global__ __launch_bounds__(ADC_THREADS) void adc_scan_kernel(AdcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float tab[];
    const int M = a.M;
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
                const uint4 v = c4[g];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int j = g * 16 + t * 4 + b;
                        sum = sum + tab[j * 256 + ((w[t] >> (8 * b)) & 0xffu)];
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

void launch_adc_scan(const float *table, int M, const uint8_t *codes, int64_t row_begin, int64_t row_end,
                     int slot, const uint8_t *mask, CandState cs, bool boot, float *all_out,
                     int64_t out_base, hipStream_t s)
{
    if (row_end <= row_begin) return;
    AdcArgs a;
    a.boot = boot ? 1 : 0;
    a.table = table; a.M = M; a.codes = codes; a.row_begin = row_begin; a.row_end = row_end;
    a.slot = slot; a.mask = mask; a.cs = cs; a.all_out = all_out; a.out_base = out_base;
    const size_t shmem = (size_t)M * 256 * sizeof(float);
    const int64_t nrows = row_end - row_begin;
    int64_t blocks = (nrows + ADC_THREADS - 1) / ADC_THREADS;
    if (blocks > 256) blocks = 256; // one 1024-thread workgroup per CU holds the table once
    const bool vec = (M % 16 == 0) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0);
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
