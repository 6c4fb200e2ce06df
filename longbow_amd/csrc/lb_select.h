// lb_select.h -- device helpers shared by the selection / re-rank kernels (kernels_select.hip, kernels_finish.hip).
#pragma once
#include "lb_device.h"

// every f32 operation of the exact re-rank is one IEEE rounding, as the Go source spells it: no FMA contraction in the
// helpers below nor in the files that include them
#pragma clang fp contract(off)

namespace lb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SEL_THREADS = 256;

__device__ __forceinline__ uint32_t next_pow2(uint32_t v)
{
    if (v <= 2) return 2;
    return 1u << (32 - __builtin_clz(v - 1));
}

// In-LDS bitonic sort of P (power of two) u64 keys, ascending, by the whole workgroup.
__device__ __forceinline__ void bitonic_sort_u64(uint64_t *sh, uint32_t P, int tid, int nthreads)
{
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (P >> 1); t += nthreads) {
                const uint32_t i = 2 * t - (t & (j - 1));
                const uint32_t l = i + j;
                const uint64_t a = sh[i], b = sh[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    sh[i] = b;
                    sh[l] = a;
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}


// exact-order accumulators (same as kernels_scan.hip)
template <int ORDER>
struct AccR {
    float s[ORDER == ORDER_UNROLL4 ? 4 : 1];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int i = 0; i < (ORDER == ORDER_UNROLL4 ? 4 : 1); i++) s[i] = 0.f;
    }
    template <int T>
    __device__ __forceinline__ void add(float v)
    {
        if (ORDER == ORDER_UNROLL4) s[T] = s[T] + v;
        else s[0] = s[0] + v;
    }
    __device__ __forceinline__ void add_tail(float v) { s[0] = s[0] + v; }
    __device__ __forceinline__ float total() const
    {
        if (ORDER == ORDER_UNROLL4) {
            float t = s[0] + s[1];
            t = t + s[2];
            t = t + s[3];
            return t;
        }
        return s[0];
    }
};


} // namespace lb
