// kernels_filter.hip -- metadata predicate masks (SURVEY f-3).
//
// simd.MatchInt64 / simd.MatchFloat32 (internal/simd/simd.go:570-761): dst[i] = (src[i] OP val) ? 1 : 0,
// one byte per element, OP in {Eq, Neq, Gt, Ge, Lt, Le} (simd.CompareOp, simd.go:38-45);
// simd.AndBytes (simd.go:119-125): dst[i] &= src[i];
// nulls -> 0 as query.*FilterOp.MatchBitmap does (internal/query/filter_evaluator.go:106-114,232-240),
// with the validity given as an Arrow LSB-first bitmap.
// Pure HBM-bound byte/integer work: 16 elements per lane, 16-B mask stores.
#include "lb_device.h"

namespace lb {

enum : int { OP_EQ = 0, OP_NEQ = 1, OP_GT = 2, OP_GE = 3, OP_LT = 4, OP_LE = 5 };

template <typename T>
__device__ __forceinline__ uint8_t cmp(T v, T val, int op)
{
    switch (op) {
    case OP_EQ: return v == val;
    case OP_NEQ: return v != val;
    case OP_GT: return v > val;
    case OP_GE: return v >= val;
    case OP_LT: return v < val;
    default: return v <= val;
    }
}

// combine: 0 = dst = m, 1 = dst &= m
template <typename T>
__global__ __launch_bounds__(256) void match_kernel(const T *src, int64_t n, T val, int op,
                                                    const uint8_t *validity, int64_t valid_offset,
                                                    uint8_t *dst, int combine)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 16;
    for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; base < n; base += stride) {
        uint8_t m[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int64_t i = base + j;
            uint8_t r = 0;
            if (i < n) {
                r = cmp<T>(src[i], val, op);
                if (validity) {
                    const int64_t b = i + valid_offset;
                    r &= (validity[b >> 3] >> (b & 7)) & 1u;
                }
            }
            m[j] = r;
        }
        if (base + 16 <= n && ((reinterpret_cast<uintptr_t>(dst + base) & 15) == 0)) {
            uint4 out;
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                w[k] = (uint32_t)m[4 * k] | ((uint32_t)m[4 * k + 1] << 8) | ((uint32_t)m[4 * k + 2] << 16) |
                       ((uint32_t)m[4 * k + 3] << 24);
            out = make_uint4(w[0], w[1], w[2], w[3]);
            uint4 *p = reinterpret_cast<uint4 *>(dst + base);
            if (combine) {
                const uint4 old = *p;
                out.x &= old.x; out.y &= old.y; out.z &= old.z; out.w &= old.w;
            }
            *p = out;
        } else {
            for (int j = 0; j < 16 && base + j < n; j++)
                dst[base + j] = combine ? (uint8_t)(dst[base + j] & m[j]) : m[j];
        }
    }
}

__global__ __launch_bounds__(256) void and_bytes_kernel(uint8_t *dst, const uint8_t *src, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] &= src[i];
}

// ---- ordered stream compaction of the predicate mask --------------------------------------------
// A selective filter (BASELINE config 5: 10 % of the rows pass) should not pay distance work for
// the rows it hides.  These three kernels turn the byte mask into the ascending list of visible
// corpus rows; the search kernels then walk *positions* of that list and fetch rows through it,
// so hidden rows cost neither HBM reads nor MFMA/VALU work.  Ascending order keeps the canonical
// (distance, row) tie-break unchanged.  Replaces the per-row `if bitmap.Contains(id)` test of
// internal/store/adaptive_index.go:180-186 / the mask walk of filter_evaluator.go:79-115.
constexpr int CP_THREADS = 256;
constexpr int CP_PER = 8; // mask bytes per thread
constexpr int CP_ROWS = CP_THREADS * CP_PER;

__device__ __forceinline__ uint32_t cp_load8(const uint8_t *mask, int64_t n, int64_t base)
{
    // bit i set <=> mask[base + i] != 0 (bytes past n read as 0)
    uint32_t bits = 0;
    if (base + CP_PER <= n) {
        const uint64_t v = *reinterpret_cast<const uint64_t *>(mask + base); // base % 8 == 0, mask is 256-B aligned
#pragma unroll
        for (int i = 0; i < CP_PER; i++) bits |= ((v >> (8 * i)) & 0xffull) ? (1u << i) : 0u;
    } else {
        for (int i = 0; i < CP_PER; i++)
            if (base + i < n && mask[base + i]) bits |= 1u << i;
    }
    return bits;
}

// inclusive scan of one value per thread over the 256-thread block; returns this thread's inclusive sum
__device__ __forceinline__ uint32_t cp_block_scan(uint32_t v, uint32_t *s_wave /*[4]*/, uint32_t &block_total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CP_THREADS / 64; w++) {
        const uint32_t t = s_wave[w];
        if (w < wave) before += t;
        tot += t;
    }
    block_total = tot;
    return x + before;
}

__global__ __launch_bounds__(CP_THREADS) void compact_count_kernel(const uint8_t *mask, int64_t n, uint32_t *block_counts)
{
    __shared__ uint32_t s_wave[CP_THREADS / 64];
    const int64_t base = ((int64_t)blockIdx.x * CP_THREADS + threadIdx.x) * CP_PER;
    const uint32_t c = base < n ? (uint32_t)__popc(cp_load8(mask, n, base)) : 0u;
    uint32_t tot;
    (void)cp_block_scan(c, s_wave, tot);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = tot;
}

// exclusive scan of the per-block counts in place (one workgroup; nb is a few hundred to a few
// hundred thousand), total to block_counts[nb]
__global__ __launch_bounds__(CP_THREADS) void compact_offsets_kernel(uint32_t *block_counts, int64_t nb)
{
    __shared__ uint32_t s_wave[CP_THREADS / 64];
    uint32_t running = 0;
    for (int64_t b0 = 0; b0 < nb; b0 += CP_THREADS) {
        const int64_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? block_counts[i] : 0u;
        uint32_t tot;
        const uint32_t inc = cp_block_scan(v, s_wave, tot);
        if (i < nb) block_counts[i] = running + inc - v;
        running += tot;
        __syncthreads(); // s_wave is rewritten by the next round
    }
    if (threadIdx.x == 0) block_counts[nb] = running;
}

__global__ __launch_bounds__(CP_THREADS) void compact_scatter_kernel(const uint8_t *mask, int64_t n,
                                                                    const uint32_t *block_offsets, uint32_t *rowmap)
{
    __shared__ uint32_t s_wave[CP_THREADS / 64];
    const int64_t base = ((int64_t)blockIdx.x * CP_THREADS + threadIdx.x) * CP_PER;
    const uint32_t bits = base < n ? cp_load8(mask, n, base) : 0u;
    const uint32_t c = (uint32_t)__popc(bits);
    uint32_t tot;
    const uint32_t inc = cp_block_scan(c, s_wave, tot);
    uint32_t at = block_offsets[blockIdx.x] + inc - c;
#pragma unroll
    for (int i = 0; i < CP_PER; i++)
        if (bits & (1u << i)) rowmap[at++] = (uint32_t)(base + i);
}

int64_t compact_scratch_words(int64_t n) { return (n + CP_ROWS - 1) / CP_ROWS + 1; }

void launch_compact_mask(const uint8_t *mask, int64_t n, uint32_t *rowmap, uint32_t *scratch, hipStream_t s)
{
    const int64_t nb = (n + CP_ROWS - 1) / CP_ROWS;
    if (nb <= 0) {
        (void)hipMemsetAsync(scratch, 0, sizeof(uint32_t), s);
        return;
    }
    hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nb), dim3(CP_THREADS), 0, s, mask, n, scratch);
    hipLaunchKernelGGL(compact_offsets_kernel, dim3(1), dim3(CP_THREADS), 0, s, scratch, nb);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)nb), dim3(CP_THREADS), 0, s, mask, n, scratch, rowmap);
}

static unsigned grid_for(int64_t n, int per_thread)
{
    int64_t blocks = (n + 256 * per_thread - 1) / (256 * per_thread);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

void launch_match_int64(const int64_t *src, int64_t n, int64_t val, int op, const uint8_t *validity,
                        int64_t valid_offset, uint8_t *dst, int combine, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(match_kernel<int64_t>, dim3(grid_for(n, 16)), dim3(256), 0, s, src, n, val, op, validity,
                       valid_offset, dst, combine);
}

void launch_match_float32(const float *src, int64_t n, float val, int op, const uint8_t *validity,
                          int64_t valid_offset, uint8_t *dst, int combine, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(match_kernel<float>, dim3(grid_for(n, 16)), dim3(256), 0, s, src, n, val, op, validity,
                       valid_offset, dst, combine);
}

void launch_and_bytes(uint8_t *dst, const uint8_t *src, int64_t n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(and_bytes_kernel, dim3(grid_for(n, 1)), dim3(256), 0, s, dst, src, n);
}

// ---------------------------------------------------------------------------
// Reciprocal-rank fusion (SURVEY f-4): store.ReciprocalRankFusion (internal/store/rrf.go:10-51).
// score(id) = sum over the lists containing id of 1 / float64(k + rank + 1), rank 0-based, accumulated
// dense first then sparse in f64, narrowed to f32; output sorted by score descending (ties: lower id
// first -- the reference's sort is unstable over a map, so any order of ties is "the reference's").
// One workgroup per query; ids are unique within a list (search results); -1 entries are padding.
__global__ __launch_bounds__(256) void rrf_kernel(int64_t nq, int kd, const int64_t *dense, int ks,
                                                  const int64_t *sparse, int k, int limit, int64_t *out_ids,
                                                  float *out_scores)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = kd + ks;
    uint32_t P = 2;
    while ((int)P < n) P <<= 1;
    int64_t *sid = reinterpret_cast<int64_t *>(rsm);         // [P] ids (dense then sparse)
    uint32_t *skey = reinterpret_cast<uint32_t *>(sid + P);  // [P] sortable(-score): ascending == score desc
    for (uint32_t i = tid; i < P; i += 256) {
        int64_t id = -1;
        if ((int)i < kd) id = dense[q * kd + i];
        else if ((int)i < n) id = sparse[q * ks + (i - kd)];
        sid[i] = id;
    }
    __syncthreads();
    for (uint32_t i = tid; i < P; i += 256) {
        const int64_t id = sid[i];
        uint32_t key = 0xffffffffu; // padding / consumed entries sort last
        if (id >= 0) {
            if ((int)i < kd) {
                double sc = 1.0 / (double)(k + (int)i + 1);
                for (int j = 0; j < ks; j++)
                    if (sid[kd + j] == id) { sc += 1.0 / (double)(k + j + 1); break; }
                key = f32_sortable(-(float)sc);
            } else {
                bool in_dense = false;
                for (int j = 0; j < kd; j++)
                    if (sid[j] == id) { in_dense = true; break; }
                if (!in_dense) key = f32_sortable(-(float)(1.0 / (double)(k + ((int)i - kd) + 1)));
            }
        }
        skey[i] = key;
    }
    __syncthreads();
    // bitonic sort by (key asc, id asc as unsigned: -1 last)
    for (uint32_t kk = 2; kk <= P; kk <<= 1)
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (P >> 1); t += 256) {
                const uint32_t i = 2 * t - (t & (j - 1)), l = i + j;
                const uint32_t ka = skey[i], kb = skey[l];
                const uint64_t ia = (uint64_t)sid[i], ib = (uint64_t)sid[l];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == ((i & kk) == 0)) { skey[i] = kb; skey[l] = ka; sid[i] = (int64_t)ib; sid[l] = (int64_t)ia; }
            }
            __syncthreads();
        }
    for (int r = tid; r < limit; r += 256) {
        const bool ok = (uint32_t)r < P && skey[r] != 0xffffffffu;
        out_ids[q * limit + r] = ok ? sid[r] : -1;
        out_scores[q * limit + r] = ok ? -sortable_f32(skey[r]) : 0.f;
    }
}

void launch_rrf(int64_t nq, int kd, const int64_t *dense, int ks, const int64_t *sparse, int k, int limit,
                int64_t *out_ids, float *out_scores, hipStream_t s)
{
    if (nq <= 0 || limit <= 0) return;
    uint32_t P = 2;
    while ((int)P < kd + ks) P <<= 1;
    const size_t shmem = (size_t)P * 12; // 98,304 B at kd + ks = 8192: above the 64 KB default, opt in
    if (shmem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rrf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)shmem);
    hipLaunchKernelGGL(rrf_kernel, dim3((unsigned)nq), dim3(256), shmem, s, nq, kd, dense, ks, sparse, k, limit,
                       out_ids, out_scores);
}

} // namespace lb
