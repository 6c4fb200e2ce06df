// kernels_pq2.hip -- PQ codec and the two-stage ADC search on gfx950.
//
// pq_encode:  pq.(*PQEncoder).Encode, K > 16 branch (internal/pq/encoder.go:76-136) =
//             simd.FindNearestCentroid, K > 8 branch (internal/simd/simd.go:305-326): per subspace the
//             batch-flat Euclidean distances to the K centroids (euclideanUnrolled4x: four f32
//             accumulators, float32(sqrt(float64(sum))), simd.go:365-396), then the FIRST strict minimum.
// pq_decode:  pq.(*PQEncoder).Decode (encoder.go:139-158): concatenated centroids.
// adc prefilter + exact pass (search only; simd.ADCDistanceBatch itself stays on the exact kernels of
// kernels_pq.hip): the f32 table gather of adc_scan_dma_kernel is bound by LDS bank conflicts (96 random
// ds_read_b32 per row: ~3.5-way, 744 LDS cycles per 64 rows against 547 cycles of HBM time).  Here the
// per-query table is quantised to ONE BYTE per entry with a rigorous lower bound
//     t_j[c] >= min_j + s * q_j[c],   q_j[c] = floor((t_j[c] - min_j) / s) in 0..255,  s = max_j range_j / 255
// so a row's integer sum S = sum_j q_j[c_j] bounds its real ADC sum from below: sum >= base + s*S.  The
// 24 KB byte table puts a subtable's 256 entries into 64 dwords -- two per LDS bank -- so a random gather
// is at most 2-way conflicted, and integer adds need no ordering.  Rows whose bound cannot beat the
// admission threshold are dropped; the few thousand survivors are scored by the exact f32 sum in j order
// (simd.go:345-355) and filtered by the exact test, so the reported ids and distances are those of the
// exact kernel, bit for bit.  A threshold that admits too many rows overflows the candidate buffer, which
// is detected and the query is redone on the exact path.
#include "lb_device.h"

#pragma clang fp contract(off)

namespace lb {

// ---------------------------------------------------------------------------
// Encode
// ---------------------------------------------------------------------------
constexpr int ENC_THREADS = 256;

// One workgroup = 256 vectors x one subspace; the subspace's codebook (K*SUB floats) sits in LDS and is
// read as wave-wide broadcasts; every lane keeps its own sub-vector in registers.
// The reference compares sqrt'd f32 distances with strict '<' (first wins).  sqrt is monotone, so a
// centroid whose f32 sum is not below the best sum so far cannot win; only the rare improving sums
// (~ln K per subspace) pay for the f64 sqrt, and the comparison itself is on the rounded sqrt values,
// exactly as results[i] < bestDist does.
template <int SUB>
__global__ __launch_bounds__(ENC_THREADS) void pq_encode_kernel(const float *codebooks, int M, int K, const float *X,
                                                                int64_t n, int D, uint8_t *codes)
{
    extern __shared__ __attribute__((aligned(16))) float cb[];
    const int m = blockIdx.x % M;
    const int64_t rb = blockIdx.x / M;
    const float *src = codebooks + (int64_t)m * K * SUB;
    for (int i = threadIdx.x; i < K * SUB; i += ENC_THREADS) cb[i] = src[i];
    __syncthreads();
    const int64_t row = rb * ENC_THREADS + threadIdx.x;
    if (row >= n) return;
    float v[SUB];
    const float *x = X + row * (int64_t)D + (int64_t)m * SUB;
#pragma unroll
    for (int t = 0; t < SUB; t++) v[t] = x[t];
    float best_s = 0.f, best_r = 0.f;
    int best = 0;
    for (int k = 0; k < K; k++) {
        const float *c = cb + k * SUB;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        constexpr int MAIN = SUB & ~3;
#pragma unroll
        for (int t = 0; t < MAIN; t += 4) {
            const float d0 = v[t] - c[t], d1 = v[t + 1] - c[t + 1], d2 = v[t + 2] - c[t + 2], d3 = v[t + 3] - c[t + 3];
            s0 = s0 + d0 * d0;
            s1 = s1 + d1 * d1;
            s2 = s2 + d2 * d2;
            s3 = s3 + d3 * d3;
        }
#pragma unroll
        for (int t = MAIN; t < SUB; t++) {
            const float d = v[t] - c[t];
            s0 = s0 + d * d;
        }
        float s = s0 + s1;
        s = s + s2;
        s = s + s3;
        if (k == 0) {
            best_s = s;
            best_r = (float)sqrt((double)s);
        } else if (s < best_s) { // (false for NaN sums, like results[i] < bestDist)
            const float r = (float)sqrt((double)s);
            if (r < best_r) {
                best_r = r;
                best_s = s;
                best = k;
            }
        }
    }
    codes[row * (int64_t)M + m] = (uint8_t)best;
}

// any SubDim: operands straight from global memory (correctness path)
__global__ __launch_bounds__(ENC_THREADS) void pq_encode_generic_kernel(const float *codebooks, int M, int K, int sub,
                                                                        const float *X, int64_t n, int D, uint8_t *codes)
{
    const int m = blockIdx.x % M;
    const int64_t row = (int64_t)(blockIdx.x / M) * ENC_THREADS + threadIdx.x;
    if (row >= n) return;
    const float *x = X + row * (int64_t)D + (int64_t)m * sub;
    const float *cbm = codebooks + (int64_t)m * K * sub;
    const int main4 = sub & ~3;
    float best_r = 0.f;
    int best = 0;
    for (int k = 0; k < K; k++) {
        const float *c = cbm + (int64_t)k * sub;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int t = 0; t < main4; t += 4) {
            const float d0 = x[t] - c[t], d1 = x[t + 1] - c[t + 1], d2 = x[t + 2] - c[t + 2], d3 = x[t + 3] - c[t + 3];
            s0 = s0 + d0 * d0;
            s1 = s1 + d1 * d1;
            s2 = s2 + d2 * d2;
            s3 = s3 + d3 * d3;
        }
        for (int t = main4; t < sub; t++) {
            const float d = x[t] - c[t];
            s0 = s0 + d * d;
        }
        float s = s0 + s1;
        s = s + s2;
        s = s + s3;
        const float r = (float)sqrt((double)s);
        if (k == 0) {
            best_r = r;
        } else if (r < best_r) {
            best_r = r;
            best = k;
        }
    }
    codes[row * (int64_t)M + m] = (uint8_t)best;
}

void launch_pq_encode(const float *codebooks, int M, int K, int sub, const float *X, int64_t n, uint8_t *codes,
                      hipStream_t s)
{
    if (n <= 0) return;
    const int D = M * sub;
    // (grid.x carries M x row-blocks with the subspace index fastest, so the M workgroups that read the
    // same 256 rows run back to back and share the rows' cache lines in L2)
    const int64_t row_blocks_total = (n + ENC_THREADS - 1) / ENC_THREADS;
    const int64_t max_rb = 0x7fffffffll / M; // grid.x limit: launch in slices
    for (int64_t rb0 = 0; rb0 < row_blocks_total; rb0 += max_rb) {
        const int64_t nrb = row_blocks_total - rb0 < max_rb ? row_blocks_total - rb0 : max_rb;
        const int64_t r0 = rb0 * ENC_THREADS;
        const int64_t cnt = (n - r0) < nrb * ENC_THREADS ? (n - r0) : nrb * ENC_THREADS;
        const float *Xs = X + r0 * (int64_t)D;
        uint8_t *cs = codes + r0 * (int64_t)M;
        const dim3 grid((unsigned)(nrb * M)), block(ENC_THREADS);
        const size_t shmem = (size_t)K * sub * sizeof(float);
#define LB_ENC(S)                                                                                                   \
    do {                                                                                                            \
        if (shmem > 64 * 1024)                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pq_encode_kernel<S>),                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                     \
        hipLaunchKernelGGL((pq_encode_kernel<S>), grid, block, shmem, s, codebooks, M, K, Xs, cnt, D, cs);         \
    } while (0)
        switch (sub) {
        case 1: LB_ENC(1); break;
        case 2: LB_ENC(2); break;
        case 4: LB_ENC(4); break;
        case 8: LB_ENC(8); break;
        case 12: LB_ENC(12); break;
        case 16: LB_ENC(16); break;
        case 32: LB_ENC(32); break;
        default:
            hipLaunchKernelGGL(pq_encode_generic_kernel, grid, block, 0, s, codebooks, M, K, sub, Xs, cnt, D, cs);
            break;
        }
#undef LB_ENC
    }
}

// ---------------------------------------------------------------------------
// Decode: out[row][m*sub + t] = codebooks[m][codes[row][m]][t]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_decode_kernel(const float *codebooks, int M, int K, int sub, const uint8_t *codes,
                                                        int64_t n, float *out)
{
    const int64_t total = n * (int64_t)M * sub;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rm = i / sub;
        const int t = (int)(i - rm * sub);
        const int m = (int)(rm % M);
        out[i] = codebooks[((int64_t)m * K + codes[rm]) * sub + t];
    }
}

void launch_pq_decode(const float *codebooks, int M, int K, int sub, const uint8_t *codes, int64_t n, float *out,
                      hipStream_t s)
{
    if (n <= 0) return;
    int64_t blocks = (n * (int64_t)M * sub + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pq_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, s, codebooks, M, K, sub, codes, n, out);
}

// ---------------------------------------------------------------------------
// Two-stage ADC search
// ---------------------------------------------------------------------------
// One workgroup per query slot: quantise the query's f32 table to the byte table and derive the integer
// admission bound from the slot's threshold.  params[slot] = {s_tau (int), ok, -, -}; ok = 0 when the
// prefilter cannot be used for this query (non-finite table entries, no threshold yet): the caller then
// runs the exact kernel.
//   U    = nextafter(tau_dist)^2                  (every admitted row has f32 sum <= U)
//   real sum <= f32 sum * (1 + 2*gamma), gamma = 1.05 * M * 2^-24   (sequential f32 sum of M terms >= 0)
//   real sum >= base + s * S - eps                 (floor quantisation; eps covers the f64 roundings)
//   => admitted rows satisfy S <= (U*(1+2 gamma) - base) / s + 2
__global__ __launch_bounds__(256) void adc_quantise_kernel(const float *tables, int M, const CandState cs, const int *slots,
                                                           uint8_t *qtabs, int *params)
{
    __shared__ float s_min[256], s_rng[256];
    __shared__ double s_base;
    __shared__ float s_scale;
    __shared__ int s_bad;
    const int slot = slots ? slots[blockIdx.x] : (int)blockIdx.x;
    const float *tab = tables + (size_t)slot * M * 256;
    uint8_t *qt = qtabs + (size_t)slot * M * 256;
    const int tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    // per-subtable min / max (one wave-free loop per thread; M <= 256)
    if (tid < M) {
        float mn = __builtin_huge_valf(), mx = -__builtin_huge_valf();
        int bad = 0;
        for (int c = 0; c < 256; c++) {
            const float t = tab[tid * 256 + c];
            if (!(t >= 0.f) || t > 3.0e38f) bad = 1; // NaN, negative or infinite entries: no prefilter
            mn = t < mn ? t : mn;
            mx = t > mx ? t : mx;
        }
        s_min[tid] = mn;
        s_rng[tid] = mx - mn;
        if (bad) atomicOr(&s_bad, 1);
    }
    __syncthreads();
    if (tid == 0) {
        double base = 0.0;
        float rmax = 0.f;
        for (int j = 0; j < M; j++) {
            base += (double)s_min[j];
            rmax = s_rng[j] > rmax ? s_rng[j] : rmax;
        }
        s_base = base;
        s_scale = rmax > 0.f ? rmax / 255.0f : 1.0f;
    }
    __syncthreads();
    const double sc = (double)s_scale;
    for (int i = tid; i < M * 256; i += 256) {
        const int j = i >> 8;
        const double rel = ((double)tab[i] - (double)s_min[j]) / sc;
        int q = (int)floor(rel);
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        qt[i] = (uint8_t)q;
    }
    if (tid == 0) {
        int ok = s_bad ? 0 : 1;
        int s_tau = 0;
        const uint64_t tau = cs.tau[slot];
        if (tau == kEntryMax) {
            ok = 0; // no threshold: every row would be admitted
        } else {
            const float td = entry_key(tau);
            if (!(td >= 0.f) || td > 1.0e18f) {
                ok = 0;
            } else {
                const float tn = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, td + 0.0f) + 1u); // next f32 above td (td >= 0, finite)
                const double U = (double)tn * (double)tn;
                const double gamma = 1.05 * (double)M * 5.9604644775390625e-8;
                const double lim = (U * (1.0 + 2.0 * gamma) - s_base) / sc + 2.0;
                if (lim < 0.0) s_tau = -1;                 // nothing can pass
                else if (lim > 2.0e9) ok = 0;
                else s_tau = (int)lim;
            }
        }
        params[slot * 4 + 0] = s_tau;
        params[slot * 4 + 1] = ok;
    }
}

void launch_adc_quantise(const float *tables, int M, CandState cs, const int *slots, int nslots, uint8_t *qtabs,
                         int *params, hipStream_t s)
{
    if (nslots <= 0) return;
    hipLaunchKernelGGL(adc_quantise_kernel, dim3(nslots), dim3(256), 0, s, tables, M, cs, slots, qtabs, params);
}

struct AdcPreArgs {
    const uint8_t *qtab; // [M*256] byte table of this query
    const int *params;   // {s_tau, ok}
    const uint8_t *codes;
    int64_t n;
    uint32_t *cand;      // candidate rows
    uint32_t cand_cap;
    uint32_t *cand_cnt;  // [0] = count (may exceed cand_cap -> overflow)
};

// Same streaming structure as adc_scan_dma_kernel: each wave DMAs its 64 rows (64*M contiguous bytes) into
// a private LDS slot with M/16 direct-to-LDS loads of 1 KiB, reads its own row back, re-issues the DMA for
// its next tile and then does the M byte gathers.  The byte table is 24 KB at M = 96, which leaves room
// for 16 waves per CU (96 KB of code bytes in flight) instead of 8.
constexpr int PRE_WAVES = 16;

template <int MCH>
__global__ __launch_bounds__(PRE_WAVES * 64) void adc_prefilter_kernel(AdcPreArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_u8[];
    constexpr int M = MCH * 16;
    unsigned char *tab = smem_u8;
    unsigned char *stage_all = smem_u8 + M * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.params[1] == 0) return; // prefilter unusable for this query (uniform): the exact path runs instead
    const int s_tau = a.params[0];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.qtab);
        uint4 *dst = reinterpret_cast<uint4 *>(tab);
        for (int i = tid; i < M * 16; i += PRE_WAVES * 64) dst[i] = src[i];
    }
    __syncthreads();
    unsigned char *stage = stage_all + wave * (64 * M);
    const int64_t ntiles = (a.n + 63) / 64;
    const int64_t tstride = (int64_t)gridDim.x * PRE_WAVES;
    const unsigned char *last16 = a.codes + a.n * (int64_t)M - 16;

    auto issue = [&](int64_t tile) {
        const unsigned char *src0 = a.codes + tile * 64 * (int64_t)M + lane * 16;
#pragma unroll
        for (int i = 0; i < MCH; i++) {
            const unsigned char *src = src0 + i * 1024;
            if (src > last16) src = last16; // tail tile: stay inside the buffer (rows past the end are discarded)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(stage + i * 1024), 16, 0,
                                             2 /* nt: the codes stream through once */);
        }
    };

    int64_t tile = (int64_t)blockIdx.x * PRE_WAVES + wave;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += tstride) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 c[MCH];
#pragma unroll
        for (int i = 0; i < MCH; i++) c[i] = *reinterpret_cast<const uint4 *>(stage + lane * M + i * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // slot fully read before it is refilled
        if (tile + tstride < ntiles) issue(tile + tstride);

        uint32_t S = 0;
#pragma unroll
        for (int g = 0; g < MCH; g++) {
            const uint32_t w[4] = {c[g].x, c[g].y, c[g].z, c[g].w};
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int b = 0; b < 4; b++)
                    S += tab[(g * 16 + t * 4 + b) * 256 + ((w[t] >> (8 * b)) & 0xffu)];
        }
        const int64_t row = tile * 64 + lane;
        if (row < a.n && (int)S <= s_tau) {
            const uint32_t pos = atomicAdd(a.cand_cnt, 1u);
            if (pos < a.cand_cap) a.cand[pos] = (uint32_t)row;
        }
    }
}

// any M (or misaligned codes): lane = row straight from global memory
__global__ __launch_bounds__(1024) void adc_prefilter_generic_kernel(AdcPreArgs a, int M)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_u8[];
    if (a.params[1] == 0) return;
    const int s_tau = a.params[0];
    for (int i = threadIdx.x; i < M * 256; i += 1024) smem_u8[i] = a.qtab[i];
    __syncthreads();
    for (int64_t row = (int64_t)blockIdx.x * 1024 + threadIdx.x; row < a.n; row += (int64_t)gridDim.x * 1024) {
        const uint8_t *c = a.codes + row * (int64_t)M;
        uint32_t S = 0;
        for (int j = 0; j < M; j++) S += smem_u8[j * 256 + c[j]];
        if ((int)S <= s_tau) {
            const uint32_t pos = atomicAdd(a.cand_cnt, 1u);
            if (pos < a.cand_cap) a.cand[pos] = (uint32_t)row;
        }
    }
}

bool launch_adc_prefilter(const uint8_t *qtab, const int *params, int M, const uint8_t *codes, int64_t n,
                          uint32_t *cand, uint32_t cand_cap, uint32_t *cand_cnt, hipStream_t s)
{
    if (n <= 0) return true;
    AdcPreArgs a{qtab, params, codes, n, cand, cand_cap, cand_cnt};
    const bool vec = (M % 16 == 0) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(qtab) & 15) == 0);
    const int64_t ntiles = (n + 63) / 64;
    int64_t blocks = (ntiles + PRE_WAVES - 1) / PRE_WAVES;
    if (blocks > 256) blocks = 256;
#define LB_PRE(MCH)                                                                                                 \
    do {                                                                                                            \
        const size_t shmem = (size_t)(MCH * 16) * 256 + (size_t)PRE_WAVES * 64 * (MCH * 16);                        \
        if (shmem > 160 * 1024) break;                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_prefilter_kernel<MCH>),                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                         \
        hipLaunchKernelGGL((adc_prefilter_kernel<MCH>), dim3((unsigned)blocks), dim3(PRE_WAVES * 64), shmem, s, a); \
        return true;                                                                                                \
    } while (0)
    if (vec && n >= 4096) {
        switch (M / 16) {
        case 1: LB_PRE(1); break;
        case 2: LB_PRE(2); break;
        case 3: LB_PRE(3); break;
        case 4: LB_PRE(4); break;
        case 6: LB_PRE(6); break;
        case 8: LB_PRE(8); break;
        default: break;
        }
    }
#undef LB_PRE
    const size_t shmem = (size_t)M * 256;
    if (shmem > 160 * 1024) return false;
    if (shmem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_prefilter_generic_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    int64_t gb = (n + 1023) / 1024;
    if (gb > 1024) gb = 1024;
    hipLaunchKernelGGL(adc_prefilter_generic_kernel, dim3((unsigned)gb), dim3(1024), shmem, s, a, M);
    return true;
}

// Exact ADC distance of the surviving rows (f32 sum in j order, sqrt in f64), admitted into the slot's
// list by the exact test `entry < tau` -- the same entries the exact full pass would have admitted.
// More candidates than the buffer holds -> flag bit 0 (the query is redone on the exact path).
// rows == nullptr: not used.  Also serves lb_gpu_pq_rerank (all_out != nullptr: plain distances + scores).
struct AdcExactArgs {
    const float *table;
    int M;
    const uint8_t *codes;
    const uint32_t *cand;
    const uint32_t *cand_cnt;
    uint32_t cand_cap;
    const int *params; // {s_tau, ok}: ok == 0 -> the prefilter did not run, nothing to do
    int slot;
    CandState cs;
};

__global__ __launch_bounds__(256) void adc_exact_candidates_kernel(AdcExactArgs a)
{
    if (a.params[1] == 0) return;
    const uint32_t raw = *a.cand_cnt;
    if (raw > a.cand_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&a.cs.flags[a.slot], 1u);
        return;
    }
    const uint64_t tau = a.cs.tau[a.slot];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < raw; i += gridDim.x * 256u) {
        const uint32_t row = a.cand[i];
        const uint8_t *c = a.codes + (int64_t)row * a.M;
        float sum = 0.f;
        for (int j = 0; j < a.M; j++) sum = sum + a.table[j * 256 + c[j]];
        const uint64_t ent = pack_entry((float)sqrt((double)sum), row);
        if (ent < tau) {
            const uint32_t pos = atomicAdd(&a.cs.cnt[a.slot], 1u);
            if (pos < a.cs.cap) a.cs.lists[(size_t)a.slot * a.cs.cap + pos] = ent;
        }
    }
}

void launch_adc_exact_candidates(const float *table, int M, const uint8_t *codes, const uint32_t *cand,
                                 const uint32_t *cand_cnt, uint32_t cand_cap, const int *params, int slot, CandState cs,
                                 hipStream_t s)
{
    AdcExactArgs a{table, M, codes, cand, cand_cnt, cand_cap, params, slot, cs};
    hipLaunchKernelGGL(adc_exact_candidates_kernel, dim3(64), dim3(256), 0, s, a);
}

// processChunkInternal's PQ branch (internal/store/parallel_search.go:292-345): ADC distance of the
// given candidate rows (gathered from the resident codes) + Score = 1/(1+d) (:355-362).
// Rows outside [0, n) report FLT_MAX / 0.
__global__ __launch_bounds__(256) void adc_rerank_kernel(const float *table, int M, const uint8_t *codes, int64_t n,
                                                         const int64_t *rows, int64_t nrows, float *out_dist,
                                                         float *out_score)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrows) return;
    const int64_t row = rows[i];
    if (row < 0 || row >= n) {
        out_dist[i] = 3.402823466e+38f;
        if (out_score) out_score[i] = 0.f;
        return;
    }
    const uint8_t *c = codes + row * (int64_t)M;
    float sum = 0.f;
    for (int j = 0; j < M; j++) sum = sum + table[j * 256 + c[j]];
    const float d = (float)sqrt((double)sum);
    out_dist[i] = d;
    if (out_score) out_score[i] = __fdiv_rn(1.0f, 1.0f + d);
}

void launch_adc_rerank(const float *table, int M, const uint8_t *codes, int64_t n, const int64_t *rows, int64_t nrows,
                       float *out_dist, float *out_score, hipStream_t s)
{
    if (nrows <= 0) return;
    hipLaunchKernelGGL(adc_rerank_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s, table, M, codes, n, rows,
                       nrows, out_dist, out_score);
}

} // namespace lb
