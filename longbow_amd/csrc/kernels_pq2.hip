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

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
// Quantise one query's f32 table to the byte table (one workgroup per subtable) and derive the integer
// admission bound from the slot's threshold.  minrng[j] = {min_j, max_j - min_j, bad} comes from the table
// builder (build_adc_table_kernel reduces each subtable as it writes it).  params = {s_tau, ok}; ok = 0 when
// the prefilter cannot be used for this query (non-finite table entries, no threshold yet): the caller's
// exact kernel then runs instead.
//   U    = nextafter(tau_dist)^2                  (every admitted row has f32 sum <= U)
//   real sum <= f32 sum * (1 + 2*gamma), gamma = 1.05 * M * 2^-24   (sequential f32 sum of M terms >= 0)
//   real sum >= base + s * S - eps                 (floor quantisation; eps covers the roundings)
//   => admitted rows satisfy S <= (U*(1+2 gamma) - base) / s + 2
__global__ __launch_bounds__(256) void adc_quantise_kernel(const float *table, const float *minrng, int M, const uint64_t *tau_p,
                                                           uint8_t *qt, int *params)
{
    __shared__ float s_inv, s_scale, s_mn;
    __shared__ double s_base;
    __shared__ int s_bad;
    const int j = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) { // one wave: max range, sum of minima, any bad entry
        float rmax = 0.f;
        double base = 0.0;
        int bad = 0;
        for (int i = tid; i < M; i += 64) {
            rmax = fmaxf(rmax, minrng[i * 4 + 1]);
            base += (double)minrng[i * 4 + 0];
            bad |= minrng[i * 4 + 2] != 0.f ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rmax = fmaxf(rmax, __shfl_xor(rmax, off));
            base += __shfl_xor(base, off);
            bad |= __shfl_xor(bad, off);
        }
        if (tid == 0) {
            s_scale = rmax > 0.f ? rmax / 255.0f : 1.0f;
            s_inv = rmax > 0.f ? 255.0f / rmax : 0.f;
            s_base = base;
            s_bad = bad;
            s_mn = minrng[j * 4 + 0];
        }
    }
    __syncthreads();
    // q = floor((t - min) / s) in f32: the roundings can lift q by one only when (t - min)/s is within
    // ~5e-5 of an integer, i.e. s*q exceeds t - min by < 5e-5 s per entry, < 0.01 s over M <= 155 entries:
    // covered by the +2 units of slack in s_tau below.
    {
        const float r = (table[j * 256 + tid] - s_mn) * s_inv;
        int qi = (int)floorf(r);
        qi = qi < 0 ? 0 : (qi > 255 ? 255 : qi);
        qt[j * 256 + tid] = (uint8_t)qi;
    }
    if (j == 0 && tid == 0) {
        int ok = s_bad ? 0 : 1;
        int s_tau = 0;
        const uint64_t tau = *tau_p;
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
                const double lim = (U * (1.0 + 2.0 * gamma) - s_base) / (double)s_scale + 2.0;
                if (lim < 0.0) s_tau = -1;                 // nothing can pass
                else if (lim > 2.0e9) ok = 0;
                else s_tau = (int)lim;
            }
        }
        params[0] = s_tau;
        params[1] = ok;
    }
}

void launch_adc_quantise(const float *table, const float *minrng, int M, const uint64_t *tau, uint8_t *qtab, int *params,
                         hipStream_t s)
{
    hipLaunchKernelGGL(adc_quantise_kernel, dim3(M), dim3(256), 0, s, table, minrng, M, tau, qtab, params);
}

struct AdcPreArgs {
    const uint8_t *qtab; // [M*256] byte table of this query
    const int *params;   // {s_tau, ok}
    const uint8_t *codes;
    int64_t n;
    uint32_t *cand;      // candidate rows
    uint32_t cand_cap;
    uint32_t *cand_cnt;  // [0] = count (may exceed cand_cap -> overflow)
    // second query of a two-query pass (NQ == 2): the code bytes are streamed ONCE for both
    const uint8_t *qtab2 = nullptr;
    const int *params2 = nullptr;
    uint32_t *cand2 = nullptr;
    uint32_t *cand_cnt2 = nullptr;
};

// Same streaming structure as adc_scan_dma_kernel: each wave DMAs its 64 rows (64*M contiguous bytes) into
// a private LDS slot with M/16 direct-to-LDS loads of 1 KiB, reads its own row back, re-issues the DMA for
// its next tile and then does the M byte gathers.  The byte table is 24 KB at M = 96, which leaves room
// for 16 waves per CU (96 KB of code bytes in flight) instead of 8.
// NQ == 2: two queries per pass over the codes.  The pass is bound by the LDS gathers (2 per code byte then) and the code
// stream together, not by the stream alone, so two queries cost ~1.3x one pass instead of 2x: the second table sits
// M * 256 bytes behind the first and its gather uses the same address register with a larger instruction offset.
template <int MCH, bool QW, int WAVES, int SLOTS, bool HYB = false, int NQ = 1>
__global__ __launch_bounds__(WAVES * 64) void adc_prefilter_kernel(AdcPreArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_u8[];
    constexpr int M = MCH * 16;
    unsigned char *tab = smem_u8;
    unsigned char *stage_all = smem_u8 + NQ * M * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // prefilter unusable for a query (uniform; decided by adc_quantise_kernel): nothing of it is admitted here, the select
    // then finds fewer than k entries, flags the query, and the host redoes it on the exact schedule
    const int ok1 = a.params[1], ok2 = NQ == 2 ? a.params2[1] : 0;
    if (ok1 == 0 && ok2 == 0) return;
    const int s_tau = ok1 ? a.params[0] : -1;
    const int s_tau2 = (NQ == 2 && ok2) ? a.params2[0] : -1;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.qtab);
        uint4 *dst = reinterpret_cast<uint4 *>(tab);
        for (int i = tid; i < M * 16; i += WAVES * 64) dst[i] = src[i];
        if (NQ == 2) {
            const uint4 *src2 = reinterpret_cast<const uint4 *>(a.qtab2);
            for (int i = tid; i < M * 16; i += WAVES * 64) dst[M * 16 + i] = src2[i];
        }
    }
    __syncthreads();
    unsigned char *stage = stage_all + wave * (SLOTS * 64 * M);
    const int64_t ntiles = (a.n + 63) / 64;
    const int64_t tstride = (int64_t)gridDim.x * WAVES;
    const unsigned char *last16 = a.codes + a.n * (int64_t)M - 16;

    auto issue = [&](int64_t tile, int slot) {
        const unsigned char *src0 = a.codes + tile * 64 * (int64_t)M + lane * 16;
#pragma unroll
        for (int i = 0; i < MCH; i++) {
            const unsigned char *src = src0 + i * 1024;
            if (src > last16) src = last16; // tail tile: stay inside the buffer (rows past the end are discarded)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(stage + slot * (64 * M) + i * 1024), 16,
                                             0, 2 /* nt: the codes stream through once */);
        }
    };

    int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    if (tile < ntiles) issue(tile, 0);
    if (SLOTS == 2 && tile + tstride < ntiles) issue(tile + tstride, 1);
    int slot = 0;
    for (; tile < ntiles; tile += tstride, slot = (SLOTS == 2) ? (slot ^ 1) : 0) {
        // two slots: the older of the (up to) two tiles in flight has landed once at most MCH DMAs remain
        if (SLOTS == 2 && tile + tstride < ntiles) {
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
        if (tile + SLOTS * tstride < ntiles) issue(tile + SLOTS * tstride, slot);

        // Gather.  A byte gather (ds_read_u8) spreads a subtable's 64 dwords over the 32 banks of the 4-byte
        // LDS path, two per bank: random codes collide 2-way on every instruction (measured:
        // SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE, 4.2 LDS cycles per instruction).  The QW variant
        // reads the qword holding the entry instead (ds_read_b64 is banked 64 wide: the subtable's 32 qwords
        // own one bank pair each and lanes that meet on a pair read the SAME qword, a broadcast) and picks
        // the byte with v_perm_b32: conflict-free (2.35 cycles per instruction) but 3.3x the VALU
        // instructions, which costs more than the conflicts did (2.03 vs 1.78 ms at 100M x 96); HYB does that
        // for every third sub-quantiser only (1.78 vs 1.80 ms: inside the noise, not the default).  Also
        // tried and slower: two DMA slots per wave (11 x 2: 1.87 ms), 16 gathers issued ahead of their adds
        // (1.91 ms), a third of the gathers through the vector cache instead of LDS (1.87-2.7 ms).
        uint32_t S = 0, S2 = 0;
#pragma unroll
        for (int g = 0; g < MCH; g++) {
            const uint32_t w[4] = {c[g].x, c[g].y, c[g].z, c[g].w};
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int j = g * 16 + t * 4 + b;
                    const uint32_t code = (w[t] >> (8 * b)) & 0xffu;
                    if (QW || (HYB && (j % 3) == 0)) {
                        const uint2 qw = *reinterpret_cast<const uint2 *>(tab + j * 256 + (code & 0xf8u));
                        S += __builtin_amdgcn_perm(qw.y, qw.x, code & 7u) & 0xffu;
                    } else {
                        S += tab[j * 256 + code];
                        if (NQ == 2) S2 += tab[M * 256 + j * 256 + code];
                    }
                }
        }
        const int64_t row = tile * 64 + lane;
        if (row < a.n && (int)S <= s_tau) {
            const uint32_t pos = atomicAdd(a.cand_cnt, 1u);
            if (pos < a.cand_cap) a.cand[pos] = (uint32_t)row;
        }
        if (NQ == 2 && row < a.n && (int)S2 <= s_tau2) {
            const uint32_t pos = atomicAdd(a.cand_cnt2, 1u);
            if (pos < a.cand_cap) a.cand2[pos] = (uint32_t)row;
        }
    }
}

// FOUR queries per pass over the codes (round 4).  The pass is bound by the LDS gather rate (2.3 ms per PAIR of queries at
// 100M x 96 against a 1.5 ms code stream), so one gather should serve more queries: the four byte tables are interleaved as
// tab4[j][code] = {b_q0, b_q1, b_q2, b_q3} -- one dword -- and ONE ds_read_b32 per code byte returns all four queries' entries;
// v_dot4_u32_u8 with a one-hot selector adds each query's byte to its sum (4 VALU per gather, exact integer sums as before).
// A subtable is 256 dwords = 4 per bank: random codes collide ~4-way, but that is one instruction where the two-query form
// issues four.  LDS: M KiB of tables + WAVES x 64 M of staging (M = 96: 96 + 60 KB with ten waves).
struct AdcPre4Args {
    const uint8_t *qtab[4];
    const int *params[4];
    uint32_t *cand[4];
    uint32_t *cand_cnt[4];
    const uint8_t *codes;
    int64_t n;
    uint32_t cand_cap;
};

template <int MCH, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void adc_prefilter4_kernel(AdcPre4Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_u8[];
    constexpr int M = MCH * 16;
    uint32_t *tab4 = reinterpret_cast<uint32_t *>(smem_u8);            // [M][256] dwords
    unsigned char *stage_all = smem_u8 + (size_t)M * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int s_tau[4];
    bool any = false;
#pragma unroll
    for (int q = 0; q < 4; q++) { // (a query whose prefilter is unusable admits nothing: its select flags it, the host redoes it)
        const int ok = a.params[q][1];
        s_tau[q] = ok ? a.params[q][0] : -1;
        any = any || ok != 0;
    }
    if (!any) return;
    for (int i = tid; i < M * 64; i += WAVES * 64) { // dword i of every table = the entries of codes 4 i .. 4 i + 3
        uint32_t d[4];
#pragma unroll
        for (int q = 0; q < 4; q++) d[q] = reinterpret_cast<const uint32_t *>(a.qtab[q])[i];
#pragma unroll
        for (int c = 0; c < 4; c++)
            tab4[4 * i + c] = ((d[0] >> (8 * c)) & 0xffu) | (((d[1] >> (8 * c)) & 0xffu) << 8) | (((d[2] >> (8 * c)) & 0xffu) << 16) |
                              (((d[3] >> (8 * c)) & 0xffu) << 24);
    }
    __syncthreads();
    unsigned char *stage = stage_all + wave * (64 * M);
    const int64_t ntiles = (a.n + 63) / 64;
    const int64_t tstride = (int64_t)gridDim.x * WAVES;
    const unsigned char *last16 = a.codes + a.n * (int64_t)M - 16;
    auto issue = [&](int64_t tile) {
        const unsigned char *src0 = a.codes + tile * 64 * (int64_t)M + lane * 16;
#pragma unroll
        for (int i = 0; i < MCH; i++) {
            const unsigned char *src = src0 + i * 1024;
            if (src > last16) src = last16; // tail tile: stay inside the buffer (rows past the end are discarded)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(stage + i * 1024), 16, 0, 2 /* nt */);
        }
    };
    int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += tstride) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 c[MCH];
#pragma unroll
        for (int i = 0; i < MCH; i++) c[i] = *reinterpret_cast<const uint4 *>(stage + lane * M + i * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // slot fully read before it is refilled
        if (tile + tstride < ntiles) issue(tile + tstride);
        uint32_t S[4] = {0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < MCH; g++) {
            const uint32_t w[4] = {c[g].x, c[g].y, c[g].z, c[g].w};
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int j = g * 16 + t * 4 + b;
                    const uint32_t code = (w[t] >> (8 * b)) & 0xffu;
                    const uint32_t v = tab4[j * 256 + code];
                    S[0] = __builtin_amdgcn_udot4(v, 0x00000001u, S[0], false);
                    S[1] = __builtin_amdgcn_udot4(v, 0x00000100u, S[1], false);
                    S[2] = __builtin_amdgcn_udot4(v, 0x00010000u, S[2], false);
                    S[3] = __builtin_amdgcn_udot4(v, 0x01000000u, S[3], false);
                }
        }
        const int64_t row = tile * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (row < a.n && (int)S[q] <= s_tau[q]) {
                const uint32_t pos = atomicAdd(a.cand_cnt[q], 1u);
                if (pos < a.cand_cap) a.cand[q][pos] = (uint32_t)row;
            }
    }
}

template <int MCH, int WAVES>
static bool try_prefilter4(const AdcPre4Args &a, hipStream_t s)
{
    const size_t shmem = (size_t)(MCH * 16) * 1024 + (size_t)WAVES * 64 * (MCH * 16);
    if (shmem > 160 * 1024) return false;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_prefilter4_kernel<MCH, WAVES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    const int64_t ntiles = (a.n + 63) / 64;
    int64_t blocks = (ntiles + WAVES - 1) / WAVES;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL((adc_prefilter4_kernel<MCH, WAVES>), dim3((unsigned)blocks), dim3(WAVES * 64), shmem, s, a);
    return true;
}

// four queries in ONE pass over the codes; false = no such form for this M (the caller runs pairs)
bool launch_adc_prefilter4(const uint8_t *const qtab[4], const int *const params[4], uint32_t *const cand[4], uint32_t *const cand_cnt[4],
                           int M, const uint8_t *codes, int64_t n, uint32_t cand_cap, hipStream_t s)
{
    if (n < 4096 || M % 16 != 0 || (reinterpret_cast<uintptr_t>(codes) & 15) != 0) return false;
    AdcPre4Args a;
    for (int q = 0; q < 4; q++) {
        if ((reinterpret_cast<uintptr_t>(qtab[q]) & 15) != 0) return false;
        a.qtab[q] = qtab[q]; a.params[q] = params[q]; a.cand[q] = cand[q]; a.cand_cnt[q] = cand_cnt[q];
    }
    a.codes = codes; a.n = n; a.cand_cap = cand_cap;
    switch (M / 16) {
    case 1: return try_prefilter4<1, 16>(a, s);
    case 2: return try_prefilter4<2, 16>(a, s);
    case 3: return try_prefilter4<3, 16>(a, s);
    case 4: return try_prefilter4<4, 16>(a, s);
    case 6: return try_prefilter4<6, 10>(a, s);
    default: return false; // (M = 128: 128 KB of tables leave room for four waves' staging only)
    }
}

// The same pass with the codes staged in REGISTERS, not LDS.  A wave's tile is 64 rows = 64 * M contiguous bytes; request i
// of the wave loads bytes [1024 i, 1024 i + 1024) of it with one coalesced global_load_dwordx4 (nt), so lane l holds the
// 16-byte chunk c = 64 i + l: sub-quantisers 16 (c % MCH) .. + 15 of row c / MCH (a chunk never straddles a row: M = 16 MCH).
// Every lane gathers its chunk's 16 byte-table entries and leaves the partial sum in a wave-private LDS word; lane r then
// adds the MCH words of row 64 tile + r.  Against the DMA variant above this takes the 6 KB of DMA writes and the 6 KB of
// row reads per tile off the LDS (the unit that bounds the pass: 67 % array-busy + the DMA writes it also serves,
// profiles/r02_pmc_pq.txt) for 2 x 1.5 KB of partial sums, and keeps TWO tiles of loads in flight per wave (the next
// tile's registers are requested before the current one is gathered): 192 KB per CU instead of 96.
// Which sub-table a lane's chunk belongs to depends on (64 i + l) % MCH only -- the same for every tile -- so the per-lane
// table bases are computed once.
template <int MCH, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void adc_prefilter_reg_kernel(AdcPreArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_u8[];
    constexpr int M = MCH * 16;
    unsigned char *tab = smem_u8;
    uint32_t *part_all = reinterpret_cast<uint32_t *>(smem_u8 + M * 256); // [WAVES][MCH * 64] partial sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.params[1] == 0) return; // prefilter unusable for this query (uniform): the exact path runs instead
    const int s_tau = a.params[0];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.qtab);
        uint4 *dst = reinterpret_cast<uint4 *>(tab);
        for (int i = tid; i < M * 16; i += WAVES * 64) dst[i] = src[i];
    }
    __syncthreads();
    uint32_t *part = part_all + wave * (MCH * 64);
    const int64_t ntiles = (a.n + 63) / 64;
    const int64_t tstride = (int64_t)gridDim.x * WAVES;
    const unsigned char *last16 = a.codes + a.n * (int64_t)M - 16;
    uint32_t base[MCH]; // byte offset of the first of this lane's 16 sub-tables, per request
#pragma unroll
    for (int i = 0; i < MCH; i++) base[i] = (uint32_t)(((i * 64 + lane) % MCH) * 16 * 256);

    auto load = [&](int64_t tile, uint4 (&c)[MCH]) {
        const unsigned char *src0 = a.codes + tile * 64 * (int64_t)M + lane * 16;
#pragma unroll
        for (int i = 0; i < MCH; i++) {
            const unsigned char *src = src0 + i * 1024;
            if (src > last16) src = last16; // tail tile: stay inside the buffer (rows past the end are discarded)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src));
            c[i] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    uint4 cur[MCH], nxt[MCH];
    int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    if (tile < ntiles) load(tile, cur);
    for (; tile < ntiles; tile += tstride) {
        const bool more = tile + tstride < ntiles;
        if (more) load(tile + tstride, nxt);
#pragma unroll
        for (int i = 0; i < MCH; i++) {
            const uint32_t w[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
            const unsigned char *tb = tab + base[i];
            uint32_t S = 0;
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int b = 0; b < 4; b++) S += tb[(t * 4 + b) * 256 + ((w[t] >> (8 * b)) & 0xffu)];
            part[i * 64 + lane] = S;
        }
        // the words are read back by OTHER lanes of this wave: LDS operations of one wave complete in order, the compiler
        // only has to keep them in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t total = 0;
#pragma unroll
        for (int p = 0; p < MCH; p++) total += part[lane * MCH + p];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier(); // (the next tile's partial sums overwrite these words)
        const int64_t row = tile * 64 + lane;
        if (row < a.n && (int)total <= s_tau) {
            const uint32_t pos = atomicAdd(a.cand_cnt, 1u);
            if (pos < a.cand_cap) a.cand[pos] = (uint32_t)row;
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < MCH; i++) cur[i] = nxt[i];
        }
    }
}

template <int MCH, int WAVES>
static bool try_prefilter_reg(const AdcPreArgs &a, hipStream_t s, int wg_per_cu = 1)
{
    const size_t shmem = (size_t)(MCH * 16) * 256 + (size_t)WAVES * MCH * 64 * sizeof(uint32_t);
    if (shmem * wg_per_cu > 160 * 1024) return false;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_prefilter_reg_kernel<MCH, WAVES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    const int64_t ntiles = (a.n + 63) / 64;
    int64_t blocks = (ntiles + WAVES - 1) / WAVES;
    if (blocks > 256 * wg_per_cu) blocks = 256 * wg_per_cu;
    hipLaunchKernelGGL((adc_prefilter_reg_kernel<MCH, WAVES>), dim3((unsigned)blocks), dim3(WAVES * 64), shmem, s, a);
    return true;
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

template <int MCH, bool QW, int WAVES, int SLOTS, bool HYB = false, int NQ = 1>
static bool try_prefilter(const AdcPreArgs &a, hipStream_t s)
{
    const size_t shmem = (size_t)NQ * (MCH * 16) * 256 + (size_t)WAVES * SLOTS * 64 * (MCH * 16);
    if (shmem > 160 * 1024) return false;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(adc_prefilter_kernel<MCH, QW, WAVES, SLOTS, HYB, NQ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    const int64_t ntiles = (a.n + 63) / 64;
    int64_t blocks = (ntiles + WAVES - 1) / WAVES;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL((adc_prefilter_kernel<MCH, QW, WAVES, SLOTS, HYB, NQ>), dim3((unsigned)blocks), dim3(WAVES * 64), shmem, s, a);
    return true;
}

// two queries per pass (see the kernel); false = no two-query form for this M (the caller runs two single passes)
bool launch_adc_prefilter2(const uint8_t *qtab, const int *params, uint32_t *cand, uint32_t *cand_cnt, const uint8_t *qtab2,
                           const int *params2, uint32_t *cand2, uint32_t *cand_cnt2, int M, const uint8_t *codes, int64_t n,
                           uint32_t cand_cap, hipStream_t s)
{
    if (n < 4096) return false;
    AdcPreArgs a{qtab, params, codes, n, cand, cand_cap, cand_cnt, qtab2, params2, cand2, cand_cnt2};
    const bool vec = (M % 16 == 0) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(qtab) & 15) == 0) && ((reinterpret_cast<uintptr_t>(qtab2) & 15) == 0);
    if (!vec) return false;
    switch (M / 16) {
    case 1: return try_prefilter<1, false, 16, 1, false, 2>(a, s);
    case 2: return try_prefilter<2, false, 16, 1, false, 2>(a, s);
    case 3: return try_prefilter<3, false, 16, 1, false, 2>(a, s);
    case 4: return try_prefilter<4, false, 16, 1, false, 2>(a, s);
    case 6: return try_prefilter<6, false, 16, 1, false, 2>(a, s);
    default: return false; // (M = 128: two 32 KB tables + 10 slots of 8 KB do not fit beside each other usefully)
    }
}

bool launch_adc_prefilter(const uint8_t *qtab, const int *params, int M, const uint8_t *codes, int64_t n,
                          uint32_t *cand, uint32_t cand_cap, uint32_t *cand_cnt, hipStream_t s)
{
    if (n <= 0) return true;
    AdcPreArgs a{qtab, params, codes, n, cand, cand_cap, cand_cnt};
    const bool vec = (M % 16 == 0) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(qtab) & 15) == 0);
    // A/B (diagnostic build): 0 = 16 waves x 1 slot, byte gathers (default); 1 = 11 x 2; 2 = 8 x 2; 3 = 12 x 1;
    // 4 = 16 x 1 with qword gathers; 5 = 16 x 1, every third gather a qword gather
    static const int cfg = lb_tunable("LB_ADC_PRE_CFG", 0);
    if (vec && n >= 4096) {
        bool ok = false;
        switch (M / 16) {
        case 1: ok = try_prefilter<1, false, 16, 1>(a, s); break;
        case 2: ok = try_prefilter<2, false, 16, 1>(a, s); break;
        case 3: ok = try_prefilter<3, false, 16, 1>(a, s); break;
        case 4: ok = try_prefilter<4, false, 16, 1>(a, s); break;
        case 6:
#ifdef LB_DIAG
            if (cfg == 10) ok = try_prefilter_reg<6, 16>(a, s);
            else if (cfg == 11) ok = try_prefilter_reg<6, 12>(a, s);
            else if (cfg == 12) ok = try_prefilter_reg<6, 8>(a, s);
            else if (cfg == 13) ok = try_prefilter_reg<6, 12>(a, s, 2);
            else if (cfg == 14) ok = try_prefilter_reg<6, 8>(a, s, 3);
            else if (cfg == 15) ok = try_prefilter_reg<6, 14>(a, s, 2);
            else if (cfg == 1) ok = try_prefilter<6, false, 11, 2>(a, s);
            else if (cfg == 2) ok = try_prefilter<6, false, 8, 2>(a, s);
            else if (cfg == 3) ok = try_prefilter<6, false, 12, 1>(a, s);
            else if (cfg == 4) ok = try_prefilter<6, true, 16, 1>(a, s);
            else if (cfg == 5) ok = try_prefilter<6, false, 16, 1, true>(a, s);
            else
#endif
                ok = try_prefilter<6, false, 16, 1>(a, s);
            break;
        case 8: ok = try_prefilter<8, false, 10, 1>(a, s); break;
        default: break;
        }
        (void)cfg;
        if (ok) return true;
    }
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
    const bool vec = (a.M % 16 == 0) && ((reinterpret_cast<uintptr_t>(a.codes) & 15) == 0);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < raw; i += gridDim.x * 256u) {
        const uint32_t row = a.cand[i];
        const uint8_t *c = a.codes + (int64_t)row * a.M;
        float sum = 0.f;
        if (vec) {
            // the row's code bytes first (M/16 independent 16-B loads), then 16 independent table reads per
            // group; only the adds are ordered
            for (int g = 0; g < a.M / 16; g++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(c + g * 16);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                float t[16];
#pragma unroll
                for (int u = 0; u < 16; u++) t[u] = a.table[(g * 16 + u) * 256 + ((w[u >> 2] >> (8 * (u & 3))) & 0xffu)];
#pragma unroll
                for (int u = 0; u < 16; u++) sum = sum + t[u];
            }
        } else {
            for (int j = 0; j < a.M; j++) sum = sum + a.table[j * 256 + c[j]];
        }
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
    hipLaunchKernelGGL(adc_exact_candidates_kernel, dim3(256), dim3(256), 0, s, a);
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
