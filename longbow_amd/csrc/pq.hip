// pq.hip -- host side of the PQ/ADC entry points of include/longbow_gpu.h.
//
// Codebooks arrive as the reference's own serialised blob
// (internal/pq/persistence.go:9-35) with DeserializePQEncoder's validation
// (persistence.go:38-73).  K must be 256: simd.adcBatchGeneric hard-codes the
// table stride 256 (internal/simd/simd.go:350) while pq.BuildADCTable writes
// stride K (internal/pq/adc_table.go:46); they agree only at K = 256.
#include "../../include/longbow_gpu.h"
#include "lb_device.h"
#include "lb_host.h"

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

using namespace lb;

namespace {
using HipErrP = lb::HipErr;
#define LBP_HIP(call) LB_HIP(call)

uint32_t rd_u32le(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
struct PqScratch;
} // namespace

struct lb_gpu_pq {
    int device = 0, dims = 0, M = 0, K = 0, sub = 0;
    std::shared_mutex mu;
    float *d_codebooks = nullptr;
    uint8_t *d_codes = nullptr;
    int64_t n = 0, capacity = 0;
    hipStream_t stream = nullptr;
    // per-search scratch (lists, tables, sample and candidate buffers) is pooled on the handle: a search
    // must not hipMalloc/hipFree (the latter synchronises the device under every concurrent search)
    std::mutex sc_mu;
    std::vector<std::unique_ptr<PqScratch>> sc_free;
    mutable std::mutex err_mu;
    std::string last_error;
    // instrumentation (bench.py): HIP events around the main code pass and the whole search of the last query
    std::atomic<int> profiling{0};
    std::atomic<int> prefilter{1}; // 0 = exact f32-table pass only (lb_gpu_pq_set_prefilter; both are exact)
    std::atomic<int> pair_pass{1}; // 0 = one query per pass over the codes even in batches (A/B, diagnostic build)
    SearchCombiner combiner;       // concurrent host-pointer searches of a few queries each are combined (lb_host.h)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float prof_ms[2] = {0.f, 0.f};
    void set_error(const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        std::lock_guard<std::mutex> g(err_mu);
        last_error = buf;
    }
};

namespace {
int pq_fail(lb_gpu_pq *p, const HipErrP &e)
{
    p->set_error("HIP error %d (%s) in %s", (int)e.e, hipGetErrorString(e.e), e.what);
    return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
}

// Geometric growth without a transient second copy of a large code buffer: beyond 1 GiB the buffer grows
// by at most 25 % + the request, and the copy runs in 256 MiB pieces (the peak is still old + new; a
// caller that knows the final size avoids it with lb_gpu_pq_reserve).
void pq_grow(lb_gpu_pq *p, int64_t need)
{
    if (need <= p->capacity) return;
    int64_t cap = std::max<int64_t>(std::max<int64_t>(need, p->capacity * 2), 4096);
    if ((size_t)p->capacity * p->M > ((size_t)1 << 30)) cap = std::max<int64_t>(need, p->capacity + p->capacity / 4);
    uint8_t *nc = nullptr;
    LBP_HIP(hipMalloc(&nc, (size_t)cap * p->M));
    if (p->n > 0) {
        hipError_t e = hipMemcpy(nc, p->d_codes, (size_t)p->n * p->M, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            (void)hipFree(nc);
            throw HipErrP{e, "hipMemcpy (pq_grow)"};
        }
    }
    if (p->d_codes) (void)hipFree(p->d_codes);
    p->d_codes = nc;
    p->capacity = cap;
}

constexpr uint32_t kCandCap = 65536; // prefilter survivors per query (expected: a few thousand)

struct PqScratch {
    int device = 0;
    int nq_cap = 0;
    uint32_t cap = 0;
    int M = 0;
    CandState cs{};
    float *d_tables = nullptr;   // [nq][M*256] f32
    uint8_t *d_qtabs = nullptr;  // [nq][M*256] u8
    float *d_minrng = nullptr;   // [nq][M][4]: subtable minimum, range, bad flag
    int *d_params = nullptr;     // [nq][4]
    uint32_t *d_cand = nullptr;  // [4][kCandCap] survivors of the (up to four) queries in flight
    uint32_t *d_cand_cnt = nullptr; // [nq]
    int *d_slots = nullptr;      // 0..nq-1
    uint64_t *d_samp = nullptr;  // ADC entries of the sampled rows
    size_t samp_entries = 0;
    uint32_t *h_flags = nullptr; // pinned
    ~PqScratch()
    {
        (void)hipSetDevice(device);
        if (cs.lists) (void)hipFree(cs.lists);
        if (cs.cnt) (void)hipFree(cs.cnt);
        if (cs.tau) (void)hipFree(cs.tau);
        if (cs.flags) (void)hipFree(cs.flags);
        if (d_tables) (void)hipFree(d_tables);
        if (d_qtabs) (void)hipFree(d_qtabs);
        if (d_minrng) (void)hipFree(d_minrng);
        if (d_params) (void)hipFree(d_params);
        if (d_cand) (void)hipFree(d_cand);
        if (d_cand_cnt) (void)hipFree(d_cand_cnt);
        if (d_slots) (void)hipFree(d_slots);
        if (d_samp) (void)hipFree(d_samp);
        if (h_flags) (void)hipHostFree(h_flags);
    }
};

std::unique_ptr<PqScratch> acquire_scratch(lb_gpu_pq *p, int nq, uint32_t cap, size_t samp_entries)
{
    std::unique_ptr<PqScratch> sc;
    {
        std::lock_guard<std::mutex> g(p->sc_mu);
        for (size_t i = 0; i < p->sc_free.size(); i++)
            if (p->sc_free[i]->nq_cap >= nq && p->sc_free[i]->cap == cap) {
                sc = std::move(p->sc_free[i]);
                p->sc_free.erase(p->sc_free.begin() + (long)i);
                break;
            }
    }
    if (!sc) {
        sc = std::make_unique<PqScratch>();
        sc->device = p->device;
        sc->nq_cap = std::max(nq, 4);
        sc->cap = cap;
        sc->M = p->M;
        sc->cs.cap = cap;
        const size_t nqc = (size_t)sc->nq_cap;
        LBP_HIP(hipMalloc(&sc->cs.lists, nqc * cap * 8));
        LBP_HIP(hipMalloc(&sc->cs.cnt, nqc * 4));
        LBP_HIP(hipMalloc(&sc->cs.tau, nqc * 8));
        LBP_HIP(hipMalloc(&sc->cs.flags, nqc * 4));
        LBP_HIP(hipMalloc(&sc->d_tables, nqc * p->M * 256 * 4));
        LBP_HIP(hipMalloc(&sc->d_qtabs, nqc * p->M * 256));
        LBP_HIP(hipMalloc(&sc->d_minrng, nqc * p->M * 4 * sizeof(float)));
        LBP_HIP(hipMalloc(&sc->d_params, nqc * 4 * sizeof(int)));
        LBP_HIP(hipMalloc(&sc->d_cand, (size_t)4 * kCandCap * 4));
        LBP_HIP(hipMalloc(&sc->d_cand_cnt, nqc * 4));
        LBP_HIP(hipMalloc(&sc->d_slots, nqc * sizeof(int)));
        LBP_HIP(hipHostMalloc(&sc->h_flags, nqc * 4, hipHostMallocDefault));
        std::vector<int> slots(nqc);
        for (size_t q = 0; q < nqc; q++) slots[q] = (int)q;
        LBP_HIP(hipMemcpy(sc->d_slots, slots.data(), nqc * sizeof(int), hipMemcpyHostToDevice));
    }
    if (sc->samp_entries < samp_entries) {
        if (sc->d_samp) (void)hipFree(sc->d_samp);
        sc->d_samp = nullptr;
        sc->samp_entries = 0;
        LBP_HIP(hipMalloc(&sc->d_samp, samp_entries * sizeof(uint64_t)));
        sc->samp_entries = samp_entries;
    }
    return sc;
}

void release_scratch(lb_gpu_pq *p, std::unique_ptr<PqScratch> sc)
{
    std::lock_guard<std::mutex> g(p->sc_mu);
    if (p->sc_free.size() < 4) p->sc_free.push_back(std::move(sc));
}

} // namespace

extern "C" {

lb_gpu_pq *lb_gpu_pq_new(int device, const uint8_t *blob, size_t len, int *out_status)
{
    auto st = [&](int v) { if (out_status) *out_status = v; };
    if (!blob || len < 12) { st(LB_ERR_INVALID_ARG); return nullptr; } // "invalid PQ data: too short"
    const uint32_t dims = rd_u32le(blob), M = rd_u32le(blob + 4), K = rd_u32le(blob + 8);
    if (M == 0 || dims % M != 0) { st(LB_ERR_INVALID_ARG); return nullptr; } // "invalid PQ parameters"
    const size_t sub = dims / M;
    if (len != 12 + (size_t)M * K * sub * 4) { st(LB_ERR_INVALID_ARG); return nullptr; } // "size mismatch"
    if (K != 256) { st(LB_ERR_UNSUPPORTED); return nullptr; }
    if ((size_t)M * 256 * 4 > 160 * 1024 - 1024) { st(LB_ERR_UNSUPPORTED); return nullptr; } // table must fit LDS
    if (!device_ok(device)) { st(LB_ERR_NO_DEVICE); return nullptr; }
    auto *p = new (std::nothrow) lb_gpu_pq();
    if (!p) { st(LB_ERR_OOM); return nullptr; }
    p->device = device; p->dims = (int)dims; p->M = (int)M; p->K = (int)K; p->sub = (int)sub;
    try {
        LBP_HIP(hipSetDevice(device));
        LBP_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        LBP_HIP(hipMalloc(&p->d_codebooks, len - 12));
        // f32 little-endian on the wire == host/device layout on this platform
        LBP_HIP(hipMemcpy(p->d_codebooks, blob + 12, len - 12, hipMemcpyHostToDevice));
    } catch (const HipErrP &e) {
        st(e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP);
        lb_gpu_pq_free(p);
        return nullptr;
    }
    st(LB_OK);
    return p;
}

void lb_gpu_pq_free(lb_gpu_pq *p)
{
    if (!p) return;
    {
        std::unique_lock<std::shared_mutex> g(p->mu);
        (void)hipSetDevice(p->device);
        (void)hipDeviceSynchronize();
        {
            std::lock_guard<std::mutex> g2(p->sc_mu);
            p->sc_free.clear();
        }
        if (p->d_codebooks) (void)hipFree(p->d_codebooks);
        if (p->d_codes) (void)hipFree(p->d_codes);
        for (auto &e : p->ev)
            if (e) (void)hipEventDestroy(e);
        if (p->stream) (void)hipStreamDestroy(p->stream);
    }
    delete p;
}

const char *lb_gpu_pq_last_error(const lb_gpu_pq *p)
{
    if (!p) return "null handle";
    std::lock_guard<std::mutex> g(p->err_mu);
    return p->last_error.c_str();
}

int lb_gpu_pq_m(const lb_gpu_pq *p) { return p ? p->M : 0; }
int lb_gpu_pq_dims(const lb_gpu_pq *p) { return p ? p->dims : 0; }
int64_t lb_gpu_pq_ntotal(const lb_gpu_pq *p)
{
    if (!p) return 0;
    std::shared_lock<std::shared_mutex> g(const_cast<lb_gpu_pq *>(p)->mu); // (add_codes / add_vectors commit under the writer lock)
    return p->n;
}

int lb_gpu_pq_reserve(lb_gpu_pq *p, int64_t n_total)
{
    if (!p || n_total < 0) return LB_ERR_INVALID_ARG;
    std::unique_lock<std::shared_mutex> g(p->mu);
    try {
        LBP_HIP(hipSetDevice(p->device));
        pq_grow(p, n_total);
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

static int add_codes_impl(lb_gpu_pq *p, int64_t n, const uint8_t *codes, bool on_device)
{
    if (!p || n < 0 || (n > 0 && !codes)) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    std::unique_lock<std::shared_mutex> g(p->mu);
    if (p->n + n > (int64_t)0xffffffffll) { p->set_error("more than 2^32 codes per device"); return LB_ERR_UNSUPPORTED; }
    try {
        LBP_HIP(hipSetDevice(p->device));
        pq_grow(p, p->n + n);
        LBP_HIP(hipMemcpy(p->d_codes + (size_t)p->n * p->M, codes, (size_t)n * p->M,
                          on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
        p->n += n;
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

int lb_gpu_pq_add_codes(lb_gpu_pq *p, int64_t n, const uint8_t *codes) { return add_codes_impl(p, n, codes, false); }
int lb_gpu_pq_add_codes_device(lb_gpu_pq *p, int64_t n, const uint8_t *d_codes) { return add_codes_impl(p, n, d_codes, true); }

int lb_gpu_pq_get_codes(lb_gpu_pq *p, int64_t row0, int64_t n, uint8_t *codes)
{
    if (!p || row0 < 0 || n < 0 || (n > 0 && !codes)) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    std::shared_lock<std::shared_mutex> g(p->mu);
    if (row0 + n > p->n) { p->set_error("rows [%lld, %lld) outside the %lld stored codes", (long long)row0, (long long)(row0 + n), (long long)p->n); return LB_ERR_INVALID_ARG; }
    try {
        LBP_HIP(hipSetDevice(p->device));
        LBP_HIP(hipMemcpy(codes, p->d_codes + (size_t)row0 * p->M, (size_t)n * p->M, hipMemcpyDeviceToHost));
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

// ---- Encode / Decode -------------------------------------------------------------------
int lb_gpu_pq_encode_device(lb_gpu_pq *p, int64_t n, const float *d_vectors, uint8_t *d_codes, void *stream)
{
    if (!p || n < 0 || (n > 0 && (!d_vectors || !d_codes))) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    std::shared_lock<std::shared_mutex> g(p->mu);
    try {
        LBP_HIP(hipSetDevice(p->device));
        hipStream_t s = stream ? (hipStream_t)stream : p->stream;
        launch_pq_encode(p->d_codebooks, p->M, p->K, p->sub, d_vectors, n, d_codes, s);
        LB_LAUNCH_CHECK();
        LBP_HIP(hipStreamSynchronize(s));
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

int lb_gpu_pq_encode(lb_gpu_pq *p, int64_t n, const float *vectors, uint8_t *codes)
{
    if (!p || n < 0 || (n > 0 && (!vectors || !codes))) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    try {
        LBP_HIP(hipSetDevice(p->device));
        // in pieces of <= 64 Mi floats so that the staging buffers stay small and reusable
        const int64_t piece = std::max<int64_t>(1, ((int64_t)64 << 20) / p->dims);
        Lease dv(p->device, (size_t)std::min(n, piece) * p->dims * 4), dc(p->device, (size_t)std::min(n, piece) * p->M);
        for (int64_t r0 = 0; r0 < n; r0 += piece) {
            const int64_t cnt = std::min(piece, n - r0);
            LBP_HIP(hipMemcpy(dv.p, vectors + (size_t)r0 * p->dims, (size_t)cnt * p->dims * 4, hipMemcpyHostToDevice));
            const int rc = lb_gpu_pq_encode_device(p, cnt, dv.as<float>(), dc.as<uint8_t>(), nullptr);
            if (rc != LB_OK) return rc;
            LBP_HIP(hipMemcpy(codes + (size_t)r0 * p->M, dc.p, (size_t)cnt * p->M, hipMemcpyDeviceToHost));
        }
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

int lb_gpu_pq_add_vectors_device(lb_gpu_pq *p, int64_t n, const float *d_vectors)
{
    if (!p || n < 0 || (n > 0 && !d_vectors)) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    std::unique_lock<std::shared_mutex> g(p->mu);
    if (p->n + n > (int64_t)0xffffffffll) { p->set_error("more than 2^32 codes per device"); return LB_ERR_UNSUPPORTED; }
    try {
        LBP_HIP(hipSetDevice(p->device));
        pq_grow(p, p->n + n);
        launch_pq_encode(p->d_codebooks, p->M, p->K, p->sub, d_vectors, n, p->d_codes + (size_t)p->n * p->M, p->stream);
        LB_LAUNCH_CHECK();
        LBP_HIP(hipStreamSynchronize(p->stream));
        p->n += n;
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

int lb_gpu_pq_decode_device(lb_gpu_pq *p, int64_t n, const uint8_t *d_codes, float *d_vectors, void *stream)
{
    if (!p || n < 0 || (n > 0 && (!d_vectors || !d_codes))) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    std::shared_lock<std::shared_mutex> g(p->mu);
    try {
        LBP_HIP(hipSetDevice(p->device));
        hipStream_t s = stream ? (hipStream_t)stream : p->stream;
        launch_pq_decode(p->d_codebooks, p->M, p->K, p->sub, d_codes, n, d_vectors, s);
        LB_LAUNCH_CHECK();
        LBP_HIP(hipStreamSynchronize(s));
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

int lb_gpu_pq_decode(lb_gpu_pq *p, int64_t n, const uint8_t *codes, float *vectors)
{
    if (!p || n < 0 || (n > 0 && (!vectors || !codes))) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    try {
        LBP_HIP(hipSetDevice(p->device));
        const int64_t piece = std::max<int64_t>(1, ((int64_t)64 << 20) / p->dims);
        Lease dv(p->device, (size_t)std::min(n, piece) * p->dims * 4), dc(p->device, (size_t)std::min(n, piece) * p->M);
        for (int64_t r0 = 0; r0 < n; r0 += piece) {
            const int64_t cnt = std::min(piece, n - r0);
            LBP_HIP(hipMemcpy(dc.p, codes + (size_t)r0 * p->M, (size_t)cnt * p->M, hipMemcpyHostToDevice));
            const int rc = lb_gpu_pq_decode_device(p, cnt, dc.as<uint8_t>(), dv.as<float>(), nullptr);
            if (rc != LB_OK) return rc;
            LBP_HIP(hipMemcpy(vectors + (size_t)r0 * p->dims, dv.p, (size_t)cnt * p->dims * 4, hipMemcpyDeviceToHost));
        }
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

// ---- ADC table / batch ----------------------------------------------------------------
int lb_gpu_pq_build_adc_table(lb_gpu_pq *p, const float *query, float *table)
{
    if (!p || !query || !table) return LB_ERR_INVALID_ARG;
    std::shared_lock<std::shared_mutex> g(p->mu);
    try {
        LBP_HIP(hipSetDevice(p->device));
        Lease dq(p->device, (size_t)p->dims * 4), dt(p->device, (size_t)p->M * p->K * 4);
        LBP_HIP(hipMemcpyAsync(dq.p, query, (size_t)p->dims * 4, hipMemcpyHostToDevice, p->stream));
        launch_build_adc_table(p->d_codebooks, p->M, p->K, p->sub, dq.as<float>(), 1, dt.as<float>(), p->stream);
        LB_LAUNCH_CHECK();
        LBP_HIP(hipMemcpyAsync(table, dt.p, (size_t)p->M * p->K * 4, hipMemcpyDeviceToHost, p->stream));
        LBP_HIP(hipStreamSynchronize(p->stream));
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

int lb_gpu_pq_adc_distance_batch(lb_gpu_pq *p, const float *table, int64_t row0, int64_t n, float *results)
{
    if (!p || n < 0 || row0 < 0) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK; // adc_table.go:58-60
    if (!table || !results) return LB_ERR_INVALID_ARG;
    std::shared_lock<std::shared_mutex> g(p->mu);
    if (row0 + n > p->n) { p->set_error("flatCodes buffer too small"); return LB_ERR_INVALID_ARG; } // adc_table.go:61-63
    try {
        LBP_HIP(hipSetDevice(p->device));
        Lease dt(p->device, (size_t)p->M * 256 * 4), dr(p->device, (size_t)n * 4);
        LBP_HIP(hipMemcpyAsync(dt.p, table, (size_t)p->M * 256 * 4, hipMemcpyHostToDevice, p->stream));
        CandState cs{};
        launch_adc_scan(dt.as<float>(), p->M, p->d_codes, row0, row0 + n, 0, nullptr, cs, false, dr.as<float>(), row0,
                        p->stream);
        LB_LAUNCH_CHECK();
        LBP_HIP(hipMemcpyAsync(results, dr.p, (size_t)n * 4, hipMemcpyDeviceToHost, p->stream));
        LBP_HIP(hipStreamSynchronize(p->stream));
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

// ---- candidate re-rank (processChunkInternal, PQ branch) ----------------------------------
int lb_gpu_pq_rerank_device(lb_gpu_pq *p, const float *d_query, const int64_t *d_rows, int64_t n, float *d_dist,
                            float *d_score, void *stream)
{
    if (!p || n < 0) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    if (!d_query || !d_rows || !d_dist) return LB_ERR_INVALID_ARG;
    std::shared_lock<std::shared_mutex> g(p->mu);
    try {
        LBP_HIP(hipSetDevice(p->device));
        hipStream_t s = stream ? (hipStream_t)stream : p->stream;
        Lease dt(p->device, (size_t)p->M * 256 * 4);
        launch_build_adc_table(p->d_codebooks, p->M, p->K, p->sub, d_query, 1, dt.as<float>(), s);
        launch_adc_rerank(dt.as<float>(), p->M, p->d_codes, p->n, d_rows, n, d_dist, d_score, s);
        LB_LAUNCH_CHECK();
        LBP_HIP(hipStreamSynchronize(s));
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

int lb_gpu_pq_rerank(lb_gpu_pq *p, const float *query, const int64_t *rows, int64_t n, float *dist, float *score)
{
    if (!p || n < 0) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    if (!query || !rows || !dist) return LB_ERR_INVALID_ARG;
    try {
        LBP_HIP(hipSetDevice(p->device));
        Lease dq(p->device, (size_t)p->dims * 4), drw(p->device, (size_t)n * 8), dd(p->device, (size_t)n * 4),
            ds(p->device, (size_t)n * 4);
        LBP_HIP(hipMemcpy(dq.p, query, (size_t)p->dims * 4, hipMemcpyHostToDevice));
        LBP_HIP(hipMemcpy(drw.p, rows, (size_t)n * 8, hipMemcpyHostToDevice));
        const int rc = lb_gpu_pq_rerank_device(p, dq.as<float>(), drw.as<int64_t>(), n, dd.as<float>(), ds.as<float>(), nullptr);
        if (rc != LB_OK) return rc;
        LBP_HIP(hipMemcpy(dist, dd.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (score) LBP_HIP(hipMemcpy(score, ds.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

int lb_gpu_pq_set_profiling(lb_gpu_pq *p, int enable)
{
    if (!p) return LB_ERR_INVALID_ARG;
    p->profiling.store(enable ? 1 : 0);
    return LB_OK;
}

int lb_gpu_pq_last_timing(const lb_gpu_pq *p, float ms[2])
{
    if (!p || !ms) return LB_ERR_INVALID_ARG;
    ms[0] = p->prof_ms[0];
    ms[1] = p->prof_ms[1];
    return LB_OK;
}

// ---- search ----------------------------------------------------------------------------
int lb_gpu_pq_set_prefilter(lb_gpu_pq *p, int enable)
{
    if (!p) return LB_ERR_INVALID_ARG;
    p->prefilter.store(enable ? 1 : 0);
    return LB_OK;
}

int lb_gpu_pq_search_device_ctx(lb_gpu_pq *p, int64_t nq, const float *d_queries, int k, float *d_dist,
                                int64_t *d_labels, void *stream, const lb_cancel *ctx)
{
    if (!p || nq < 0 || k <= 0 || (nq > 0 && (!d_queries || !d_dist || !d_labels))) return LB_ERR_INVALID_ARG;
    if (nq == 0) return LB_OK;
    if (const int st = ctx_state(ctx)) { p->set_error(st == LB_ERR_CANCELLED ? "context canceled" : "context deadline exceeded"); return st; }
    if (k > 4096) { p->set_error("k=%d exceeds the supported maximum 4096", k); return LB_ERR_UNSUPPORTED; }
    if (nq > 65536) { p->set_error("nq=%lld exceeds 65536 queries per call", (long long)nq); return LB_ERR_UNSUPPORTED; }
    std::shared_lock<std::shared_mutex> g(p->mu);
    std::unique_ptr<PqScratch> scp;
    try {
        LBP_HIP(hipSetDevice(p->device));
        hipStream_t s = stream ? (hipStream_t)stream : p->stream;
        const int nqi = (int)nq;
        // Sampled admission threshold (same reasoning as index.hip: sample_plan): one row in `stride` is scored
        // exactly, the m-th best sample entry becomes tau, and the codes are walked once.  About m*stride rows
        // pass (4096 at stride 512); fewer than k or more than the list holds is detected by the select and
        // the query is redone by the bootstrap schedule.  Two-level m-th minimum: the sample (195k entries at
        // 100M rows) is larger than one list.  Stride 512 with a 16384-entry list (mean + 5 sigma = 11.3k
        // admitted rows) instead of stride 256 / 8192 halves the sampling pass (49 -> 25 us at 100M rows).
        static const int sample_on = lb_tunable("LB_SAMPLE_TAU", 1);
        uint32_t samp_count = 0;
        int samp_m = 0;
        uint32_t cap = std::max<uint32_t>(8192u, 4u * next_pow2_host((uint32_t)k));
        if (sample_on && p->n >= 65536 && p->n < ((int64_t)1 << 32)) {
            const uint32_t cap_s = std::max<uint32_t>(16384u, cap);
            const int64_t stride = p->n >= ((int64_t)8192 * 512) ? 512 : 256;
            const int64_t cnt = std::max<int64_t>(8192, (p->n + stride - 1) / stride);
            const double lambda = (double)k * (double)cnt / (double)p->n;
            const int m = std::max(8, (int)std::ceil(lambda + 5.0 * std::sqrt(lambda) + 4.0));
            const double loose = (double)m * ((double)p->n / (double)cnt) * (1.0 + 5.0 / std::sqrt((double)m));
            if (m <= 32 && loose <= (double)(cap_s - (uint32_t)k) && cnt <= (int64_t)8192 * (8192 / m)) {
                samp_count = (uint32_t)cnt;
                samp_m = m;
                cap = cap_s;
            }
        }
        scp = acquire_scratch(p, nqi, cap, samp_count);
        PqScratch &sc = *scp;
        const bool prefilter = samp_count != 0 && p->prefilter.load() != 0;
        const bool prof = p->profiling.load() != 0;
        if (prof) {
            for (auto &e : p->ev)
                if (!e) LBP_HIP(hipEventCreate(&e));
            LBP_HIP(hipEventRecord(p->ev[0], s));
        }
        launch_build_adc_table(p->d_codebooks, p->M, p->K, p->sub, d_queries, nqi, sc.d_tables, s, prefilter ? sc.d_minrng : nullptr,
                               sc.cs.flags, prefilter ? sc.d_cand_cnt : nullptr); // (also clears the slots' status words)
        // the search's last select writes the k results AND the slot's status word into pinned host memory (no D2H copy)
        const EmitArgs em{k, nullptr, d_dist, d_labels, sc.h_flags};
        // sampled threshold of one query: sample -> m-th best -> tau (cnt = 0); false = no sampled pass for this search
        auto threshold = [&](int q) -> bool {
            const float *tab = sc.d_tables + (size_t)q * p->M * 256;
            launch_adc_sample(tab, p->M, p->d_codes, p->n, samp_count, sc.d_samp, s);
            const uint32_t groups = launch_sample_topm(sc.d_samp, samp_count, samp_m, sc.cs, q, s);
            if (!groups) return false;
            launch_sample_tau(sc.cs, sc.d_slots + q, 1, groups * (uint32_t)samp_m, samp_m, false, s); // sets tau, cnt = 0
            return true;
        };
        // mode 0: sampled threshold (+ byte-table prefilter), 1: bootstrap chunks, 2: chunks that cannot overflow
        auto scan_query = [&](int q, int mode) {
            const float *tab = sc.d_tables + (size_t)q * p->M * 256;
            if (mode == 0 && samp_count) {
                if (threshold(q)) {
                    if (prefilter) {
                        // rows whose byte-table lower bound cannot pass tau are dropped; the survivors are scored
                        // exactly.  params.ok == 0 (decided on the device: a table with NaN / negative / infinite
                        // entries): nothing is admitted, the select below flags the query (fewer than k entries) and
                        // the host redoes it on the exact schedule.
                        int *prm = sc.d_params + q * 4;
                        uint8_t *qtab = sc.d_qtabs + (size_t)q * p->M * 256;
                        launch_adc_quantise(tab, sc.d_minrng + (size_t)q * p->M * 4, p->M, sc.cs.tau + q, qtab, prm, s);
                        if (prof && q == nqi - 1) (void)hipEventRecord(p->ev[2], s);
                        launch_adc_prefilter(qtab, prm, p->M, p->d_codes, p->n, sc.d_cand, kCandCap, sc.d_cand_cnt + q, s);
                        if (prof && q == nqi - 1) (void)hipEventRecord(p->ev[3], s);
                        launch_adc_exact_candidates(tab, p->M, p->d_codes, sc.d_cand, sc.d_cand_cnt + q, kCandCap, prm, q,
                                                    sc.cs, s);
                    } else {
                        if (prof && q == nqi - 1) (void)hipEventRecord(p->ev[2], s);
                        launch_adc_scan(tab, p->M, p->d_codes, 0, p->n, q, nullptr, sc.cs, false, nullptr, 0, s, nullptr);
                        if (prof && q == nqi - 1) (void)hipEventRecord(p->ev[3], s);
                    }
                    // the search's last select also writes the k results (redone queries overwrite them below)
                    launch_select(sc.cs, sc.d_slots + q, 1, k, 0u, s, false, (uint32_t)std::min<int64_t>(k, p->n), &em);
                    return;
                }
            }
            launch_init_cand(sc.cs, sc.d_slots + q, 1, s);
            int64_t pos = 0;
            int step = 0;
            while (pos < p->n) {
                const int64_t end = chunk_end_host(step, pos, p->n, k, cap, mode == 2, /*big_boot=*/true);
                const bool boot = step == 0;
                launch_adc_scan(tab, p->M, p->d_codes, pos, end, q, nullptr, sc.cs, boot, nullptr, 0, s);
                launch_select(sc.cs, sc.d_slots + q, 1, k, boot ? (uint32_t)(end - pos) : 0u, s, false, 0u,
                              end >= p->n ? &em : nullptr);
                pos = end;
                step++;
            }
            if (p->n == 0) launch_emit_lists(sc.cs, sc.d_slots + q, 1, k, nullptr, d_dist, d_labels, sc.h_flags, s);
        };
        // two queries share ONE pass over the codes: thresholds and byte tables for both, then the two-query
        // prefilter, then the exact survivors and the select of each.  false = not applicable (run them one by one)
        auto scan_pair = [&](int q) -> bool {
            if (!prefilter || !samp_count) return false;
            int *prm[2];
            uint8_t *qtab[2];
            for (int j = 0; j < 2; j++) {
                const int qq = q + j;
                const float *tab = sc.d_tables + (size_t)qq * p->M * 256;
                if (!threshold(qq)) return false; // (never after the first of the pair succeeded: same counts)
                prm[j] = sc.d_params + qq * 4;
                qtab[j] = sc.d_qtabs + (size_t)qq * p->M * 256;
                launch_adc_quantise(tab, sc.d_minrng + (size_t)qq * p->M * 4, p->M, sc.cs.tau + qq, qtab[j], prm[j], s);
            }
            const bool last = q + 1 == nqi - 1;
            if (prof && last) (void)hipEventRecord(p->ev[2], s);
            if (!launch_adc_prefilter2(qtab[0], prm[0], sc.d_cand, sc.d_cand_cnt + q, qtab[1], prm[1], sc.d_cand + kCandCap,
                                       sc.d_cand_cnt + q + 1, p->M, p->d_codes, p->n, kCandCap, s)) {
                for (int j = 0; j < 2; j++)
                    launch_adc_prefilter(qtab[j], prm[j], p->M, p->d_codes, p->n, sc.d_cand + (size_t)j * kCandCap, kCandCap,
                                         sc.d_cand_cnt + q + j, s);
            }
            if (prof && last) (void)hipEventRecord(p->ev[3], s);
            for (int j = 0; j < 2; j++) {
                const int qq = q + j;
                const float *tab = sc.d_tables + (size_t)qq * p->M * 256;
                launch_adc_exact_candidates(tab, p->M, p->d_codes, sc.d_cand + (size_t)j * kCandCap, sc.d_cand_cnt + qq, kCandCap,
                                            prm[j], qq, sc.cs, s);
                launch_select(sc.cs, sc.d_slots + qq, 1, k, 0u, s, false, (uint32_t)std::min<int64_t>(k, p->n), &em);
            }
            return true;
        };
        // four queries share ONE pass (interleaved byte tables: one LDS gather per code byte serves all four)
        auto scan_quad = [&](int q) -> bool {
            if (!prefilter || !samp_count) return false;
            const int *prm[4];
            const uint8_t *qtab[4];
            uint32_t *cand[4], *ccnt[4];
            for (int j = 0; j < 4; j++) {
                const int qq = q + j;
                const float *tab = sc.d_tables + (size_t)qq * p->M * 256;
                if (!threshold(qq)) return false;
                int *prm_w = sc.d_params + qq * 4;
                uint8_t *qt = sc.d_qtabs + (size_t)qq * p->M * 256;
                launch_adc_quantise(tab, sc.d_minrng + (size_t)qq * p->M * 4, p->M, sc.cs.tau + qq, qt, prm_w, s);
                prm[j] = prm_w; qtab[j] = qt;
                cand[j] = sc.d_cand + (size_t)j * kCandCap;
                ccnt[j] = sc.d_cand_cnt + qq;
            }
            const bool last = q + 3 == nqi - 1;
            if (prof && last) (void)hipEventRecord(p->ev[2], s);
            if (!launch_adc_prefilter4(qtab, prm, cand, ccnt, p->M, p->d_codes, p->n, kCandCap, s)) {
                for (int j = 0; j < 4; j += 2)
                    if (!launch_adc_prefilter2(qtab[j], prm[j], cand[j], ccnt[j], qtab[j + 1], prm[j + 1], cand[j + 1], ccnt[j + 1], p->M,
                                               p->d_codes, p->n, kCandCap, s))
                        for (int u = j; u < j + 2; u++)
                            launch_adc_prefilter(qtab[u], prm[u], p->M, p->d_codes, p->n, cand[u], kCandCap, ccnt[u], s);
            }
            if (prof && last) (void)hipEventRecord(p->ev[3], s);
            for (int j = 0; j < 4; j++) {
                const int qq = q + j;
                const float *tab = sc.d_tables + (size_t)qq * p->M * 256;
                launch_adc_exact_candidates(tab, p->M, p->d_codes, cand[j], ccnt[j], kCandCap, prm[j], qq, sc.cs, s);
                launch_select(sc.cs, sc.d_slots + qq, 1, k, 0u, s, false, (uint32_t)std::min<int64_t>(k, p->n), &em);
            }
            return true;
        };
        for (int q = 0; q < nqi; q++) {
            if (ctx && q > 0) { // a cancellable call waits for each pass before it enqueues the next (~10 us each)
                LBP_HIP(hipStreamSynchronize(s));
                if (const int st = ctx_state(ctx)) {
                    release_scratch(p, std::move(scp));
                    p->set_error(st == LB_ERR_CANCELLED ? "context canceled" : "context deadline exceeded");
                    return st;
                }
            }
            if (q + 3 < nqi && p->pair_pass.load() != 0 && scan_quad(q)) {
                q += 3;
                continue;
            }
            if (q + 1 < nqi && p->pair_pass.load() != 0 && scan_pair(q)) {
                q++;
                continue;
            }
            scan_query(q, 0);
        }
        auto read_flags = [&]() { // (each query's last select wrote its status word into the pinned h_flags)
            LBP_HIP(hipStreamSynchronize(s));
        };
        read_flags();
        if (samp_count) {
            bool any = false;
            for (int q = 0; q < nqi; q++)
                if (sc.h_flags[q] & (1u | 4u)) { scan_query(q, 1); any = true; } // the sampled threshold missed
            if (any) read_flags();
        }
        for (int q = 0; q < nqi; q++)
            if (sc.h_flags[q] & 1u) scan_query(q, 2); // chunks that cannot overflow the list
        LB_LAUNCH_CHECK();
        if (prof) LBP_HIP(hipEventRecord(p->ev[1], s));
        LBP_HIP(hipStreamSynchronize(s));
        if (prof) {
            float a = 0.f, b = 0.f;
            if (samp_count && hipEventElapsedTime(&a, p->ev[2], p->ev[3]) != hipSuccess) a = 0.f;
            if (hipEventElapsedTime(&b, p->ev[0], p->ev[1]) != hipSuccess) b = 0.f;
            (void)hipGetLastError();
            p->prof_ms[0] = a;
            p->prof_ms[1] = b;
        }
        release_scratch(p, std::move(scp));
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

int lb_gpu_pq_search_device(lb_gpu_pq *p, int64_t nq, const float *d_queries, int k, float *d_dist,
                            int64_t *d_labels, void *stream)
{
    return lb_gpu_pq_search_device_ctx(p, nq, d_queries, k, d_dist, d_labels, stream, nullptr);
}

int lb_gpu_pq_search(lb_gpu_pq *p, int64_t nq, const float *queries, int k, float *dist, int64_t *labels)
{
    return lb_gpu_pq_search_ctx(p, nq, queries, k, dist, labels, nullptr);
}

// Host-pointer search of one or several requests with the same k as ONE device batch (two queries share a pass over the
// codes): borrowed host buffers -> pooled pinned slab -> HBM, results back through the slab -- small ones (the latency path)
// written into it by the last kernel itself.
static int pq_host_search_multi(lb_gpu_pq *p, HostReq *const *reqs, int nreq, int k, const lb_cancel *ctx)
{
    int64_t nq = 0;
    for (int i = 0; i < nreq; i++) nq += reqs[i]->nq;
    if (k > 4096) { p->set_error("k=%d exceeds the supported maximum 4096", k); return LB_ERR_UNSUPPORTED; }
    if (nq > 65536) { p->set_error("nq=%lld exceeds 65536 queries per call", (long long)nq); return LB_ERR_UNSUPPORTED; }
    int rc = LB_OK;
    try {
        LBP_HIP(hipSetDevice(p->device));
        const size_t qb = (size_t)nq * p->dims * 4, db = (((size_t)nq * k * 4) + 15) & ~(size_t)15, lbb = (size_t)nq * k * 8;
        const size_t doff = (qb + 15) & ~(size_t)15, loff = doff + db, total = loff + lbb;
        const bool direct = db + lbb <= ((size_t)64 << 10);
        Lease hs(p->device, total, /*pinned=*/true), dq(p->device, direct ? qb : total);
        char *hb = hs.as<char>(), *dbuf = dq.as<char>();
        size_t off = 0;
        for (int i = 0; i < nreq; i++) {
            const size_t b = (size_t)reqs[i]->nq * p->dims * 4;
            std::memcpy(hb + off, reqs[i]->q, b);
            off += b;
        }
        LBP_HIP(hipMemcpy(dbuf, hb, qb, hipMemcpyHostToDevice));
        char *obuf = direct ? hb : dbuf;
        rc = lb_gpu_pq_search_device_ctx(p, nq, reinterpret_cast<const float *>(dbuf), k, reinterpret_cast<float *>(obuf + doff),
                                         reinterpret_cast<int64_t *>(obuf + loff), nullptr, ctx);
        if (rc == LB_OK) {
            if (!direct) LBP_HIP(hipMemcpy(hb + doff, dbuf + doff, db + lbb, hipMemcpyDeviceToHost));
            size_t row = 0;
            for (int i = 0; i < nreq; i++) {
                const size_t n = (size_t)reqs[i]->nq * k;
                std::memcpy(reqs[i]->dist, hb + doff + row * 4, n * 4);
                std::memcpy(reqs[i]->labels, hb + loff + row * 8, n * 8);
                row += n;
            }
        }
    } catch (const HipErrP &e) {
        rc = pq_fail(p, e);
    } catch (...) {
        p->set_error("internal error (exception)");
        rc = LB_ERR_INTERNAL;
    }
    return rc;
}

int lb_gpu_pq_search_ctx(lb_gpu_pq *p, int64_t nq, const float *queries, int k, float *dist, int64_t *labels,
                         const lb_cancel *ctx)
{
    if (!p || nq < 0 || k <= 0 || (nq > 0 && (!queries || !dist || !labels))) return LB_ERR_INVALID_ARG;
    if (nq == 0) return LB_OK;
    HostReq me{queries, nq, dist, labels, k};
    // (concurrent calls of a few queries each are answered together: two queries share a pass over the codes, 1.48x the
    // queries per second of one call after the other; a call with a cancellation context is searched on its own)
    if (!ctx && nq <= SearchCombiner::kMaxNq && k <= 4096 && p->combiner.on.load() != 0)
        return p->combiner.search(me, [p](HostReq *const *reqs, int n, int kk) { return pq_host_search_multi(p, reqs, n, kk, nullptr); });
    HostReq *one = &me;
    return pq_host_search_multi(p, &one, 1, k, ctx);
}

int lb_gpu_pq_set_search_combining(lb_gpu_pq *p, int enable)
{
    if (!p) return LB_ERR_INVALID_ARG;
    p->combiner.on.store(enable ? 1 : 0);
    return LB_OK;
}

int lb_gpu_pq_combining_stats(const lb_gpu_pq *p, int64_t out[2])
{
    if (!p || !out) return LB_ERR_INVALID_ARG;
    out[0] = p->combiner.batches.load();
    out[1] = p->combiner.requests.load();
    return LB_OK;
}

} // extern "C"
