// pq.hip -- host side of the PQ/ADC entry points of include/longbow_gpu.h.
//
// Codebooks arrive as the reference's own serialised blob
// (internal/pq/persistence.go:9-35) with DeserializePQEncoder's validation
// (persistence.go:38-73).  K must be 256: simd.adcBatchGeneric hard-codes the
// table stride 256 (internal/simd/simd.go:350) while pq.BuildADCTable writes
// stride K (internal/pq/adc_table.go:46); they agree only at K = 256.
#include "../../include/longbow_gpu.h"
#include "lb_device.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

using namespace lb;

namespace {
struct HipErrP {
    hipError_t e;
    const char *what;
};
#define LBP_HIP(call)                                       \
    do {                                                    \
        hipError_t _e = (call);                             \
        if (_e != hipSuccess) throw HipErrP{_e, #call};     \
    } while (0)

uint32_t rd_u32le(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
} // namespace

struct lb_gpu_pq {
    int device = 0, dims = 0, M = 0, K = 0, sub = 0;
    std::shared_mutex mu;
    float *d_codebooks = nullptr;
    uint8_t *d_codes = nullptr;
    int64_t n = 0, capacity = 0;
    hipStream_t stream = nullptr;
    // sample buffers (sampled admission threshold) are pooled: a 3 MB hipMalloc per search costs 0.25 ms
    std::mutex samp_mu;
    std::vector<std::pair<uint64_t *, size_t>> samp_free;
    mutable std::mutex err_mu;
    std::string last_error;
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

bool pq_device_ok(int device)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return false;
    return device >= 0 && device < cnt;
}

void pq_grow(lb_gpu_pq *p, int64_t need)
{
    if (need <= p->capacity) return;
    int64_t cap = std::max<int64_t>(std::max<int64_t>(need, p->capacity * 2), 4096);
    uint8_t *nc = nullptr;
    LBP_HIP(hipMalloc(&nc, (size_t)cap * p->M));
    if (p->n > 0) LBP_HIP(hipMemcpy(nc, p->d_codes, (size_t)p->n * p->M, hipMemcpyDeviceToDevice));
    if (p->d_codes) (void)hipFree(p->d_codes);
    p->d_codes = nc;
    p->capacity = cap;
}

struct PqScratch {
    CandState cs{};
    float *d_tables = nullptr;
    int *d_slots = nullptr;
    uint64_t *d_samp = nullptr; // ADC entries of the sampled rows (borrowed from the handle's pool)
    size_t samp_entries = 0;
    lb_gpu_pq *owner = nullptr;
    ~PqScratch()
    {
        if (d_slots) (void)hipFree(d_slots);
        if (d_samp && owner) {
            std::lock_guard<std::mutex> g(owner->samp_mu);
            if (owner->samp_free.size() < 4) {
                owner->samp_free.emplace_back(d_samp, samp_entries);
                d_samp = nullptr;
            }
        }
        if (d_samp) (void)hipFree(d_samp);
        if (cs.lists) (void)hipFree(cs.lists);
        if (cs.cnt) (void)hipFree(cs.cnt);
        if (cs.tau) (void)hipFree(cs.tau);
        if (cs.flags) (void)hipFree(cs.flags);
        if (d_tables) (void)hipFree(d_tables);
    }
};
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
    if (!pq_device_ok(device)) { st(LB_ERR_NO_DEVICE); return nullptr; }
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
        if (p->d_codebooks) (void)hipFree(p->d_codebooks);
        if (p->d_codes) (void)hipFree(p->d_codes);
        for (auto &b : p->samp_free) (void)hipFree(b.first);
        p->samp_free.clear();
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
int64_t lb_gpu_pq_ntotal(const lb_gpu_pq *p) { return p ? p->n : 0; }

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

int lb_gpu_pq_build_adc_table(lb_gpu_pq *p, const float *query, float *table)
{
    if (!p || !query || !table) return LB_ERR_INVALID_ARG;
    std::shared_lock<std::shared_mutex> g(p->mu);
    float *d_q = nullptr, *d_t = nullptr;
    int rc = LB_OK;
    try {
        LBP_HIP(hipSetDevice(p->device));
        LBP_HIP(hipMalloc(&d_q, (size_t)p->dims * 4));
        LBP_HIP(hipMalloc(&d_t, (size_t)p->M * p->K * 4));
        LBP_HIP(hipMemcpy(d_q, query, (size_t)p->dims * 4, hipMemcpyHostToDevice));
        launch_build_adc_table(p->d_codebooks, p->M, p->K, p->sub, d_q, 1, d_t, nullptr);
        LBP_HIP(hipMemcpy(table, d_t, (size_t)p->M * p->K * 4, hipMemcpyDeviceToHost));
    } catch (const HipErrP &e) {
        rc = pq_fail(p, e);
    }
    if (d_q) (void)hipFree(d_q);
    if (d_t) (void)hipFree(d_t);
    return rc;
}

int lb_gpu_pq_adc_distance_batch(lb_gpu_pq *p, const float *table, int64_t row0, int64_t n, float *results)
{
    if (!p || n < 0 || row0 < 0) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK; // adc_table.go:58-60
    if (!table || !results) return LB_ERR_INVALID_ARG;
    std::shared_lock<std::shared_mutex> g(p->mu);
    if (row0 + n > p->n) { p->set_error("flatCodes buffer too small"); return LB_ERR_INVALID_ARG; } // adc_table.go:61-63
    float *d_t = nullptr, *d_r = nullptr;
    int rc = LB_OK;
    try {
        LBP_HIP(hipSetDevice(p->device));
        LBP_HIP(hipMalloc(&d_t, (size_t)p->M * 256 * 4));
        LBP_HIP(hipMalloc(&d_r, (size_t)n * 4));
        LBP_HIP(hipMemcpy(d_t, table, (size_t)p->M * 256 * 4, hipMemcpyHostToDevice));
        CandState cs{};
        launch_adc_scan(d_t, p->M, p->d_codes, row0, row0 + n, 0, nullptr, cs, false, d_r, row0, nullptr);
        LBP_HIP(hipMemcpy(results, d_r, (size_t)n * 4, hipMemcpyDeviceToHost));
    } catch (const HipErrP &e) {
        rc = pq_fail(p, e);
    }
    if (d_t) (void)hipFree(d_t);
    if (d_r) (void)hipFree(d_r);
    return rc;
}

int lb_gpu_pq_search_device(lb_gpu_pq *p, int64_t nq, const float *d_queries, int k, float *d_dist,
                            int64_t *d_labels, void *stream)
{
    if (!p || nq < 0 || k <= 0 || (nq > 0 && (!d_queries || !d_dist || !d_labels))) return LB_ERR_INVALID_ARG;
    if (nq == 0) return LB_OK;
    if (k > 4096) { p->set_error("k=%d exceeds the supported maximum 4096", k); return LB_ERR_UNSUPPORTED; }
    std::shared_lock<std::shared_mutex> g(p->mu);
    PqScratch sc;
    try {
        LBP_HIP(hipSetDevice(p->device));
        hipStream_t s = stream ? (hipStream_t)stream : p->stream;
        const uint32_t cap = std::max<uint32_t>(8192u, 4u * next_pow2_host((uint32_t)k));
        sc.cs.cap = cap;
        const int nqi = (int)nq;
        LBP_HIP(hipMalloc(&sc.cs.lists, (size_t)nqi * cap * 8));
        LBP_HIP(hipMalloc(&sc.cs.cnt, (size_t)nqi * 4));
        LBP_HIP(hipMalloc(&sc.cs.tau, (size_t)nqi * 8));
        LBP_HIP(hipMalloc(&sc.cs.flags, (size_t)nqi * 4));
        LBP_HIP(hipMalloc(&sc.d_tables, (size_t)nqi * p->M * 256 * 4));
        launch_build_adc_table(p->d_codebooks, p->M, p->K, p->sub, d_queries, nqi, sc.d_tables, s);
        // device list 0..nq-1 so one slot can be addressed as a 1-element selection
        std::vector<int> slots(nqi);
        for (int q = 0; q < nqi; q++) slots[q] = q;
        LBP_HIP(hipMalloc(&sc.d_slots, (size_t)nqi * sizeof(int)));
        LBP_HIP(hipMemcpyAsync(sc.d_slots, slots.data(), (size_t)nqi * sizeof(int), hipMemcpyHostToDevice, s));
        // Sampled admission threshold (same reasoning as index.hip: sample_plan): one row in `stride` is scored
        // exactly, the m-th best sample entry becomes tau, and the codes are walked once.  About m*stride rows
        // pass (2048 at 100M rows); fewer than k or more than the list holds is detected by the select and
        // the query is redone by the bootstrap schedule.  Two-level m-th minimum: the sample (390k entries at
        // 100M rows) is larger than one list.
        static const int sample_on = [] { const char *e = getenv("LB_SAMPLE_TAU"); return e ? atoi(e) : 1; }();
        uint32_t samp_count = 0;
        int samp_m = 0;
        if (sample_on && p->n >= 65536 && p->n < ((int64_t)1 << 32)) {
            const int64_t cnt = std::max<int64_t>(8192, (p->n + 255) / 256);
            const double lambda = (double)k * (double)cnt / (double)p->n;
            const int m = std::max(8, (int)std::ceil(lambda + 5.0 * std::sqrt(lambda) + 4.0));
            const double loose = (double)m * ((double)p->n / (double)cnt) * (1.0 + 5.0 / std::sqrt((double)m));
            if (m <= 32 && loose <= (double)(cap - (uint32_t)k) && cnt <= (int64_t)8192 * (8192 / m)) {
                samp_count = (uint32_t)cnt;
                samp_m = m;
                sc.owner = p;
                {
                    std::lock_guard<std::mutex> g2(p->samp_mu);
                    for (size_t i = 0; i < p->samp_free.size(); i++)
                        if (p->samp_free[i].second >= samp_count) {
                            sc.d_samp = p->samp_free[i].first;
                            sc.samp_entries = p->samp_free[i].second;
                            p->samp_free.erase(p->samp_free.begin() + (long)i);
                            break;
                        }
                }
                if (!sc.d_samp) {
                    LBP_HIP(hipMalloc(&sc.d_samp, (size_t)samp_count * sizeof(uint64_t)));
                    sc.samp_entries = samp_count;
                }
            }
        }
        // mode 0: sampled threshold, 1: bootstrap chunks, 2: chunks that cannot overflow the list
        auto scan_query = [&](int q, int mode) {
            const float *tab = sc.d_tables + (size_t)q * p->M * 256;
            launch_init_cand(sc.cs, sc.d_slots + q, 1, s);
            if (mode == 0 && samp_count) {
                launch_adc_sample(tab, p->M, p->d_codes, p->n, samp_count, sc.d_samp, s);
                const uint32_t groups = launch_sample_topm(sc.d_samp, samp_count, samp_m, sc.cs, q, s);
                if (groups) {
                    launch_sample_tau(sc.cs, sc.d_slots + q, 1, groups * (uint32_t)samp_m, samp_m, false, s);
                    launch_adc_scan(tab, p->M, p->d_codes, 0, p->n, q, nullptr, sc.cs, false, nullptr, 0, s);
                    launch_select(sc.cs, sc.d_slots + q, 1, k, 0u, s, false, (uint32_t)std::min<int64_t>(k, p->n));
                    return;
                }
            }
            int64_t pos = 0;
            int step = 0;
            while (pos < p->n) {
                const int64_t end = chunk_end_host(step, pos, p->n, k, cap, mode == 2, /*big_boot=*/true);
                const bool boot = step == 0;
                launch_adc_scan(tab, p->M, p->d_codes, pos, end, q, nullptr, sc.cs, boot, nullptr, 0, s);
                launch_select(sc.cs, sc.d_slots + q, 1, k, boot ? (uint32_t)(end - pos) : 0u, s);
                pos = end;
                step++;
            }
        };
        for (int q = 0; q < nqi; q++) scan_query(q, 0);
        std::vector<uint32_t> flags(nqi);
        auto read_flags = [&]() {
            LBP_HIP(hipMemcpyAsync(flags.data(), sc.cs.flags, (size_t)nqi * 4, hipMemcpyDeviceToHost, s));
            LBP_HIP(hipStreamSynchronize(s));
        };
        read_flags();
        if (samp_count) {
            bool any = false;
            for (int q = 0; q < nqi; q++)
                if (flags[q] & (1u | 4u)) { scan_query(q, 1); any = true; } // the sampled threshold missed
            if (any) read_flags();
        }
        for (int q = 0; q < nqi; q++)
            if (flags[q] & 1u) scan_query(q, 2); // chunks that cannot overflow the list
        launch_emit_lists(sc.cs, nullptr, nqi, k, nullptr, d_dist, d_labels, nullptr, s);
        LBP_HIP(hipStreamSynchronize(s));
    } catch (const HipErrP &e) {
        return pq_fail(p, e);
    }
    return LB_OK;
}

int lb_gpu_pq_search(lb_gpu_pq *p, int64_t nq, const float *queries, int k, float *dist, int64_t *labels)
{
    if (!p || nq < 0 || k <= 0 || (nq > 0 && (!queries || !dist || !labels))) return LB_ERR_INVALID_ARG;
    if (nq == 0) return LB_OK;
    float *d_q = nullptr, *d_d = nullptr;
    int64_t *d_l = nullptr;
    int rc = LB_OK;
    try {
        LBP_HIP(hipSetDevice(p->device));
        LBP_HIP(hipMalloc(&d_q, (size_t)nq * p->dims * 4));
        LBP_HIP(hipMalloc(&d_d, (size_t)nq * k * 4));
        LBP_HIP(hipMalloc(&d_l, (size_t)nq * k * 8));
        LBP_HIP(hipMemcpy(d_q, queries, (size_t)nq * p->dims * 4, hipMemcpyHostToDevice));
        rc = lb_gpu_pq_search_device(p, nq, d_q, k, d_d, d_l, nullptr);
        if (rc == LB_OK) {
            LBP_HIP(hipMemcpy(dist, d_d, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
            LBP_HIP(hipMemcpy(labels, d_l, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
        }
    } catch (const HipErrP &e) {
        rc = pq_fail(p, e);
    }
    if (d_q) (void)hipFree(d_q);
    if (d_d) (void)hipFree(d_d);
    if (d_l) (void)hipFree(d_l);
    return rc;
}

} // extern "C"
