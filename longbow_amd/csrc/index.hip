// index.hip -- host side of liblongbow_gpu.so: the C ABI of include/longbow_gpu.h.
//
// Mirrors the contract of Longbow's gpu.Index (internal/gpu/interface.go:3-19) as
// the FAISS binding realises it (internal/gpu/faiss_gpu.go:44-167): opaque handle,
// int return codes, Add appends, Search is safe from many threads at once
// (RWMutex: Add/Close exclusive, Search shared), nothing aborts the process.
//
// There is NO CPU fallback in this library: without a HIP device every entry point
// returns LB_ERR_NO_DEVICE / NULL.
#include "../../include/longbow_gpu.h"
#include "lb_device.h"
#include "lb_host.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

using namespace lb;

namespace lb { extern int g_adc_ablation; extern int g_gemm_ablation; extern int g_gemm_glds; void read_clock_probe(unsigned long long out[8], bool reset); int debug_gemm_occupancy(); void read_fused_probe(unsigned long long out[8], bool reset); void read_tall2_probe(unsigned long long out[8], bool reset); void read_tall16_probe(unsigned long long out[8], bool reset); void read_finish_probe(unsigned long long out[8], bool reset); }

namespace {

constexpr int kScanMaxQ = 8;          // queries per scan launch (register accumulators)
constexpr int kGemmMinQ = 17;         // below this the exact scan path is used for everything
constexpr int kMaxBatch = 4096;       // queries per internal batch (workspace sizing)
constexpr size_t kStageBytes = 32u << 20; // pinned staging slab (x2)
constexpr int kFinishSplitMaxQ = 128;     // largest batch the finish launch serves with several workgroups per query
constexpr uint32_t kFinishSmaxMax = 4096; // most members (rows re-ranked exactly) a query may have

struct Event {
    hipEvent_t a = nullptr, b = nullptr;
    int cls = 0;
};

struct Workspace {
    int device = 0;
    hipStream_t stream = nullptr;
    int nq_cap = 0;
    uint32_t cap = 0;
    CandState cs{};
    float *d_qna = nullptr;
    float *d_qs = nullptr; // split-bf16 image of the query batch
    size_t d_qs_bytes = 0;
    void *d_qh = nullptr;  // fp16 image of the query batch (scaled per query) + [nq] inverse scales behind it
    size_t d_qh_bytes = 0;
    int *d_qsel = nullptr;
    int *d_iota = nullptr;       // [nq_cap] 0,1,2,...: the slot list of "every query", filled once
    uint32_t *d_smap = nullptr;  // [cap] sampled rows of the first pass
    uint32_t *d_done = nullptr;  // [nq_cap] arrival tickets of the finish launch's split form (zero between launches)
    uint32_t *d_xcnt = nullptr;  // [nq_cap] members handed in per query by that form (zero between launches)
    void *d_xscratch = nullptr;  // its per-query result blocks (finish_scratch_bytes)
    // fused sample (kernels_gemm_narrow.hip, FUSED): [0] = ticket counter that only grows, [1 ..] = ready epochs per slot
    uint32_t *d_fsync = nullptr;
    uint32_t fs_base = 0, fs_epoch = 0; // host mirror of the ticket counter; last epoch used
    uint32_t *h_fail = nullptr;         // pinned: epoch of a launch whose waits gave up
    uint32_t *h_flags = nullptr; // pinned
    int *h_qsel = nullptr;       // pinned
    // host-API staging (device side)
    float *d_q = nullptr;
    size_t d_q_bytes = 0;
    float *d_dist = nullptr;
    int64_t *d_lab = nullptr;
    size_t d_out_n = 0;
    std::vector<Event> events;
    size_t ev_used = 0;
    const lb_cancel *ctx = nullptr; // the running call's cancellation context (or null)

    ~Workspace()
    {
        (void)hipSetDevice(device);
        if (cs.lists) (void)hipFree(cs.lists);
        if (cs.cnt) (void)hipFree(cs.cnt);
        if (cs.tau) (void)hipFree(cs.tau);
        if (cs.flags) (void)hipFree(cs.flags);
        if (cs.stripes) (void)hipFree(cs.stripes);
        if (d_qna) (void)hipFree(d_qna);
        if (d_qs) (void)hipFree(d_qs);
        if (d_qh) (void)hipFree(d_qh);
        if (d_qsel) (void)hipFree(d_qsel);
        if (d_iota) (void)hipFree(d_iota);
        if (d_smap) (void)hipFree(d_smap);
        if (d_done) (void)hipFree(d_done);
        if (d_xcnt) (void)hipFree(d_xcnt);
        if (d_xscratch) (void)hipFree(d_xscratch);
        if (d_fsync) (void)hipFree(d_fsync);
        if (h_fail) (void)hipHostFree(h_fail);
        if (h_flags) (void)hipHostFree(h_flags);
        if (h_qsel) (void)hipHostFree(h_qsel);
        if (d_q) (void)hipFree(d_q);
        if (d_dist) (void)hipFree(d_dist);
        if (d_lab) (void)hipFree(d_lab);
        for (auto &e : events) {
            if (e.a) (void)hipEventDestroy(e.a);
            if (e.b) (void)hipEventDestroy(e.b);
        }
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// Device + pinned-host staging for the host-pointer search entry point, pooled per index so a
// serving loop does not pay hipMalloc/hipFree per call.
struct HostStage {
    int device = 0;
    hipStream_t stream = nullptr;
    void *d_buf = nullptr, *h_buf = nullptr; // [queries | dist | labels]
    size_t bytes = 0;
    ~HostStage()
    {
        (void)hipSetDevice(device);
        if (d_buf) (void)hipFree(d_buf);
        if (h_buf) (void)hipHostFree(h_buf);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

} // namespace

// The corpus buffer grows IN PLACE: one virtual range the size of the device's HBM is reserved per index
// and physical chunks are mapped behind it as rows arrive (hipMemAddressReserve / hipMemCreate /
// hipMemMap).  Appending never copies the rows already resident and never needs old + new at once, so an
// index can grow to fill the 288 GB.  If the driver refuses any of the calls the index falls back to
// geometric hipMalloc + copy (vmm.ok == false).
std::atomic<int> g_vmm_fail_next{0}; // test hook: the next mapping attempt reports a driver refusal

struct VmmBuf {
    bool ok = false;
    int device = 0;
    char *base = nullptr;
    size_t reserved = 0, mapped = 0, gran = 0;
    struct Chunk { hipMemGenericAllocationHandle_t h; size_t off, bytes; };
    std::vector<Chunk> chunks;

    bool init(int dev)
    {
        device = dev;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t g = 0;
        if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || g == 0) return false;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return false;
        // the driver reports 4 KiB, but maps whose offsets are not 64 KiB-aligned are refused by
        // hipMemSetAccess (measured, tools/probe/vmm_probe.cpp): keep every chunk a multiple of 2 MiB
        gran = g < ((size_t)2 << 20) ? ((size_t)2 << 20) : g;
        reserved = ((total_b + gran - 1) / gran) * gran;
        void *ptr = nullptr;
        if (hipMemAddressReserve(&ptr, reserved, gran, nullptr, 0) != hipSuccess || !ptr) { (void)hipGetLastError(); return false; }
        base = static_cast<char *>(ptr);
        ok = true;
        return true;
    }
    // make [0, need) backed by memory; throws HipErr (OOM) when the device has no more to give
    void ensure(size_t need)
    {
        if (need <= mapped) return;
        if (need > reserved) throw lb::HipErr{hipErrorOutOfMemory, "corpus larger than the device"};
        if (g_vmm_fail_next.exchange(0)) throw lb::HipErr{hipErrorInvalidValue, "hipMemSetAccess (forced by the test hook)"};
        // geometric steps (at least the request, at least what is mapped already, at most 1 GiB beyond the
        // request): small indexes stay small, 288 GB take ~300 handles
        size_t want = need - mapped;
        size_t step = mapped < ((size_t)1 << 30) ? mapped : ((size_t)1 << 30);
        if (want < step) want = step;
        want = ((want + gran - 1) / gran) * gran;
        if (mapped + want > reserved) want = reserved - mapped;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        hipMemGenericAllocationHandle_t hnd;
        hipError_t e = hipMemCreate(&hnd, want, &prop, 0);
        if (e != hipSuccess && want > ((need - mapped + gran - 1) / gran) * gran) { // retry with the exact need
            (void)hipGetLastError();
            want = ((need - mapped + gran - 1) / gran) * gran;
            e = hipMemCreate(&hnd, want, &prop, 0);
        }
        if (e != hipSuccess) throw lb::HipErr{hipErrorOutOfMemory, "hipMemCreate (corpus chunk)"};
        e = hipMemMap(base + mapped, want, 0, hnd, 0);
        if (e != hipSuccess) { (void)hipMemRelease(hnd); throw lb::HipErr{e, "hipMemMap"}; }
        hipMemAccessDesc acc{};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(base + mapped, want, &acc, 1);
        if (e != hipSuccess) {
            (void)hipMemUnmap(base + mapped, want);
            (void)hipMemRelease(hnd);
            throw lb::HipErr{e, "hipMemSetAccess"};
        }
        chunks.push_back({hnd, mapped, want});
        mapped += want;
    }
    void destroy()
    {
        for (auto &c : chunks) {
            (void)hipMemUnmap(base + c.off, c.bytes);
            (void)hipMemRelease(c.h);
        }
        chunks.clear();
        if (base) (void)hipMemAddressFree(base, reserved);
        base = nullptr;
        mapped = reserved = 0;
        ok = false;
    }
};

struct lb_gpu_index {
    int device = 0, dim = 0, metric = 0;
    std::atomic<int> order{LB_ORDER_SEQ};
    std::shared_mutex mu;
    bool closed = false;

    float *d_X = nullptr;
    VmmBuf vmm;             // backs d_X when vmm.ok (d_X == vmm.base): rows are appended in place
    int64_t x_rows_cap = 0; // rows d_X can hold (>= capacity of the side arrays when vmm.ok)
    int64_t n = 0, capacity = 0;
    float *d_norm2 = nullptr, *d_rnorm = nullptr;
    uint32_t *d_maxnorm2 = nullptr;
    bool nonfinite = false; // some row holds an inf / NaN: every search takes the exact scan path
    bool f16_ok = false;    // row norms within the fp16 single-product contraction's range (kernels_gemm_tall16.hip)
    bool norm_spread = false; // the longest row is more than 16 times the shortest non-zero one: dot-product searches then keep
                              // to the kernels with lower-bound keys under AUTO (plain keys leave such corpora to the exact scan)
    // AUTO backs off from the fp16 route on data whose neighbours are too close for its error bound (many queries then
    // fail the containment proof and are redone by the exact scan): searches left to skip it, and the next back-off span
    std::atomic<int> f16_skip{0}, f16_span{16};
    int64_t *d_ids = nullptr;
    bool has_ids = false;
    uint8_t *d_mask = nullptr;
    bool has_mask = false;
    // ascending list of the rows the mask leaves visible (rebuilt whenever the mask or the corpus
    // changes); searches walk it instead of the corpus when the filter is selective enough
    uint32_t *d_rowmap = nullptr, *d_cscratch = nullptr;
    int64_t rowmap_cap = 0, cscratch_words = 0;
    int64_t n_visible = 0;
    bool rowmap_on = false;
    // strided sample of the current corpus view (sample_plan), built by the first batched search after
    // a change and shared by all searches with the same plan
    std::mutex smap_mu;
    uint32_t *d_smap = nullptr;
    int64_t smap_span = 0;
    uint32_t smap_count = 0, smap_cap = 0;
    bool smap_valid = false;
    // optional split-bf16 image of the corpus for the 3x-bf16 candidate contraction (same byte shape as d_X)
    std::atomic<int> cand_mode{LB_CAND_AUTO};
    float *d_Xs = nullptr;
    int64_t xs_rows = 0; // rows of d_X already mirrored in d_Xs
    // fp16 image of the corpus for the single-product route (K-blocked [dim / 32][xh_cap][32], kernels_gemm_tall16.hip): kept
    // while that route is on offer and memory allows (sync_f16_image); half the bytes to stage per batched search
    void *d_Xh = nullptr;
    int64_t xh_rows = 0, xh_cap = 0;
    std::atomic<int> xh_mode{1}; // lb_gpu_index_set_f16_image: 0 never, 1 when it pays and fits
    bool xh_failed = false;      // an allocation was refused: not tried again for this handle
    bool xh_shed = false;        // the copy was given back to let an Add through: retaken only with twice the margin free
    // L2 indexes keep the image CENTRED: fp16(x - c), c = the column means when the image was built.  L2 distances do not move
    // when both sides are shifted, the key |x - c|^2 - 2 (q - c).(x - c) = d^2 - |q - c|^2 orders rows as the plain key does, and
    // its errors scale with the centred norms: data with a large common offset (|c| >> spread), whose plain keys cancel, keeps
    // the matrix-core route.  d_norm2c: [xh_cap] centred norms (the keys' side input); d_cstats: their max / smallest non-zero
    // (float bits, as d_maxnorm2); xh_c_ok: those are within the fp16 contraction's range
    float *d_center = nullptr, *d_norm2c = nullptr;
    uint32_t *d_cstats = nullptr;
    bool xh_centred = false, xh_c_ok = false;
    // what the image loses, measured: max over its rows of |x - fp16(x)| / |x| (kernels_gemm_tall16.hip: f16_residual_kernel);
    // 0 = not measured (the per-element worst case 2^-11 stands in)
    uint32_t *d_xh_rho2 = nullptr;
    float xh_rho = 0.f;
    bool xh_offset_dom = false;  // |c|^2 is several times the largest centred |x - c|^2: plain L2 keys cancel on this data, so
                                 // AUTO keeps batched searches on the centred image whatever the cost model says of other routes
    int64_t xh_declined_n = 0;   // a centred image was out of fp16's range at this many rows: not tried again below twice that
    // data whose neighbours the candidate keys cannot separate (tight clusters): batched searches start with the widened
    // candidate list that proved the last such batch, for the next kc_hint_left searches (search_batch_device)
    std::atomic<int> kc_hint{0}, kc_hint_left{0};

    hipStream_t add_stream = nullptr;
    void *h_stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};

    std::mutex ws_mu;
    std::vector<std::unique_ptr<Workspace>> ws_free;
    std::vector<std::unique_ptr<HostStage>> hs_free;

    SearchCombiner combiner; // concurrent host-pointer searches of a few queries each are combined (lb_host.h)

    mutable std::mutex err_mu;
    std::string last_error;

    std::atomic<int64_t> last_fallbacks{0};
    std::atomic<int> last_route{0}; // RouteKind * 10 + operand form of the most recent batched search (0: exact scan path)
    std::atomic<int64_t> fused_giveups{0}; // fused sample launches whose waits gave up (batch redone on the exact path)
    std::atomic<int> profiling{0};
    std::mutex prof_mu;
    float prof_ms[5] = {0, 0, 0, 0, 0};
    int prof_n[5] = {0, 0, 0, 0, 0};

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

int fail_hip(lb_gpu_index *h, const HipErr &e)
{
    (void)hipGetLastError(); // the failure is reported through the return code; leave no sticky error behind
    h->set_error("HIP error %d (%s) in %s", (int)e.e, hipGetErrorString(e.e), e.what);
    return (e.e == hipErrorOutOfMemory) ? LB_ERR_OOM : LB_ERR_HIP;
}

// candidate-list geometry for a request of k
void cand_geometry(int k, int &kc, uint32_t &cap)
{
    int want = std::max(2 * k, k + 32);
    kc = (int)next_pow2_host((uint32_t)std::max(want, 64));
    cap = std::max<uint32_t>(8192u, 4u * (uint32_t)kc);
    // (from 1024 candidates -- k beyond 240 -- the fp16 routes keep 2048 per query: with 8192-entry lists that is a quarter of the
    // capacity, the sampled span ends short of the corpus and the rest runs the classic schedule -- 1M x 768, k = 300: 0.71 ms a
    // query where k = 100 takes 0.29)
    if (kc >= 1024) cap = std::max<uint32_t>(cap, 16384u);
}

std::unique_ptr<Workspace> acquire_ws(lb_gpu_index *h, int nq, uint32_t cap)
{
    {
        std::lock_guard<std::mutex> g(h->ws_mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < h->ws_free.size(); i++)
            if (h->ws_free[i]->nq_cap >= nq && h->ws_free[i]->cap == cap &&
                (best == (size_t)-1 || h->ws_free[i]->nq_cap < h->ws_free[best]->nq_cap))
                best = i;
        if (best != (size_t)-1) {
            auto w = std::move(h->ws_free[best]);
            h->ws_free.erase(h->ws_free.begin() + (long)best);
            return w;
        }
    }
    auto w = std::make_unique<Workspace>();
    w->device = h->device;
    w->nq_cap = (int)next_pow2_host((uint32_t)std::max(nq, 8)); // nearby batch sizes share a workspace
    w->cap = cap;
    w->cs.cap = cap;
    LB_HIP(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
    LB_HIP(hipMalloc(&w->cs.lists, (size_t)w->nq_cap * cap * sizeof(uint64_t)));
    LB_HIP(hipMalloc(&w->cs.cnt, (size_t)w->nq_cap * sizeof(uint32_t)));
    LB_HIP(hipMalloc(&w->cs.tau, (size_t)w->nq_cap * sizeof(uint64_t)));
    LB_HIP(hipMalloc(&w->cs.flags, (size_t)w->nq_cap * sizeof(uint32_t)));
    LB_HIP(hipMalloc(&w->d_done, (size_t)w->nq_cap * sizeof(uint32_t)));
    LB_HIP(hipMemset(w->d_done, 0, (size_t)w->nq_cap * sizeof(uint32_t)));
    LB_HIP(hipMalloc(&w->d_xcnt, (size_t)w->nq_cap * sizeof(uint32_t)));
    LB_HIP(hipMemset(w->d_xcnt, 0, (size_t)w->nq_cap * sizeof(uint32_t)));
    LB_HIP(hipMalloc(&w->d_xscratch, finish_scratch_bytes(kFinishSplitMaxQ, kFinishSmaxMax)));
    LB_HIP(hipMalloc(&w->d_fsync, (size_t)(1 + w->nq_cap) * sizeof(uint32_t)));
    LB_HIP(hipMemset(w->d_fsync, 0, (size_t)(1 + w->nq_cap) * sizeof(uint32_t)));
    LB_HIP(hipMemset(w->cs.flags, 0, (size_t)w->nq_cap * sizeof(uint32_t)));
    LB_HIP(hipHostMalloc(&w->h_fail, sizeof(uint32_t), hipHostMallocDefault));
    *w->h_fail = 0;
    LB_HIP(hipMalloc(&w->cs.stripes, (size_t)kScanMaxQ * LB_STRIPES * LB_STRIPE_PAD * sizeof(uint32_t)));
    LB_HIP(hipMalloc(&w->d_qna, (size_t)w->nq_cap * sizeof(float)));
    LB_HIP(hipMalloc(&w->d_qsel, (size_t)w->nq_cap * sizeof(int)));
    LB_HIP(hipMalloc(&w->d_iota, (size_t)w->nq_cap * sizeof(int)));
    {
        std::vector<int> iota((size_t)w->nq_cap);
        for (int i = 0; i < w->nq_cap; i++) iota[(size_t)i] = i;
        LB_HIP(hipMemcpy(w->d_iota, iota.data(), iota.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    LB_HIP(hipMalloc(&w->d_smap, (size_t)cap * sizeof(uint32_t)));
    LB_HIP(hipHostMalloc(&w->h_flags, (size_t)w->nq_cap * sizeof(uint32_t), hipHostMallocDefault));
    LB_HIP(hipHostMalloc(&w->h_qsel, (size_t)w->nq_cap * sizeof(int), hipHostMallocDefault));
    return w;
}

void release_ws(lb_gpu_index *h, std::unique_ptr<Workspace> w)
{
    std::unique_ptr<Workspace> drop; // (freed outside the lock: hipFree synchronises the device)
    {
        std::lock_guard<std::mutex> g(h->ws_mu);
        if (h->ws_free.size() < 8) {
            h->ws_free.push_back(std::move(w));
            return;
        }
        // pool full: keep the LARGER workspaces.  (A pool that dropped the newcomer instead re-allocated the workspace of
        // every batch size beyond the first eight on every call: +1.4 ms per search, tools/route_grid.py.)
        size_t smallest = 0;
        for (size_t i = 1; i < h->ws_free.size(); i++)
            if (h->ws_free[i]->nq_cap < h->ws_free[smallest]->nq_cap) smallest = i;
        if (h->ws_free[smallest]->nq_cap < w->nq_cap) {
            drop = std::move(h->ws_free[smallest]);
            h->ws_free[smallest] = std::move(w);
        } else {
            drop = std::move(w);
        }
    }
}

struct ProfScope {
    Workspace *w;
    hipStream_t s;
    bool on;
    size_t idx = 0;
    ProfScope(Workspace *w_, hipStream_t s_, bool on_, int cls) : w(w_), s(s_), on(on_)
    {
        if (!on) return;
        if (w->ev_used == w->events.size()) {
            Event e;
            if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) {
                on = false;
                return;
            }
            w->events.push_back(e);
        }
        idx = w->ev_used++;
        w->events[idx].cls = cls;
        (void)hipEventRecord(w->events[idx].a, s);
    }
    ~ProfScope()
    {
        if (on) (void)hipEventRecord(w->events[idx].b, s);
    }
};

// What a search walks: all corpus rows (optionally testing the mask per row), or the compacted
// list of visible rows (rebuild_rowmap decides).
struct RowView {
    const uint8_t *mask;
    const uint32_t *rowmap;
    int64_t n;
};
static RowView row_view(const lb_gpu_index *h)
{
    if (!h->has_mask) return {nullptr, nullptr, h->n};
    if (h->rowmap_on) return {nullptr, h->d_rowmap, h->n_visible};
    return {h->d_mask, nullptr, h->n};
}

// First pass with a sampled threshold.  Instead of bootstrapping on the first few thousand rows and
// growing the chunks geometrically (3-4 launches + selects per search), `count` evenly spaced rows of
// the first `span` positions are scored, their m-th best entry becomes the admission threshold and
// the whole span is then walked ONCE.  The threshold is only a filter: every row below it is collected,
// so a list that ends with >= keep entries holds exactly the span's best `keep`.
//   too tight:  fewer than `keep` rows pass iff >= m sampled rows are among the span's best keep-1;
//               that count is ~Poisson(lambda = keep*count/span), and m = lambda + 5 sqrt(lambda) + 4
//               puts the tail below 1e-6 (m = 10 for k = 100, 14 for the 256 MFMA candidates at 1M rows, 27-41 for the
//               1024 candidates the fp16 route keeps beyond 1024 dimensions);
//   too loose:  about m*span/count rows pass (1.2k-1.8k at 1M rows), relative spread 1/sqrt(m); the span
//               is capped so that mean + 5 sigma stays below the list capacity.
// Either miss is detected (flag bit 2 / bit 0) and the query is redone by the classic bootstrap
// schedule, so results never depend on the sample.  The stride makes the estimate independent of the
// row order (sorted or clustered corpora included).  Fewer admitted rows also matter for speed: each
// admission is a returning atomic on one hot counter, ~12 ns apiece in the 1-query scan.
struct SamplePlan {
    bool on = false;
    int64_t span = 0;   // positions covered by the first pass
    uint32_t count = 0; // sampled rows
    int m = 0;
};
std::atomic<int> g_sample_tau{lb_tunable("LB_SAMPLE_TAU", 1)};
std::atomic<int> g_fused_fail_next{0}; // test hook: treat the next fused launch as one whose waits gave up
std::atomic<int> g_last_route{0}; // diagnostic build: kind * 10 + split of the last batched search's route
std::atomic<int> g_search_fail_next{0}; // test hook (diagnostic build): the next search on this process fails with LB_ERR_INTERNAL
// Add batches of at least this many bytes pin the caller's buffer instead of staging it (0 = never)
std::atomic<long long> g_add_register_min{(long long)lb_tunable("LB_ADD_REGISTER_MIN_MB", 64) << 20};
static SamplePlan sample_plan_for(int64_t n, int keep, uint32_t cap, uint32_t count)
{
    SamplePlan p;
    int m = 8;
    int64_t span = 0;
    for (int it = 0; it < 8; it++) {
        const double loose = (double)m * (1.0 + 5.0 / std::sqrt((double)m)); // mean + 5 sigma, in units of span/count
        const double span_max = (double)(cap - (uint32_t)keep) * (double)count / loose;
        span = span_max >= (double)n ? n : (int64_t)span_max;
        const double lambda = (double)keep * (double)count / (double)span;
        const int need = (int)std::ceil(lambda + 5.0 * std::sqrt(lambda) + 4.0);
        if (need <= m) break;
        m = need;
        if (m > 64) return p; // (sample_tau_kernel: m pops of a minimum, m <= 64)
    }
    if (span < 8 * (int64_t)count || !sample_tau_supported(count, m)) return p;
    p.on = true;
    p.span = span;
    p.count = count;
    p.m = m;
    return p;
}
static SamplePlan sample_plan(int64_t n, int keep, uint32_t cap, uint32_t count_max = 8192u)
{
    // (round 4: sampled thresholds from 16,384 rows -- was 65,536: below it the classic schedule ran three corpus launches and
    // three selects where the sampled one runs a sample, one pass and the finish: 40k x 768 at 64 queries 0.19 -> 0.10 ms)
    static const int64_t sample_min_rows = lb_tunable("LB_SAMPLE_MIN_ROWS", 16384);
    if (!g_sample_tau.load() || n < sample_min_rows || (uint32_t)keep >= cap) return SamplePlan{};
    // the largest sample whose threshold rank stays within the kernel's reach and whose span covers the view: a small view
    // (a selective filter) or a long candidate list (large k) takes a smaller sample -- a sample that is a large share of
    // the rows would need its several-hundredth smallest entry
    SamplePlan best;
    for (uint32_t count = std::min<uint32_t>(cap, count_max); count >= 1024u; count >>= 1) {
        const SamplePlan p = sample_plan_for(n, keep, cap, count);
        if (!p.on) continue;
        if (!best.on || p.span > best.span) best = p;
        if (p.span >= n) break;
    }
    return best;
}

// Exact scan of all rows for the query slots sel[0..nsel) (indices into d_q rows).
// mode 0: sampled first pass, 1: classic (bootstrap / growing chunks), 2: chunks that cannot overflow.
bool run_scan_path(lb_gpu_index *h, Workspace *w, hipStream_t s, const float *d_q, const int *d_sel,
                   int nsel, int k, int mode, float *d_dist, int64_t *d_lab, bool prof)
{
    const int metric = h->metric, order = h->order.load();
    const RowView rv = row_view(h);
    const int64_t n = rv.n;
    const uint8_t *mask = rv.mask;
    const int kkeep = std::max(k, 1);
    const bool safe = mode == 2;
    // (the latency path samples half as many rows: the sample launches are on its critical path, and the
    // extra admissions -- ~2k instead of ~1.2k at 1M rows -- are spread over the striped counters)
    static const uint32_t scan_count = (uint32_t)lb_tunable("LB_SCAN_SAMPLE", 4096);
    const SamplePlan sp = mode == 0 ? sample_plan(n, kkeep, w->cap, scan_count) : SamplePlan{};
    if (!sp.on) launch_init_cand(w->cs, d_sel, nsel, s);
    for (int g0 = 0; g0 < nsel; g0 += kScanMaxQ) {
        ctx_check(w->ctx);
        const int gn = std::min(kScanMaxQ, nsel - g0);
        const int *use_sel = d_sel + g0; // d_sel is always an explicit slot list here
        int64_t pos = 0;
        int step = 0;
        bool emitted = false;
        const EmitArgs em{k, h->has_ids ? h->d_ids : nullptr, d_dist, d_lab, w->h_flags};
        if (sp.on) {
            {   // sample scores (clear the flags; exact query norms ride along), threshold
                ProfScope p(w, s, prof, 1);
                launch_sample_scores(metric, order, h->d_X, h->dim, sp.span, sp.count, rv.rowmap, mask, d_q, use_sel,
                                     gn, w->cs, w->d_qna, s);
                launch_sample_tau(w->cs, use_sel, gn, sp.count, sp.m, /*zero_stripes=*/true, s);
            }
            {   // one pass over the span
                ProfScope p(w, s, prof, 3);
                launch_scan(metric, order, false, h->d_X, 0, sp.span, h->dim, d_q, use_sel, gn, w->d_qna, mask,
                            rv.rowmap, w->cs, /*boot=*/false, nullptr, 0, s, /*striped=*/true);
            }
            {
                ProfScope p(w, s, prof, 1);
                launch_select(w->cs, use_sel, gn, kkeep, 0u, s, false, (uint32_t)kkeep, sp.span >= n ? &em : nullptr,
                              /*striped=*/true);
            }
            pos = sp.span;
            step = 1;
            emitted = sp.span >= n;
        } else if (metric == LB_METRIC_COSINE) {
            ProfScope p(w, s, prof, 3);
            launch_query_norms(order, d_q, use_sel, gn, h->dim, w->d_qna, s);
        }
        while (pos < n) {
            ctx_check(w->ctx);
            const int64_t end = chunk_end_host(step, pos, n, kkeep, w->cap, safe, /*big_boot=*/true);
            const bool boot = step == 0;
            {
                ProfScope p(w, s, prof, 3);
                launch_scan(metric, order, false, h->d_X, pos, end, h->dim, d_q, use_sel, gn, w->d_qna,
                            mask, rv.rowmap, w->cs, boot, nullptr, 0, s);
            }
            {
                ProfScope p(w, s, prof, 1);
                launch_select(w->cs, use_sel, gn, kkeep, boot ? (uint32_t)(end - pos) : 0u, s, false, 0u,
                              end >= n ? &em : nullptr);
            }
            emitted = end >= n;
            pos = end;
            step++;
        }
        if (!emitted) // (only an empty corpus view gets here)
            launch_emit_lists(w->cs, use_sel, gn, k, h->has_ids ? h->d_ids : nullptr, d_dist, d_lab, w->h_flags, s);
    }
    return sp.on;
}

// download flags for slots [0,nq) and return those with any of `bits` set
// (on_host: the last kernel already wrote the subset's flags into the pinned h_flags)
int collect_flagged(Workspace *w, hipStream_t s, int nq, uint32_t bits, const int *h_subset,
                    int nsubset, std::vector<int> &out, bool on_host = false)
{
    if (!on_host)
        LB_HIP(hipMemcpyAsync(w->h_flags, w->cs.flags, (size_t)nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LB_HIP(hipStreamSynchronize(s));
    out.clear();
    if (h_subset) {
        for (int i = 0; i < nsubset; i++)
            if (w->h_flags[h_subset[i]] & bits) out.push_back(h_subset[i]);
    } else {
        for (int i = 0; i < nq; i++)
            if (w->h_flags[i] & bits) out.push_back(i);
    }
    return (int)out.size();
}

void upload_sel(Workspace *w, hipStream_t s, const std::vector<int> &sel)
{
    std::memcpy(w->h_qsel, sel.data(), sel.size() * sizeof(int));
    LB_HIP(hipMemcpyAsync(w->d_qsel, w->h_qsel, sel.size() * sizeof(int), hipMemcpyHostToDevice, s));
}

void scan_with_retry(lb_gpu_index *h, Workspace *w, hipStream_t s, const float *d_q, int nq_total,
                     const std::vector<int> &sel, int k, float *d_dist, int64_t *d_lab, bool prof)
{
    if (sel.empty()) return;
    bool identity = true; // "all queries": the prefilled list saves an upload on the latency path
    for (size_t i = 0; i < sel.size() && identity; i++) identity = sel[i] == (int)i;
    if (!identity) upload_sel(w, s, sel);
    const bool sampled = run_scan_path(h, w, s, d_q, identity ? w->d_iota : w->d_qsel, (int)sel.size(), k, 0,
                                       d_dist, d_lab, prof);
    std::vector<int> redo, over;
    if (collect_flagged(w, s, nq_total, 1u | 4u, sel.data(), (int)sel.size(), redo, true) > 0) {
        // the sampled threshold missed (or a list overflowed): classic schedule, then overflow-proof chunks
        if (sampled) {
            upload_sel(w, s, redo);
            run_scan_path(h, w, s, d_q, w->d_qsel, (int)redo.size(), k, 1, d_dist, d_lab, prof);
        }
        if (!sampled || collect_flagged(w, s, nq_total, 1u, redo.data(), (int)redo.size(), over, true) > 0) {
            if (!sampled) over = redo;
            upload_sel(w, s, over);
            run_scan_path(h, w, s, d_q, w->d_qsel, (int)over.size(), k, 2, d_dist, d_lab, prof);
            LB_HIP(hipStreamSynchronize(s));
        }
    }
}

// ---- which kernel generates the candidates of a batch ----------------------------------------------------------
// Every route is exact (re-rank + containment proof, else the exact scan redoes the query), so this is a cost choice.
//   NARROW32 / NARROW64  256 x 32 or 128 x 64 tile, operands split to bf16 in registers: one HBM-bound corpus pass per
//                        32 / 64 queries (kernels_gemm_narrow.hip)
//   TALL2                256 x 256 tile on the split contraction (kernels_gemm_tall2.hip): corpus image (split 1) or f32
//                        corpus split in registers (split 2)
//   TALL16 / NARROW16    ONE fp16 product per element (kernels_gemm_tall16.hip): 256 x 256 tiles, or one 64- / 128-query
//                        tile over the index's fp16 image (persistent workgroups, LDS-DMA ring)
//   WIDE                 128 x 128 tile (kernels_gemm.hip): f32 MFMA (split 0: the strict mode's route beyond 384
//                        queries), or the split contraction where the tall tiles cannot run
// Candidate modes (lb_gpu_index_set_candidate_mode):
//   LB_CAND_AUTO (default)    the cheapest route by the cost model below -- with the fp16 image that is the single-product
//                             route at every batch size, the split contraction where the data's range rules fp16 keys out
//   LB_CAND_F32_MFMA          as AUTO up to 384 queries, the f32-MFMA 128 x 128 tile beyond (rounds 1-2 default)
//   LB_CAND_SPLIT_BF16        corpus image for everything beyond the narrow tiles
//   LB_CAND_SPLIT_BF16_INREG  in-register split for everything beyond the narrow tiles
// Cost model: a pass over `n` positions of dimension D costs  n * (alpha * D + beta) [+ gamma]  per query tile, with
// the constants measured per kernel on MI355X over D in {128 .. 1536} x n in {100k .. 10M} (tools/route_grid.py; the
// GPU test test_route_choice_is_near_the_best_forced_route checks the choice against every forced route).
enum RouteKind { ROUTE_NARROW32 = 1, ROUTE_NARROW64 = 2, /* 3: the 256 x 128 split tile of rounds 2-3, removed */ ROUTE_WIDE = 4, ROUTE_TALL2 = 5, ROUTE_TALL16 = 6,
                 ROUTE_NARROW16 = 7 /* the fp16 route's 64- / 128-query tile over the fp16 copy: same pipeline as TALL16, reported apart */ };
struct Route {
    int kind = ROUTE_WIDE;
    int split = 0;
    double cost_ms = 0;
};
// One pass of a route's kernel over n positions of dimension D with `tiles` query tiles costs
//     ms = max(tiles * 1e-6 * n * (alpha * D + beta),  1e-6 * n * D * hbm)  +  1e-6 * n * D * first
// alpha: the contraction (per position, dimension and query tile); beta: the per-position work that does not scale with
// D (epilogue: key, admission test, side inputs); hbm: the corpus stream under that kernel (4 bytes per element at the
// rate the kernel's staging reaches); first: what the first query tile of a multi-tile pass waits for the corpus.
// Fitted to tools/route_grid.py on MI355X (D in {128, 384, 768, 1536} x n in {100k, 1M, 4M}), see LABNOTES.md 3.2.
struct RouteCost {
    double alpha, beta, hbm, first;
    double fixed = 0; // ms per search whatever the corpus: launches, thresholds, select, re-rank (fp16: + the query image, twice the
                      // candidates; the one-tile form over the fp16 copy: its re-rank is counted per query)
};
// (narrow tiles: one launch, the corpus tile is read from HBM once and re-used from L2 by the other query tiles)
constexpr RouteCost kCostNarrow32{0.000323, 0.0247, 0.000640, 0.000300, 0.09};   // per 32-query tile: 1.09 ms at 4M x 768, 0.27 at 4M x 128
constexpr RouteCost kCostNarrow64{0.000527, 0.0225, 0.000640, 0.000200, 0.09};   // per 64-query tile: 1.71 ms at 4M x 768, 0.36 at 4M x 128
constexpr RouteCost kCostTall2Inreg{0.001260, 0.1300, 0.000640, 0.000250, 0.09}; // per 256-query tile: 3.9 ms at 4M x 768, 1.16 at 4M x 128
constexpr RouteCost kCostTall2Image{0.001150, 0.1300, 0.000640, 0.000220, 0.09};
constexpr RouteCost kCostWideF32{0.001940, 0.0600, 0.000640, 0.0, 0.09};
constexpr RouteCost kCostTall16{0.000540, 0.0800, 0.000660, 0.000140, 0.19}; // per 256-query tile, one fp16 product, persistent form: 0.49 ms per
                                                                             // tile at 1M x 768, a single tile streams the corpus at 6 TB/s
constexpr RouteCost kCostTall16Img{0.000460, 0.0500, 0.000330, 0.000040, 0.11}; // the same from the corpus's fp16 image: 0.40 ms per tile at
                                                                                // 1M x 768, never bound by the stream (1.5 GB)
constexpr RouteCost kCostNarrow16{0.000100, 0.0300, 0.000340, 0.000030, 0.06};  // up to 64 queries over the fp16 copy: its HBM stream (6 TB/s)
inline double route_ms(const RouteCost &c, int64_t n, int D, int tiles)
{
    const double nd = 1e-6 * (double)n;
    const double compute = (double)tiles * nd * (c.alpha * (double)D + c.beta);
    const double stream = nd * (double)D * c.hbm;
    return (compute > stream ? compute : stream) + nd * (double)D * c.first + c.fixed;
}

static Route choose_route(int nq, int64_t n, int D, int cmode, bool narrow_ok, bool have_image, bool f16_ok, bool have_f16_image = false)
{
    static const int narrow_max = lb_tunable("LB_NARROW_MAXQ", 384);
    static const bool nsplit_on = lb_tunable("LB_NARROW_SPLIT", 1) != 0;
    const int tiles32 = (nq + 31) / 32, tiles64 = (nq + 63) / 64, tiles128 = (nq + 127) / 128, tiles256 = (nq + 255) / 256;
    // (tall tiles only with enough of them to fill the chip a few times over: 512 workgroups run at once, and at 125k
    // visible rows x 256 queries the 978 tall tiles came out 5 % behind the 3908 smaller ones)
    const bool tall2_fills = (n / 256) * tiles256 >= 1024; // (one workgroup per CU: 256 run at once)
    Route cand[8];
    int nc = 0;
    auto add = [&](int kind, int split, double ms) { cand[nc].kind = kind; cand[nc].split = split; cand[nc].cost_ms = ms; nc++; };
    const bool image = cmode == LB_CAND_SPLIT_BF16 && have_image;
    if (narrow_ok && !image && (cmode == LB_CAND_AUTO || nq <= narrow_max)) {
        const int nsp = nsplit_on ? 2 : 0;
        if (nq <= 32 || tiles32 <= 10) add(ROUTE_NARROW32, nsp, route_ms(kCostNarrow32, n, D, tiles32));
        if (nq > 32) add(ROUTE_NARROW64, nsp, route_ms(kCostNarrow64, n, D, tiles64));
    }
    // (round 4: the 256 x 128 tile -- ROUTE_TALL, kernels_gemm_tall.hip -- is gone: within 2-5 % of the 64-query narrow tile and of
    // this one wherever it was picked, profiles/r03_route_grid.txt, and never picked once the fp16 route is on offer)
    if (narrow_ok) {
        if (image) {
            add(ROUTE_TALL2, 1, route_ms(kCostTall2Image, n, D, tiles256));
        } else if (nsplit_on && (cmode != LB_CAND_F32_MFMA || nq <= narrow_max)) {
            const bool forced = cmode == LB_CAND_SPLIT_BF16_INREG;
            if (nq > 128 && (tall2_fills || forced)) add(ROUTE_TALL2, 2, route_ms(kCostTall2Inreg, n, D, tiles256));
        }
    }
    // one fp16 product instead of three bf16 ones (split code 3): AUTO and the explicit LB_CAND_F16, while the corpus norms
    // allow it (f16_ok) and there are enough tiles to fill the chip
    static const int f16_on = lb_tunable("LB_F16", 1);
    // (a filtered view of a corpus that has its fp16 image: from the size a sampled threshold exists for, the cost model
    // decides -- round 3's gate of 128 Mi elements kept 100k x 768 views on the 64-query split tile: 0.21 / 0.30 / 0.50 ms at
    // 128 / 256 / 512 queries where the image serves them in 0.16 / 0.20 / 0.28)
    static const int64_t f16_view_min = lb_tunable("LB_F16_VIEW_MIN", 16384);
    // (over the fp16 copy the route needs neither dim % 32 == 0 nor aligned queries: both images are zero-padded planes)
    if ((narrow_ok || have_f16_image) && f16_ok && f16_on && !image && (nq > 32 || have_f16_image) &&
        (cmode == LB_CAND_F16 || (cmode == LB_CAND_AUTO && (n >= 262144 || (have_f16_image && n >= f16_view_min)))))
    { // (below: launch overheads decide, and the narrow tiles win; a filtered view of a large corpus counts by its elements)
        if (have_f16_image && nq <= 128) { // (one query tile: the 64- / 128-query form of the persistent kernel)
            // + what the route pays per query beside the stream: twice (beyond 1024 dimensions four times) the candidates to
            // select and re-rank
            // (round 4: 0.0016 -> 0.0008 beyond 1024 dimensions -- the finish launch re-ranks ~270 rows per query where select +
            // re-rank scored 1024; with 0.0016 a row list of 125k x 1536 went to the 64-query split tile at exactly 64 and 128
            // queries: 0.30 / 0.40 ms where the image serves them in 0.24 / 0.31)
            const double q1 = D > 1024 ? 0.0008 : 0.0003;
            add(ROUTE_NARROW16, 3, route_ms(kCostNarrow16, n, D, 1) + q1 * nq);
        }
        else add(ROUTE_TALL16, 3, route_ms(have_f16_image ? kCostTall16Img : kCostTall16, n, D, tiles256));
    }
    if (cmode == LB_CAND_F32_MFMA || cmode == LB_CAND_AUTO || nc == 0) add(ROUTE_WIDE, 0, route_ms(kCostWideF32, n, D, tiles128));
#ifdef LB_DIAG
    { // A/B (tools/route_grid.py): force a route when it is available for this batch (read per call: the tool flips it)
        if (getenv("LB_TRACE_ROUTE")) {
            fprintf(stderr, "[route] nq %d n %lld D %d cmode %d narrow_ok %d f16_ok %d image %d:", nq, (long long)n, D, cmode, (int)narrow_ok, (int)f16_ok, (int)have_f16_image);
            for (int i = 0; i < nc; i++) fprintf(stderr, " kind %d/%d %.3f ms", cand[i].kind, cand[i].split, cand[i].cost_ms);
            fprintf(stderr, "\n");
        }
        const char *e = getenv("LB_FORCE_ROUTE");
        const int force = e ? atoi(e) : 0;
        for (int i = 0; i < nc; i++)
            if (cand[i].kind == force) return cand[i];
    }
#endif
    if (cmode == LB_CAND_F32_MFMA && nq > narrow_max) return cand[nc - 1]; // strict mode: the f32 tile beyond the narrow range
    if (cmode == LB_CAND_F16)
        for (int i = 0; i < nc; i++)
            if (cand[i].kind == ROUTE_TALL16 || cand[i].kind == ROUTE_NARROW16) return cand[i];
    if (nq <= 32 && nc > 0 && cand[0].kind == ROUTE_NARROW32) {
        // one pass of the 32-query tile (sample and thresholds inside the launch): 0.09 ms + 0.47 ms per 1M x 768 whatever the
        // model above says about tile counts -- nothing is cheaper but the stream of the fp16 copy
        cand[0].cost_ms = 0.09 + 1e-6 * (double)n * (0.014 + 0.000612 * (double)D);
        for (int i = 1; i < nc; i++)
            if (cand[i].kind == ROUTE_NARROW16 && cand[i].cost_ms < cand[0].cost_ms) return cand[i];
        return cand[0];
    }
    int best = 0;
    for (int i = 1; i < nc; i++)
        if (cand[i].cost_ms < cand[best].cost_ms) best = i;
    return cand[best];
}

int search_batch_device(lb_gpu_index *h, Workspace *w, hipStream_t s, int nq, const float *d_q, int k,
                        float *d_dist, int64_t *d_lab, int kc_in, bool prof, int64_t &fallbacks, bool allow_f16 = true, int attempt = 0)
{
    const int metric = h->metric, order = h->order.load();
    const RowView rv = row_view(h);
    const int64_t n = rv.n; // positions to walk: corpus rows, or the visible-row list under a selective filter
    const uint8_t *mask = rv.mask;
    ctx_check(w->ctx);
    ProfScope whole(w, s, prof, 4);

    // Path selection: <= 4 queries exact scan (0.50-0.55 ms at 1M x 768); from 5 queries a candidate route picked by
    // choose_route's cost model (narrow / tall / tall2 / fp16 / f32 tile), exact re-rank behind every one of them.
    const bool narrow_ok = h->dim % 32 == 0 && ((reinterpret_cast<uintptr_t>(d_q) & 15) == 0);
    static const int narrow_min = lb_tunable("LB_NARROW_MINQ", 5);
    static const int narrow_max = lb_tunable("LB_NARROW_MAXQ", 384);
    // rows with inf / NaN components: the MFMA pipeline's keys and error bounds assume finite data;
    // the scan path orders non-finite distances canonically (NaN last)
    const int cmode = h->cand_mode.load();
    // (the fp16 image serves unfiltered searches and searches over a row list -- the persistent kernels gather out of it,
    // dimensions from 256; under a per-row mask test the kernel stages f32 rows)
    const bool have_xh = h->d_Xh != nullptr && h->xh_rows == h->n && !mask && (!rv.rowmap || h->dim >= 256) &&
                         // (a centred image -- L2 -- is only ever read by the persistent kernels: nothing else knows the centre)
                         (!h->xh_centred || tall16_runs_persistent(h->dim, nq, true, rv.rowmap != nullptr, false));
    const bool centred = have_xh && h->xh_centred; // L2 keys about the image's centre (sync_f16_image)
    // within the fp16 contraction's range: the centred norms when the image is centred, the rows' own norms otherwise
    bool f16_range_ok = centred ? h->xh_c_ok : h->f16_ok;
    // Mid-size corpora and views (below 262,144 rows: offered the fp16 routes since round 4): only while ONE sampled span covers
    // the view with the candidate list the fp16 keys need (twice / four times the split tiles') -- with k = 300 that list is a
    // quarter of the lists' capacity, the sampled span ends short of 200k rows and the rest runs the classic schedule:
    // 0.45 ms where the split tiles over the f32 rows take 0.15.  (Larger corpora: as before.)
    if (cmode == LB_CAND_AUTO && n < 262144 && kc_in > 0) {
        const int kc16 = std::min(std::min(kc_in * (h->dim > 1024 ? 4 : 2), std::max(1024, 2 * kc_in)), (int)(w->cap / 4));
        const SamplePlan sp16 = sample_plan(n, kc16, w->cap);
        if (!(sp16.on && sp16.span >= n)) f16_range_ok = false;
    }
    // 1 .. 4 queries: the exact scan streams the f32 corpus (0.52 ms per 1M x 768); with the fp16 copy the candidate pass
    // streams half the bytes and the exact re-rank of 512 candidates costs 0.03 ms -- taken when the model says it is cheaper
    bool small_on_copy = false;
    if (!h->nonfinite && nq < narrow_min && have_xh && f16_range_ok && allow_f16 &&
        (cmode == LB_CAND_F16 || (cmode == LB_CAND_AUTO && h->f16_skip.load(std::memory_order_relaxed) == 0))) {
        const double scan_ms = 0.04 + 1e-6 * (double)n * ((double)h->dim * 0.00066 + 0.04); // (0.04: sample, thresholds, select + emit)
        const double copy_ms = route_ms(kCostNarrow16, n, h->dim, 1) + (h->dim > 1024 ? 0.0008 : 0.0003) * nq;
        small_on_copy = cmode == LB_CAND_F16 || copy_ms < scan_ms;
    }
    // (dimensions that are not multiples of 32: the MFMA tiles over f32 rows do not apply; the fp16 copy does)
    const bool copy_route_ok = have_xh && f16_range_ok && allow_f16 &&
                               (cmode == LB_CAND_F16 || (cmode == LB_CAND_AUTO && h->f16_skip.load(std::memory_order_relaxed) == 0));
    if (h->nonfinite || (nq < ((narrow_ok || copy_route_ok) ? narrow_min : kGemmMinQ) && !small_on_copy)) {
        h->last_route.store(0, std::memory_order_relaxed);
        std::vector<int> all(nq);
        for (int i = 0; i < nq; i++) all[i] = i;
        scan_with_retry(h, w, s, d_q, nq, all, k, d_dist, d_lab, prof);
        return LB_OK;
    }

    // ---- batched path: MFMA candidate generation + exact re-rank --------------------
    const bool have_image = h->d_Xs != nullptr && h->xs_rows == h->n && h->dim % 32 == 0;
    bool f16_offer = f16_range_ok && allow_f16;
    if (f16_offer && cmode == LB_CAND_AUTO && h->f16_skip.load(std::memory_order_relaxed) > 0) {
        h->f16_skip.fetch_sub(1, std::memory_order_relaxed);
        f16_offer = false;
    }
    // (offset-dominated L2 data: only the centred image's keys resolve anything; dot product over rows of very different
    // lengths: only the lower-bound keys do)
    const bool keys_matter = (centred && h->xh_offset_dom) || (metric == LB_METRIC_DOT && h->norm_spread);
    const int cmode_route = (cmode == LB_CAND_AUTO && keys_matter && f16_offer) ? LB_CAND_F16 : cmode;
    const Route route = choose_route(nq, n, h->dim, cmode_route, narrow_ok, have_image, f16_offer, have_xh);
    h->last_route.store(route.kind * 10 + route.split, std::memory_order_relaxed);
#ifdef LB_DIAG
    g_last_route.store(route.kind * 10 + route.split);
#endif
    // candidates kept per query.  The fp16 single-product route keeps twice as many: its keys are good to ~1.1e-3 of |q||x|,
    // and the containment proof needs the gap between the k-th and the LAST kept candidate to exceed about 2.5x that --
    // on the benchmark data the gap to the 256th is 0.0019-0.0028 (most queries would fail), to the 512th 0.0039-0.0045.
    // (Uniform random data is the hard case: its distances concentrate like 1/sqrt(D), so the gap shrinks with the dimension
    // while the bound does not -- beyond 1024 dimensions four times as many candidates are kept.)
    static const int f16_kc_mult = lb_tunable("LB_F16_KC_MULT", 0);
    if (attempt == 0 && h->kc_hint_left.load(std::memory_order_relaxed) > 0) { // (see the retries at the end)
        h->kc_hint_left.fetch_sub(1, std::memory_order_relaxed);
        kc_in = std::max(kc_in, std::min(h->kc_hint.load(std::memory_order_relaxed), (int)(w->cap / 4)));
    }
    int kc = kc_in;
    // (since the finish launch prunes in key space, kc only sizes the admission threshold: the list must hold the rows within
    // the error bound of the k-th key -- 2-4 k of them with fp16 keys -- not a fixed number of rows to re-rank; capped so that
    // a sampled threshold of that depth still exists)
    if (route.split == 3)
        kc = std::min(std::min(kc_in * (f16_kc_mult > 0 ? f16_kc_mult : (h->dim > 1024 ? 4 : 2)), std::max(1024, 2 * kc_in)), (int)(w->cap / 4));
    const SamplePlan sp = sample_plan(n, kc, w->cap);
    // up to 8 queries the sample is scored by the wave-per-row kernel (candidate keys; 22-28 us against
    // 44 us for 8192 rows through the 32-workgroup MFMA launch); larger batches sample through the MFMA
    // kernel itself.  Up to 64 queries the exact query norms ride in the threshold launch.
    // candidate contraction: exact f32 MFMA, or 3 x bf16 MFMA on split operands (choose_route)
    const int split = route.split;                 // of the operands handed to the kernel: 0 f32, 1 images, 2 f32 split in registers
    const bool use_narrow = route.kind == ROUTE_NARROW32 || route.kind == ROUTE_NARROW64;
    const bool tile64 = route.kind == ROUTE_NARROW64;
    const bool nsplit = use_narrow && route.split == 2;
    const bool use_tall = route.kind == ROUTE_TALL2 || route.kind == ROUTE_TALL16 || route.kind == ROUTE_NARROW16;
    const bool use_tall2 = route.kind == ROUTE_TALL2;
    const bool use_tall16 = route.kind == ROUTE_TALL16 || route.kind == ROUTE_NARROW16; // (the launcher takes the 64-query tile by itself)
    const int wsplit = use_narrow ? 0 : route.split;
    const float *gx = h->d_X, *gq = d_q;
    const float u24 = 5.9604645e-8f;
    // rounding-error bound of the candidate inner products, per unit of ||q|| ||x||: a k-ordered f32 fma chain of
    // length D, or (split contraction) 3D/16 MFMA accumulations + <=16-term block sums plus the dropped lo*lo /
    // residual terms
    // ... or (split 3) one fp16 product: 2^-11 per operand, the subnormal term under the route's norm conditions, and the
    // f32 accumulation of D products (kernels_gemm_tall16.hip)
    // ... with the index's fp16 image the operands' share is MEASURED instead: |q.x - q~.x~| <= |q||x - x~| + |q - q~||x~|, the
    // residual norms taken when the image was written (rho_x = max over rows of |x - x~| / |x|, sync_f16_image) and when the
    // query image is (rho_q per query, query_prep_body) -- a third to a half of the worst case for data that fills the
    // mantissa, subnormal effects included:  gamma(q) = 1.05 (rho_x + A (1 + rho_x) + 2^-21) + 1.05 (1 + rho_x)(1 + A) rho_q,
    // A = (D + 8) 2^-24 the f32 accumulation of D exact products
    static const bool rho_on = lb_tunable("LB_MEASURED_RHO", 1) != 0;
    const bool measured = rho_on && route.split == 3 && have_xh && h->xh_rho > 0.f && h->xh_rho < 4.0e-4f; // (else the worst case is the better bound)
    const float accA = (float)(h->dim + 8) * u24;
    const float qrho_k = measured ? 1.05f * (1.0f + h->xh_rho) * (1.0f + accA) : 0.f;
    const float gamma = route.split == 0   ? 1.05f * (float)(h->dim + 8) * u24
                        : measured         ? 1.05f * (h->xh_rho + accA * (1.0f + h->xh_rho) + 4.7683716e-7f)
                        : route.split == 3 ? 1.05f * (9.765625e-4f + 4.7683716e-7f + (float)(h->dim + 8) * u24 +
                                                      2.9802322e-8f * std::sqrt((float)h->dim) * 65.0f)
                                           : 1.05f * ((float)(3 * h->dim / 16 + 24) * u24 + 3.0f * 3.8146973e-6f);
    auto split_queries = [&]() { // hi / lo bf16 image of the batch (same bytes as the f32 rows)
        const size_t need = (size_t)nq * h->dim * sizeof(float);
        if (w->d_qs_bytes < need) {
            if (w->d_qs) (void)hipFree(w->d_qs);
            w->d_qs = nullptr;
            w->d_qs_bytes = 0;
            LB_HIP(hipMalloc(&w->d_qs, need));
            w->d_qs_bytes = need;
        }
        launch_split_bf16(d_q, w->d_qs, nq, h->dim, s);
    };
    if (!use_narrow && (route.split == 1 || route.split == 2)) split_queries(); // the tall / wide split kernels take the batch as an image
    float *d_qinv = nullptr, *d_qnrm = nullptr, *d_qrho = nullptr;
    // dot product on the persistent fp16 kernels: LOWER-BOUND keys -(q.x)~ / G - |x|, G = (gamma_a + gamma_o) |q| (a few very long
    // rows then sort to the front of the lists and are scored exactly instead of widening every row's error bound)
    const float gsum = gamma + 1.05f * (float)(h->dim + 8) * u24;
    const float rho_gain = (measured && gsum > 0.f) ? qrho_k / gsum : 0.f; // (dot: G(q) = (gsum + qrho_k rho_q) |q|, folded into qnrm)
    const bool dot_lb = metric == LB_METRIC_DOT && use_tall16 &&
                        tall16_runs_persistent(h->dim, nq, have_xh, rv.rowmap != nullptr, mask != nullptr);
    const bool own_keys = centred || dot_lb; // keys only the persistent kernels produce: sample and boot chunks must come from them
    static const int riders_max = lb_tunable("LB_NORM_RIDERS_MAXQ", 384);
    const bool prep_riders = sp.on && nq <= riders_max; // (cosine: the exact query norms come out of the threshold launch)
    if (use_tall16) { // fp16 image of the batch (scaled per query) + the inverse scales
        const size_t img = (((size_t)nq * (size_t)((h->dim + 31) & ~31) * 2) + 255) & ~(size_t)255, need = img + 3 * (size_t)nq * sizeof(float);
        if (w->d_qh_bytes < need) {
            if (w->d_qh) (void)hipFree(w->d_qh);
            w->d_qh = nullptr;
            w->d_qh_bytes = 0;
            LB_HIP(hipMalloc(&w->d_qh, need));
            w->d_qh_bytes = need;
        }
        d_qinv = reinterpret_cast<float *>(static_cast<char *>(w->d_qh) + img);
        d_qnrm = d_qinv + nq;
        d_qrho = measured ? d_qnrm + nq : nullptr;
    }
    if (!use_narrow && route.split == 1) {
        gx = h->d_Xs;
        gq = w->d_qs;
    }
    // Up to 64 queries on the narrow split tiles the sample and its thresholds ride INSIDE the candidate launch (FUSED
    // in kernels_gemm_narrow.hip: 19-34 us of sample + 13 us of threshold kernel off the critical path).
    static const int fused_max = lb_tunable("LB_FUSED_SAMPLE_MAXQ", 32); // (33-64 queries, the 64-query tile: measured level)
    // (from 131,072 rows: the fused launch has a floor of ~115 us whatever the corpus -- 70k x 768 at 8 queries 0.156 ms fused,
    // 0.124 with the sample and the thresholds as launches of their own, level at 150k-300k, 20 us ahead at 1M -- and on the
    // 16k-64k-row corpora that take a sampled threshold since round 4 its waits gave up: 17k rows at 16 queries, 40k at 32,
    // the batch redone exactly in 1 ms)
    // (and thresholds of rank up to 32: with k = 300 -- 1024 candidates, m = 48 -- the threshold workgroups outlast the waits of
    // the corpus workgroups: 200k x 768 at 32 queries gave up on every search, 1.5 ms)
    const bool fused = sp.on && use_narrow && nsplit && nq <= fused_max && nq <= 64 && n >= 131072 && sp.m <= 32;
    // a search over a row list on the persistent fp16 kernels: its candidate entries carry positions of the list
    const bool entries_pos = use_tall16 && rv.rowmap != nullptr &&
                             tall16_entries_are_positions(h->dim, nq, have_xh, true, mask != nullptr);
    // over the fp16 copy the sample goes through the persistent kernel itself: 512 granules of 16 consecutive rows, evenly
    // spaced over the span (whole KiB of the K-blocked image; every workgroup takes a share of them)
    static const bool granule_on = lb_tunable("LB_GRANULE_SAMPLE", 1) != 0;
    // (up to 8 queries the wave-per-row kernel over the f32 rows is 5 us quicker: every load of a row in flight at once)
    static const int light_max = lb_tunable("LB_LIGHT_SAMPLE_MAXQ", 8); // (also over a row list: at 32 queries the wave-per-row
                                                                        // kernel took 105 us against the granule sample's 40)
    // (dot product's lower-bound keys: the sample must come out of the same kernel; centred L2 keys the wave-per-row kernel
    // computes too, from the f32 rows about the same centre)
    const bool granule_sample = sp.on && use_tall16 && sp.count % 16 == 0 && (rv.rowmap == nullptr || entries_pos) &&
                                ((have_xh && granule_on && nq > light_max) || dot_lb || (centred && nq > light_max));
    // (not in front of the split tiles: up to 32 queries they always ran fused, and the wave-per-row sample was never paired
    // with them -- L2, k = 300, 8 queries over 50k rows: every query flagged and scanned)
    const bool light_sample = sp.on && !fused && !granule_sample && nq <= light_max && !use_narrow;
    // Up to 128 queries on the one-tile kernel over the image, one span: the candidate launch turns the sample into the
    // thresholds ITSELF (its first nq workgroups, on shorter row ranges; kernels_gemm_tall16.hip, TAUIN) -- no threshold launch
    // and no gap behind it in front of the pass.
    static const bool tauin_on = lb_tunable("LB_TAUIN", 1) != 0;
    const bool tauin = tauin_on && sp.on && sp.span >= n && use_tall16 && route.kind == ROUTE_NARROW16 && !fused &&
                       (light_sample || granule_sample) && prep_riders &&
                       tall16_tin_ok(h->dim, nq, sp.span, have_xh, rv.rowmap != nullptr, mask != nullptr, (prep_riders && metric == LB_METRIC_COSINE), sp.count, sp.m);
    Tall16Tin tin{};
    uint32_t tin_epoch = 0;
    if (tauin) {
        if (++w->fs_epoch == 0) w->fs_epoch = 1; // (0 = "never published")
        tin_epoch = w->fs_epoch;
        tin.count = sp.count;
        tin.m = sp.m;
        tin.tag = tin_epoch;
        tin.fail_host = w->h_fail;
        tin.Q = d_q;
        tin.qna = (prep_riders && metric == LB_METRIC_COSINE) ? w->d_qna : nullptr;
        tin.order = order;
    }
    // (up to 8 queries: the query preparation rides in the sample launch -- one launch and one gap less in front of the pass)
    const bool prep_rides = use_tall16 && light_sample && prep_riders;
    if (use_tall16 && !prep_rides)
        // one launch: the image, the scales, the exact query norms (cosine) and the reset of the candidate state
        // (the exact norm is a serial chain of D additions, 3.5 us at 768: up to 384 queries it rides in the threshold launch
        // instead, where nothing waits for it)
        launch_query_prep(d_q, nq, h->dim, w->d_qh, d_qinv, (metric == LB_METRIC_COSINE && !prep_riders) ? w->d_qna : nullptr, order,
                          w->cs, s, centred ? h->d_center : nullptr, dot_lb ? d_qnrm : nullptr, tauin, d_qrho, rho_gain);
    const bool norm_riders = prep_riders && metric == LB_METRIC_COSINE;
    // the last launch: key-space pruning + exact re-rank + proof in one (kernels_finish.hip); beta: how far beyond one error
    // bound the cut lies (the proof itself never depends on it)
    static const float finish_beta = 0.01f * (float)lb_tunable("LB_FINISH_BETA_PCT", 25);
    if (!light_sample && !fused && !use_tall16) launch_init_cand(w->cs, nullptr, nq, s);
    if (metric == LB_METRIC_COSINE && !norm_riders && !use_tall16) launch_query_norms(order, d_q, nullptr, nq, h->dim, w->d_qna, s); // (fp16 route: query_prep)
    static const bool sample_narrow = lb_tunable("LB_TALL_SAMPLE_NARROW", 1) != 0;
    // (fp16 route, several 256-query tiles: the sample goes through the fp16 kernel itself -- 32 row tiles x nq/256 workgroups
    // against nq/64 x 64 of the narrow tile; 1024 queries: 1.94 -> 1.88 ms, 512: 1.015 -> 1.00; tools/probe/sample_route_probe.py)
    static const int sample_narrow_maxq = lb_tunable("LB_TALL_SAMPLE_NARROW_MAXQ", 384);
    auto candidates = [&](int64_t b, int64_t e, const uint32_t *rowmap, bool boot) {
        if (w->ctx && !boot) LB_HIP(hipStreamSynchronize(s)); // a cancellable call waits for the work in front of every corpus pass
        ctx_check(w->ctx);
        ProfScope p(w, s, prof, 0);
        if (use_narrow)
            launch_gemm_filter_narrow(metric, gx, h->d_norm2, h->d_rnorm, b, e, h->dim, gq, nq, mask, rowmap,
                                      w->cs, boot, s, tile64, nsplit);
        else if (use_tall16 && h->dim % 32 != 0 && (rowmap || mask) && !entries_pos)
            // the sample of a search over the fp16 copy when the dimension is not a multiple of 32: the f32 tile takes any
            launch_gemm_filter(metric, h->d_X, h->d_norm2, h->d_rnorm, b, e, h->dim, d_q, nq, mask, rowmap, w->cs, boot, 0, s);
        else if (use_tall && boot && (wsplit == 2 || wsplit == 3) && sample_narrow && narrow_ok && !entries_pos && !own_keys &&
                 !(use_tall16 && nq > sample_narrow_maxq))
            // the 8192-row sample of a tall-tile search: the 64-query tile of the narrow kernel (same contraction, f32
            // operands) gets through its 24 K-steps of 32 in 31-35 us, the tall tile through its 48 of 16 in 57
            // (narrow_ok: that kernel reads the f32 queries in 16-B pieces -- a batch pointer that is not 16-B aligned stays
            // on the fp16 kernel below, which reads its own image of the batch)
            launch_gemm_filter_narrow(metric, h->d_X, h->d_norm2, h->d_rnorm, b, e, h->dim, d_q, nq, mask, rowmap, w->cs,
                                      true, s, /*tile64=*/true, /*split=*/true);
        else if (use_tall16)
            launch_gemm_filter_tall16(metric, h->d_X, centred ? h->d_norm2c : h->d_norm2, h->d_rnorm, b, e, h->dim, w->d_qh, d_qinv, nq,
                                      mask, rowmap, w->cs, boot, s, have_xh ? h->d_Xh : nullptr, h->xh_cap, 0u, d_qnrm, gsum,
                                      (tauin && !boot && b == 0 && e == sp.span) ? &tin : nullptr);
        else if (use_tall2)
            launch_gemm_filter_tall2(metric, gx, h->d_norm2, h->d_rnorm, b, e, h->dim, w->d_qs, nq, mask, rowmap, w->cs,
                                     boot, wsplit, s);
        else
            launch_gemm_filter(metric, gx, h->d_norm2, h->d_rnorm, b, e, h->dim, gq, nq, mask, rowmap, w->cs,
                               boot, wsplit, s);
    };
    int64_t pos = 0;
    int step = 0;
    uint32_t fused_epoch = tin_epoch; // (a wait inside a launch that gave up is reported through one pinned word, whichever launch it was)
    if (sp.on) { // sampled threshold, then one pass over the span (see sample_plan)
        const uint32_t *smap = w->d_smap;
        if (light_sample) {
            ProfScope p(w, s, prof, 1);
            // (centred keys: the same key about the image's centre, from the f32 rows)
            const SamplePrep sprep{w->d_qh, d_qinv, dot_lb ? d_qnrm : nullptr, centred ? h->d_center : nullptr, tauin, d_qrho, rho_gain};
            launch_sample_scores(metric, order, h->d_X, h->dim, sp.span, sp.count, rv.rowmap, mask, d_q, nullptr, nq,
                                 w->cs, nullptr, s, centred ? h->d_norm2c : h->d_norm2, h->d_rnorm, centred ? h->d_center : nullptr,
                                 prep_rides ? &sprep : nullptr);
            if (!tauin)
                launch_sample_tau(w->cs, nullptr, nq, sp.count, sp.m, false, s, d_q, h->dim,
                                  norm_riders ? w->d_qna : nullptr, order);
        } else if (granule_sample) {
            {
                ctx_check(w->ctx);
                ProfScope p(w, s, prof, 1); // (timing class "select": threshold work, so that class "gemm" is the corpus pass alone)
                launch_gemm_filter_tall16(metric, h->d_X, centred ? h->d_norm2c : h->d_norm2, h->d_rnorm, 0, sp.count, h->dim, w->d_qh, d_qinv,
                                          nq, nullptr, rv.rowmap, w->cs, /*boot=*/true, s, have_xh ? h->d_Xh : nullptr, h->xh_cap,
                                          (uint32_t)(sp.span / (int64_t)(sp.count / 16)), d_qnrm, gsum);
            }
            ProfScope p(w, s, prof, 1);
            if (!tauin) launch_sample_tau(w->cs, nullptr, nq, sp.count, sp.m, false, s, d_q, h->dim, norm_riders ? w->d_qna : nullptr, order);
        } else {
            {
                std::lock_guard<std::mutex> g(h->smap_mu);
                if (!h->smap_valid) {
                    if (h->smap_cap < sp.count) {
                        if (h->d_smap) (void)hipFree(h->d_smap);
                        h->d_smap = nullptr;
                        h->smap_cap = 0;
                        LB_HIP(hipMalloc(&h->d_smap, (size_t)sp.count * sizeof(uint32_t)));
                        h->smap_cap = sp.count;
                    }
                    launch_sample_map(rv.rowmap, sp.span, sp.count, h->d_smap, s);
                    LB_HIP(hipStreamSynchronize(s));
                    h->smap_span = sp.span;
                    h->smap_count = sp.count;
                    h->smap_valid = true;
                }
                if (h->smap_span == sp.span && h->smap_count == sp.count) smap = h->d_smap;
            }
            if (smap == w->d_smap) launch_sample_map(rv.rowmap, sp.span, sp.count, w->d_smap, s);
            if (!fused) {
                candidates(0, sp.count, smap, /*boot=*/true);
                ProfScope p(w, s, prof, 1);
                launch_sample_tau(w->cs, nullptr, nq, sp.count, sp.m, false, s, d_q, h->dim,
                                  norm_riders ? w->d_qna : nullptr, order);
            }
        }
        if (fused) {
            ProfScope p(w, s, prof, 0);
            FusedSample fs;
            fs.smap = smap;
            fs.count = sp.count;
            fs.m = sp.m;
            fs.ticket = w->d_fsync;
            fs.ticket_base = w->fs_base;
            fs.ready = w->d_fsync + 1;
            if (++w->fs_epoch == 0) w->fs_epoch = 1; // (0 = "never published")
            fs.epoch = w->fs_epoch;
            fs.qna = norm_riders ? w->d_qna : nullptr;
            fs.order = order;
            fs.fail_host = w->h_fail;
            fs.relaxed = lb_tunable("LB_FUSED_RELAXED", 0);
            launch_gemm_filter_narrow_fused(metric, gx, h->d_norm2, h->d_rnorm, 0, sp.span, h->dim, gq, nq, mask, rv.rowmap,
                                            w->cs, s, tile64, fs);
            w->fs_base += fused_sample_blocks(sp.count, nq, tile64);
            fused_epoch = fs.epoch;
        } else {
            candidates(0, sp.span, rv.rowmap, /*boot=*/false);
        }
        // (one span covers the view: the finish launch prunes the raw list itself -- everything below tau is in it)
        if (sp.span < n) {
            ProfScope p(w, s, prof, 1);
            launch_select(w->cs, nullptr, nq, kc, 0u, s, false, (uint32_t)kc, nullptr, false, /*unsorted=*/true);
        }
        pos = sp.span;
        step = 1;
    }
    while (pos < n) {
        const int64_t end = chunk_end_host(step, pos, n, kc, w->cap, false);
        const bool boot = step == 0;
        candidates(pos, end, rv.rowmap, boot);
        {
            ProfScope p(w, s, prof, 1);
            launch_select(w->cs, nullptr, nq, kc, boot ? (uint32_t)(end - pos) : 0u, s, false, 0u, nullptr, false, /*unsorted=*/true);
        }
        pos = end;
        step++;
    }
    {
        ProfScope p(w, s, prof, 2);
        {
            // members a query may have: twice the results wanted, at least 1024 (the lists hold up to cap entries below tau)
            const uint32_t smax = std::min<uint32_t>(kFinishSmaxMax, std::max<uint32_t>(1024u, 2u * next_pow2_host((uint32_t)k)));
            const bool ckeys = centred && use_tall16; // (the keys of this search were taken about the image's centre)
            launch_finish(metric, order, h->d_X, h->dim, d_q, nq, w->d_qna, w->cs, k, ckeys ? h->d_cstats : h->d_maxnorm2, gamma, finish_beta,
                          h->has_ids ? h->d_ids : nullptr, entries_pos ? rv.rowmap : nullptr, d_dist, d_lab, s, w->h_flags, w->d_done, w->d_xcnt,
                          w->d_xscratch, kFinishSplitMaxQ, smax, ckeys ? h->d_center : nullptr, dot_lb ? h->d_norm2 : nullptr, d_qnrm, gsum,
                          dot_lb ? nullptr : d_qrho, qrho_k);
        }
    }
    std::vector<int> bad;
    int nbad = collect_flagged(w, s, nq, 3u | 4u, nullptr, 0, bad, /*on_host=*/true);
#ifdef LB_DIAG
    if (getenv("LB_TRACE_RETRY")) {
        uint32_t orf = 0;
        int c1 = 0, c2 = 0, c4 = 0;
        for (int i = 0; i < nq; i++) { orf |= w->h_flags[i]; c1 += (w->h_flags[i] & 1u) != 0; c2 += (w->h_flags[i] & 2u) != 0; c4 += (w->h_flags[i] & 4u) != 0; }
        fprintf(stderr, "[retry] attempt %d route %d split %d kc %d sample %d m %d nbad %d flags|=%x (bit0 %d bit1 %d bit2 %d)\n", attempt, route.kind, route.split, kc,
                (int)sp.on, sp.m, nbad, orf, c1, c2, c4);
    }
#endif
    bool fused_gave_up = false;
    if (fused_epoch != 0 && (*w->h_fail == fused_epoch || g_fused_fail_next.exchange(0) != 0)) {
        fused_gave_up = true;
        // a wait inside the fused launch gave up (its workgroups were not co-resident): every query goes the exact way,
        // and the ticket counter is re-based in case the launch did not run to completion
        bad.resize((size_t)nq);
        for (int i = 0; i < nq; i++) bad[(size_t)i] = i;
        nbad = nq;
        h->fused_giveups.fetch_add(1);
        LB_HIP(hipMemsetAsync(w->d_fsync, 0, sizeof(uint32_t), s));
        w->fs_base = 0;
    }
    const int kc_max = (int)(w->cap / 4);
    const bool can_widen = attempt <= 2 && kc < kc_max && (int64_t)kc * 2 < n; // (up to three widenings: x4, x16, x64, capped)
    if (route.split == 3 && cmode == LB_CAND_AUTO && attempt <= 3 && !(nbad > 8 && can_widen)) {
        if (nbad * 8 > nq) { // the fp16 keys are too coarse for this data even with the widened list: leave the route alone
            const int span = h->f16_span.load(); // for a while (doubling spans)
            h->f16_skip.store(span);
            h->f16_span.store(std::min(span * 2, 4096));
        } else if (nbad == 0) {
            h->f16_span.store(16);
        }
    }
    if (attempt >= 1 && attempt <= 3 && nbad <= 8) { // the widened list proved the batch: start the next searches there
        h->kc_hint.store(kc_in, std::memory_order_relaxed);
        h->kc_hint_left.store(256, std::memory_order_relaxed);
    }
    // More than a handful of unproven queries (each would cost an exact scan of the corpus: 0.5 ms per group of 8 at
    // 1M x 768) and the batch is redone instead, cheapest remedy first:
    //   1. the same route keeping FOUR TIMES the candidates (up to three times over, capped at a quarter of the list) -- the usual cause is a tight cluster (hundreds of rows whose
    //      distances differ by less than the keys resolve): once the whole cluster is inside the list, the gap to the first
    //      row outside it is wide and the proof goes through (1M x 768 in clusters of ~1000: 1024 queries 93 ms -> a few ms);
    //   2. (fp16 keys) the split-bf16 route, ~100x finer keys, with the widened list;
    //   3. the exact scan for what is still unproven.
    if (nbad > 8 && !fused_gave_up && attempt < 4) {
        if (can_widen)
            return search_batch_device(h, w, s, nq, d_q, k, d_dist, d_lab, std::min(kc_in * 4, kc_max), prof, fallbacks, allow_f16,
                                       attempt + 1);
        if (route.split == 3 && allow_f16)
            return search_batch_device(h, w, s, nq, d_q, k, d_dist, d_lab, kc_in, prof, fallbacks, /*allow_f16=*/false, 4);
    }
    if (nbad > 0) {
        fallbacks += (int64_t)bad.size();
        scan_with_retry(h, w, s, d_q, nq, bad, k, d_dist, d_lab, prof);
    }
    return LB_OK;
}

// one wave per CU spins for `ticks` of the constant 100 MHz counter and adds (shader cycles, ticks) to out[0 .. 1]
__global__ void clock_probe_kernel(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], c1 - c0);
        atomicAdd(&out[1], r1 - r0);
    }
}

__global__ void fill_empty_kernel(float *dist, int64_t *lab, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dist[i] = FLT_MAX;
        lab[i] = -1;
    }
}
} // namespace
namespace lb {
// the canonical "no result" block: label -1 / distance FLT_MAX (also what comm.hip ships for a failed shard)
void launch_fill_empty(float *dist, int64_t *lab, int64_t n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dist, lab, n);
}
} // namespace lb
namespace {

// candidate rows (int64 positions) -> u32 row map for the mapped scan; rows outside the corpus read row 0
// and are overwritten afterwards
__global__ void rerank_map_kernel(const int64_t *rows, int64_t n, int64_t ntotal, uint32_t *map)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int64_t r = rows[i];
        map[i] = (r >= 0 && r < ntotal) ? (uint32_t)r : 0u;
    }
}
// Score = 1/(1+d) (parallel_search.go:360); invalid rows: MaxFloat32 / 0
__global__ void rerank_score_kernel(const int64_t *rows, int64_t n, int64_t ntotal, float *dist, float *score)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int64_t r = rows[i];
        const bool ok = r >= 0 && r < ntotal;
        const float d = ok ? dist[i] : FLT_MAX;
        dist[i] = d;
        if (score) score[i] = ok ? __fdiv_rn(1.0f, 1.0f + d) : 0.f;
    }
}

void finish_profile(lb_gpu_index *h, Workspace *w)
{
    float ms[5] = {0, 0, 0, 0, 0};
    int cnt[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < w->ev_used; i++) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, w->events[i].a, w->events[i].b) == hipSuccess) {
            ms[w->events[i].cls] += t;
            cnt[w->events[i].cls]++;
        }
    }
    w->ev_used = 0;
    std::lock_guard<std::mutex> g(h->prof_mu);
    for (int i = 0; i < 5; i++) {
        h->prof_ms[i] = ms[i];
        h->prof_n[i] = cnt[i];
    }
}

// Make room for `need` rows.  The corpus itself grows in place (VmmBuf); only the small per-row side
// arrays (norms, ids, mask: 17 B per row) are reallocated geometrically and copied.  Without VMM the
// corpus follows the same malloc + copy scheme (needs old + new resident at once).
void drop_f16_image(lb_gpu_index *h);

int grow(lb_gpu_index *h, int64_t need)
{
    const size_t row_bytes = (size_t)h->dim * sizeof(float);
    int64_t cap = std::max<int64_t>(need, h->capacity * 2);
    cap = std::max<int64_t>(cap, 1024);
    if (h->vmm.ok) {
        try {
            h->vmm.ensure((size_t)need * row_bytes);
            h->d_X = reinterpret_cast<float *>(h->vmm.base);
            h->x_rows_cap = (int64_t)(h->vmm.mapped / row_bytes);
        } catch (const HipErr &e) {
            (void)hipGetLastError(); // do not leave a sticky error behind for the caller's next HIP call
            if (e.e == hipErrorOutOfMemory) throw;
            // The driver refused to extend the mapping (hipMemSetAccess reports "invalid argument" for some
            // chunk sequences on this stack, tools/probe/): move the rows into one hipMalloc'd buffer and
            // grow geometrically from here on.
            float *nx = nullptr;
            LB_HIP(hipMalloc(&nx, (size_t)cap * row_bytes));
            if (h->n > 0) {
                hipError_t ce = hipMemcpyAsync(nx, h->d_X, (size_t)h->n * row_bytes, hipMemcpyDeviceToDevice, h->add_stream);
                if (ce == hipSuccess) ce = hipStreamSynchronize(h->add_stream);
                if (ce != hipSuccess) { (void)hipFree(nx); throw HipErr{ce, "hipMemcpyAsync (leaving the mapped corpus)"}; }
            }
            h->vmm.destroy();
            h->d_X = nx;
            h->x_rows_cap = cap;
        }
    }
    if (need <= h->capacity && (h->vmm.ok || need <= h->x_rows_cap)) return LB_OK;
    if (need <= h->capacity) cap = h->capacity; // only the corpus buffer (malloc mode) is short
    struct Guard { // frees whatever was allocated when a later allocation throws
        float *nx = nullptr, *n2 = nullptr, *rn = nullptr;
        int64_t *ni = nullptr;
        uint8_t *nm = nullptr;
        bool keep = false;
        ~Guard()
        {
            if (keep) return;
            if (nx) (void)hipFree(nx);
            if (n2) (void)hipFree(n2);
            if (rn) (void)hipFree(rn);
            if (ni) (void)hipFree(ni);
            if (nm) (void)hipFree(nm);
        }
    } g;
    const bool grow_x = !h->vmm.ok && cap > h->x_rows_cap;
    const bool grow_side = cap > h->capacity;
    if (grow_x) LB_HIP(hipMalloc(&g.nx, (size_t)cap * row_bytes));
    if (grow_side) {
        LB_HIP(hipMalloc(&g.n2, (size_t)cap * sizeof(float)));
        LB_HIP(hipMalloc(&g.rn, (size_t)cap * sizeof(float)));
        LB_HIP(hipMalloc(&g.ni, (size_t)cap * sizeof(int64_t)));
        LB_HIP(hipMalloc(&g.nm, (size_t)cap));
    }
    if (h->n > 0) {
        if (g.nx) LB_HIP(hipMemcpyAsync(g.nx, h->d_X, (size_t)h->n * row_bytes, hipMemcpyDeviceToDevice, h->add_stream));
        if (grow_side) {
            LB_HIP(hipMemcpyAsync(g.n2, h->d_norm2, (size_t)h->n * sizeof(float), hipMemcpyDeviceToDevice, h->add_stream));
            LB_HIP(hipMemcpyAsync(g.rn, h->d_rnorm, (size_t)h->n * sizeof(float), hipMemcpyDeviceToDevice, h->add_stream));
            LB_HIP(hipMemcpyAsync(g.ni, h->d_ids, (size_t)h->n * sizeof(int64_t), hipMemcpyDeviceToDevice, h->add_stream));
            LB_HIP(hipMemcpyAsync(g.nm, h->d_mask, (size_t)h->n, hipMemcpyDeviceToDevice, h->add_stream));
        }
        LB_HIP(hipStreamSynchronize(h->add_stream));
    }
    g.keep = true;
    if (g.nx) {
        if (h->d_X) (void)hipFree(h->d_X);
        h->d_X = g.nx;
        h->x_rows_cap = cap;
    }
    if (grow_side) {
        if (h->d_norm2) (void)hipFree(h->d_norm2);
        if (h->d_rnorm) (void)hipFree(h->d_rnorm);
        if (h->d_ids) (void)hipFree(h->d_ids);
        if (h->d_mask) (void)hipFree(h->d_mask);
        h->d_norm2 = g.n2; h->d_rnorm = g.rn; h->d_ids = g.ni; h->d_mask = g.nm;
        h->capacity = cap;
    }
    if (h->d_Xs) { // the mirror is rebuilt at the new capacity by the next sync_split_image
        (void)hipFree(h->d_Xs);
        h->d_Xs = nullptr;
        h->xs_rows = 0;
    }
    if (h->d_Xh) drop_f16_image(h); // its planes are `capacity` rows apart: rebuilt by the next sync_f16_image
    return LB_OK;
}

void sync_split_image(lb_gpu_index *h);
void sync_f16_image(lb_gpu_index *h);

// grow(), giving the accelerator copies back first when the device is full.  The fp16 copy (half the corpus's bytes again)
// and the split-bf16 image (all of them again) only speed searches up; the rows are the index.  An Add / reserve that runs
// out of memory while either is resident frees them and tries once more before it reports LB_ERR_OOM; the copy is then
// taken again only when twice the usual margin is free (sync_f16_image), so an index at the edge does not rebuild it per Add.
int grow_or_shed(lb_gpu_index *h, int64_t need)
{
    try {
        return grow(h, need);
    } catch (const HipErr &e) {
        if (e.e != hipErrorOutOfMemory || (!h->d_Xh && !h->d_Xs)) throw;
        (void)hipGetLastError();
        if (h->d_Xh) {
            drop_f16_image(h);
            h->xh_shed = true;
        }
        if (h->d_Xs) { // (the explicit split-image mode cannot be kept: back to the default routes)
            (void)hipFree(h->d_Xs);
            h->d_Xs = nullptr;
            h->xs_rows = 0;
            h->cand_mode.store(LB_CAND_AUTO);
        }
        buf_pool().trim(h->device);
        return grow(h, need);
    }
}

void drop_f16_image(lb_gpu_index *h)
{
    if (h->d_Xh) (void)hipFree(h->d_Xh);
    if (h->d_norm2c) (void)hipFree(h->d_norm2c);
    h->d_Xh = nullptr;
    h->d_norm2c = nullptr;
    h->xh_rows = h->xh_cap = 0;
    h->xh_centred = h->xh_c_ok = h->xh_offset_dom = false;
    h->xh_rho = 0.f;
    if (h->d_xh_rho2) (void)hipMemset(h->d_xh_rho2, 0, sizeof(uint32_t));
}

// Recompute the visible-row list from d_mask (caller holds the exclusive lock).  The list is used
// by searches when at most LB_ROWMAP_MAX_PCT % of the corpus is visible (default 95; measured on
// 1.25M x 1536: time scales with the visible fraction all the way up -- 8.6 ms unfiltered, 7.8 ms at
// 90 %, 4.5 ms at 50 %, 1.16 ms at 10 % for 256 queries -- so only near-total masks keep the per-row test).
void rebuild_rowmap(lb_gpu_index *h)
{
    h->smap_valid = false; // the corpus view changes (callers hold the exclusive lock)
    h->rowmap_on = false;
    h->n_visible = h->n;
    if (!h->has_mask || h->n == 0) return;
    static const int max_pct = lb_tunable("LB_ROWMAP_MAX_PCT", 95);
    if (max_pct <= 0) return;
    hipStream_t s = h->add_stream;
    if (h->rowmap_cap < h->n) {
        if (h->d_rowmap) (void)hipFree(h->d_rowmap);
        h->d_rowmap = nullptr;
        h->rowmap_cap = 0;
        LB_HIP(hipMalloc(&h->d_rowmap, (size_t)h->capacity * sizeof(uint32_t)));
        h->rowmap_cap = h->capacity;
    }
    const int64_t words = compact_scratch_words(h->n);
    if (h->cscratch_words < words) {
        if (h->d_cscratch) (void)hipFree(h->d_cscratch);
        h->d_cscratch = nullptr;
        h->cscratch_words = 0;
        const int64_t cap_words = compact_scratch_words(h->capacity);
        LB_HIP(hipMalloc(&h->d_cscratch, (size_t)cap_words * sizeof(uint32_t)));
        h->cscratch_words = cap_words;
    }
    launch_compact_mask(h->d_mask, h->n, h->d_rowmap, h->d_cscratch, s);
    uint32_t total = 0;
    LB_HIP(hipMemcpyAsync(&total, h->d_cscratch + (words - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LB_HIP(hipStreamSynchronize(s));
    h->n_visible = (int64_t)total;
    h->rowmap_on = h->n_visible * 100 <= h->n * (int64_t)max_pct;
}

__global__ void iota_ids_kernel(int64_t *ids, int64_t start, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[start + i] = start + i;
}

// Rows [h->n, h->n + n) are already in d_X; finish the append (norms, ids, mask).
void finish_add(lb_gpu_index *h, int64_t n, const int64_t *ids_src, bool ids_on_device)
{
    hipStream_t s = h->add_stream;
    const int64_t start = h->n;
    launch_row_norms(h->d_X + (size_t)start * h->dim, n, h->dim, h->d_norm2 + start, h->d_rnorm + start,
                     h->d_maxnorm2, s);
    if (ids_src) {
        if (!h->has_ids && start > 0)
            hipLaunchKernelGGL(iota_ids_kernel, dim3((unsigned)((start + 255) / 256)), dim3(256), 0, s, h->d_ids,
                               (int64_t)0, start);
        LB_HIP(hipMemcpyAsync(h->d_ids + start, ids_src, (size_t)n * sizeof(int64_t),
                              ids_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
        h->has_ids = true;
    } else if (h->has_ids) {
        hipLaunchKernelGGL(iota_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->d_ids, start, n);
    }
    LB_HIP(hipMemsetAsync(h->d_mask + start, 1, (size_t)n, s));
    uint32_t nbits[2] = {0, 0}; // max ||x||^2 so far, as float bits: >= +inf <=> a row with an inf or NaN component
    LB_HIP(hipMemcpyAsync(nbits, h->d_maxnorm2, sizeof nbits, hipMemcpyDeviceToHost, s));
    LB_HIP(hipStreamSynchronize(s));
    LB_LAUNCH_CHECK();
    const uint32_t maxbits = nbits[0];
    h->nonfinite = maxbits >= 0x7f800000u;
    {   // fp16 single-product candidates: no element may overflow fp16 (|x_i| <= ||x|| <= 2^13) and the smallest non-zero
        // row norm bounds the subnormal rounding term of the contraction's error bound (>= 2^-6: kernels_gemm_tall16.hip)
        const float mx = __builtin_bit_cast(float, maxbits), mn = __builtin_bit_cast(float, nbits[1]);
        h->f16_ok = !h->nonfinite && mx <= 67108864.0f /* 2^26 */ && (nbits[1] == 0x7f800000u || mn >= 0.000244140625f /* 2^-12 */);
        h->norm_spread = !h->nonfinite && nbits[1] != 0x7f800000u && mx > 256.0f * mn;
    }
    h->n += n; // the rows are committed from here on: nothing below may fail the call (a retry would duplicate them)
    try {
        sync_split_image(h);
    } catch (const HipErr &) { // no room for the bf16 mirror: back to the default routes
        (void)hipGetLastError();
        h->cand_mode.store(LB_CAND_AUTO);
        if (h->d_Xs) { (void)hipFree(h->d_Xs); h->d_Xs = nullptr; h->xs_rows = 0; }
    }
    sync_f16_image(h); // (never throws: without the image the route stages f32 rows)
    try {
        rebuild_rowmap(h); // appended rows are visible; keep the list in step with the corpus
    } catch (const HipErr &) { // searches fall back to the per-row mask test
        (void)hipGetLastError();
        h->rowmap_on = false;
        h->n_visible = h->n;
    }
}

// bring the split-bf16 mirror up to date with d_X (no-op unless the mode is enabled)
void sync_split_image(lb_gpu_index *h)
{
    if (h->cand_mode.load() != 1 || h->dim % 32 != 0 || h->n == 0) return;
    if (h->d_Xs == nullptr || h->xs_rows > h->n) {
        if (h->d_Xs) (void)hipFree(h->d_Xs);
        h->d_Xs = nullptr;
        h->xs_rows = 0;
        LB_HIP(hipMalloc(&h->d_Xs, (size_t)h->capacity * h->dim * sizeof(float)));
    }
    if (h->xs_rows < h->n) {
        launch_split_bf16(h->d_X + (size_t)h->xs_rows * h->dim, h->d_Xs + (size_t)h->xs_rows * h->dim,
                          h->n - h->xs_rows, h->dim, h->add_stream);
        LB_HIP(hipStreamSynchronize(h->add_stream));
        h->xs_rows = h->n;
    }
}

// Bring the corpus's fp16 image up to date (caller holds the exclusive lock), or drop it when it is not wanted any more.
// Wanted: the fp16 route is on offer for this index (mode, dimension, norms, size) and the copy fits -- half the corpus's
// bytes again, taken only while that leaves max(2 GiB, 1/16 of the device) free.  Never fails the caller: without the image
// the route stages the f32 rows.
void sync_f16_image(lb_gpu_index *h)
{
    const int cm = h->cand_mode.load();
    // (round 4: from 65,536 rows -- the size from which a sampled threshold exists -- instead of 262,144: 100k x 768 at 256 queries
    // 0.298 -> 0.163 ms, 1024: 0.82 -> 0.43; 200k x 768: 0.49 -> 0.21, 1.25 -> 0.57; +50 % of a corpus of this size is 0.15-0.4 GB)
    static const int64_t f16_image_min_rows = lb_tunable("LB_F16_IMAGE_MIN_ROWS", 16384);
    const bool l2 = h->metric == LB_METRIC_EUCLIDEAN;
    // (L2: the image is centred, so what must fit fp16 are the centred norms -- known once the centre is)
    const bool range_ok = l2 ? (!h->nonfinite && (h->xh_declined_n == 0 || h->n >= 2 * h->xh_declined_n)) : h->f16_ok;
    const bool want = h->xh_mode.load() != 0 && !h->xh_failed && range_ok && h->n > 0 &&
                      (cm == LB_CAND_F16 || (cm == LB_CAND_AUTO && h->n >= f16_image_min_rows));
    try {
        if (!want || (h->d_Xh && (h->xh_cap < h->n || h->xh_rows > h->n))) {
            drop_f16_image(h);
            if (!want) return;
        }
        hipStream_t s = h->add_stream;
        if (h->d_Xh == nullptr) {
            const int pd = corpus_f16_plane_dims();
            const size_t need = (size_t)h->capacity * (size_t)((h->dim + pd - 1) / pd * pd) * 2 // (whole planes, the last zero-padded)
                                + (l2 ? (size_t)h->capacity * sizeof(float) : 0);
            size_t fr = 0, tot = 0;
            LB_HIP(hipMemGetInfo(&fr, &tot));
            const size_t keep = std::max<size_t>((size_t)2 << 30, tot / 16) * (h->xh_shed ? 2 : 1);
            if (fr < need + keep) return; // (not remembered: memory may be free again at the next Add)
            if (hipMalloc(&h->d_Xh, need - (l2 ? (size_t)h->capacity * sizeof(float) : 0)) != hipSuccess) {
                (void)hipGetLastError();
                h->d_Xh = nullptr;
                h->xh_failed = true;
                return;
            }
            h->xh_cap = h->capacity;
            h->xh_rows = 0;
            if (l2) { // the centre: column means of the rows there are (fixed from here on: appended rows are shifted by the same)
                const int dpad = ((h->dim + 31) & ~31) + 8;
                if (!h->d_center) LB_HIP(hipMalloc(&h->d_center, (size_t)dpad * sizeof(float)));
                if (!h->d_cstats) LB_HIP(hipMalloc(&h->d_cstats, 2 * sizeof(uint32_t)));
                LB_HIP(hipMalloc(&h->d_norm2c, (size_t)h->capacity * sizeof(float)));
                const uint32_t init[2] = {0u, 0x7f800000u};
                LB_HIP(hipMemcpyAsync(h->d_cstats, init, sizeof init, hipMemcpyHostToDevice, s));
                Lease part(h->device, (size_t)256 * h->dim * sizeof(float));
                launch_column_means(h->d_X, h->n, h->dim, part.as<float>(), h->d_center, dpad, s);
                LB_HIP(hipStreamSynchronize(s)); // (the lease goes back to the pool)
                h->xh_centred = true;
            }
        }
        if (h->xh_rows < h->n) {
            launch_corpus_to_f16(h->d_X, h->xh_rows, h->n, h->dim, h->d_Xh, h->xh_cap, s, h->xh_centred ? h->d_center : nullptr);
            if (h->xh_centred) {
                launch_row_norms(h->d_X + (size_t)h->xh_rows * h->dim, h->n - h->xh_rows, h->dim, h->d_norm2c + h->xh_rows, nullptr,
                                 h->d_cstats, s, h->d_center);
                uint32_t cb[2] = {0, 0};
                LB_HIP(hipMemcpyAsync(cb, h->d_cstats, sizeof cb, hipMemcpyDeviceToHost, s));
                LB_HIP(hipStreamSynchronize(s));
                const float mx = __builtin_bit_cast(float, cb[0]), mn = __builtin_bit_cast(float, cb[1]);
                h->xh_c_ok = cb[0] < 0x7f800000u && mx <= 67108864.0f /* 2^26 */ && (cb[1] == 0x7f800000u || mn >= 0.000244140625f /* 2^-12 */);
                if (!h->xh_c_ok) { // the spread itself is beyond fp16: no image for this index until it has doubled
                    h->xh_declined_n = h->n;
                    drop_f16_image(h);
                    return;
                }
                std::vector<float> hc((size_t)h->dim);
                LB_HIP(hipMemcpy(hc.data(), h->d_center, hc.size() * sizeof(float), hipMemcpyDeviceToHost));
                double c2 = 0.0;
                for (float v : hc) c2 += (double)v * (double)v;
                h->xh_offset_dom = c2 > 4.0 * (double)mx;
            }
            { // the loss of the new rows' images, measured (the candidate keys' error bound: search_batch_device, gamma)
                if (!h->d_xh_rho2) {
                    LB_HIP(hipMalloc(&h->d_xh_rho2, sizeof(uint32_t)));
                    LB_HIP(hipMemsetAsync(h->d_xh_rho2, 0, sizeof(uint32_t), s));
                }
                launch_f16_residual(h->d_X, h->xh_rows, h->n, h->dim, h->xh_centred ? h->d_center : nullptr, h->d_xh_rho2, s);
                uint32_t rb = 0;
                LB_HIP(hipMemcpyAsync(&rb, h->d_xh_rho2, sizeof rb, hipMemcpyDeviceToHost, s));
                LB_HIP(hipStreamSynchronize(s));
                const float r2 = __builtin_bit_cast(float, rb);
                // (a ratio beyond the worst case of normal fp16 values, 2^-22, means elements in the subnormal range or flushed
                // to zero carry weight: still a valid bound as long as it is finite and small enough to be of use)
                h->xh_rho = (rb < 0x7f800000u && r2 <= 1.0e-4f) ? std::sqrt(r2) * 1.000001f : 0.f;
            }
            LB_HIP(hipStreamSynchronize(s));
            h->xh_rows = h->n;
        }
    } catch (const HipErr &) {
        (void)hipGetLastError();
        drop_f16_image(h);
        h->xh_failed = true;
    }
}

} // namespace

namespace lb {
BufPool &buf_pool()
{
    static BufPool *pool = new BufPool(); // leaked on purpose: must outlive every handle at process exit
    return *pool;
}
} // namespace lb

template <typename T>
static int filter_column(lb_gpu_index *h, const T *column, int64_t n, T value, int op, const uint8_t *validity,
                         int64_t voff, int combine)
{
    if (!h || op < 0 || op > 5 || voff < 0) return LB_ERR_INVALID_ARG;
    std::unique_lock<std::shared_mutex> g(h->mu);
    if (h->closed) return LB_ERR_CLOSED;
    if (n != h->n || (n > 0 && !column)) {
        h->set_error("filter column has %lld values, index has %lld rows", (long long)n, (long long)h->n);
        return LB_ERR_INVALID_ARG;
    }
    if (n == 0) { h->has_mask = true; h->rowmap_on = false; return LB_OK; }
    int rc = LB_OK;
    try {
        LB_HIP(hipSetDevice(h->device));
        // column (and validity bitmap) go up through pooled buffers on the index's own stream
        Lease dcol(h->device, (size_t)n * sizeof(T)), dval;
        T *d_col = dcol.as<T>();
        uint8_t *d_val = nullptr;
        LB_HIP(hipMemcpyAsync(d_col, column, (size_t)n * sizeof(T), hipMemcpyHostToDevice, h->add_stream));
        if (validity) {
            const size_t vb = (size_t)((voff + n + 7) / 8);
            dval.reset(h->device, vb);
            d_val = dval.as<uint8_t>();
            LB_HIP(hipMemcpyAsync(d_val, validity, vb, hipMemcpyHostToDevice, h->add_stream));
        }
        const int comb = (combine && h->has_mask) ? 1 : 0; // AND into "no filter" == replace
        if constexpr (sizeof(T) == 8)
            launch_match_int64(reinterpret_cast<const int64_t *>(d_col), n, (int64_t)value, op, d_val, voff, h->d_mask, comb, h->add_stream);
        else
            launch_match_float32(reinterpret_cast<const float *>(d_col), n, (float)value, op, d_val, voff, h->d_mask, comb, h->add_stream);
        LB_LAUNCH_CHECK();
        LB_HIP(hipStreamSynchronize(h->add_stream));
        h->has_mask = true;
        rebuild_rowmap(h);
    } catch (const HipErr &e) {
        rc = fail_hip(h, e);
    }
    return rc;
}


template <typename T>
static int match_host(int device, const T *src, int64_t n, T value, int op, uint8_t *dst)
{
    if (n < 0 || op < 0 || op > 5) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    if (!src || !dst) return LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    try {
        LB_HIP(hipSetDevice(device));
        Lease ds(device, (size_t)n * sizeof(T)), dd(device, (size_t)n);
        LB_HIP(hipMemcpy(ds.p, src, (size_t)n * sizeof(T), hipMemcpyHostToDevice));
        if constexpr (sizeof(T) == 8) launch_match_int64(ds.as<int64_t>(), n, (int64_t)value, op, nullptr, 0, dd.as<uint8_t>(), 0, nullptr);
        else launch_match_float32(ds.as<float>(), n, (float)value, op, nullptr, 0, dd.as<uint8_t>(), 0, nullptr);
        LB_LAUNCH_CHECK();
        LB_HIP(hipMemcpy(dst, dd.p, (size_t)n, hipMemcpyDeviceToHost));
    } catch (const HipErr &e) {
        return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
    }
    return LB_OK;
}


// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int lb_gpu_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

const char *lb_gpu_version(void) { return "longbow_gpu 0.1.0 (gfx950, HIP)"; }

const char *lb_gpu_status_string(int status)
{
    switch (status) {
    case LB_OK: return "ok";
    case LB_ERR_INVALID_ARG: return "invalid argument";
    case LB_ERR_CLOSED: return "index is closed";
    case LB_ERR_NO_DEVICE: return "GPU not available";
    case LB_ERR_HIP: return "HIP runtime error";
    case LB_ERR_OOM: return "out of device memory";
    case LB_ERR_UNSUPPORTED: return "unsupported configuration";
    case LB_ERR_CANCELLED: return "context canceled";
    case LB_ERR_DEADLINE: return "context deadline exceeded";
    default: return "internal error";
    }
}

lb_gpu_index *lb_gpu_index_new(int device, int dim, int metric, int *out_status)
{
    auto st = [&](int v) { if (out_status) *out_status = v; };
    if (dim <= 0 || metric < 0 || metric > 2) { st(LB_ERR_INVALID_ARG); return nullptr; }
    // the kernels stage one query row (and the re-rank a 256 x 64-float tile plus up to 4096 keys) in LDS:
    // LB_MAX_DIM keeps every launch inside the 160 KB a workgroup may declare
    if (dim > LB_MAX_DIM) { st(LB_ERR_UNSUPPORTED); return nullptr; }
    if (!device_ok(device)) { st(LB_ERR_NO_DEVICE); return nullptr; }
    auto *h = new (std::nothrow) lb_gpu_index();
    if (!h) { st(LB_ERR_OOM); return nullptr; }
    h->device = device; h->dim = dim; h->metric = metric;
    try {
        LB_HIP(hipSetDevice(device));
        LB_HIP(hipStreamCreateWithFlags(&h->add_stream, hipStreamNonBlocking));
        LB_HIP(hipMalloc(&h->d_maxnorm2, 2 * sizeof(uint32_t)));
        const uint32_t norm_init[2] = {0u, 0x7f800000u}; // max ||x||^2 so far, smallest non-zero ||x||^2 so far (float bits)
        LB_HIP(hipMemcpy(h->d_maxnorm2, norm_init, sizeof norm_init, hipMemcpyHostToDevice));
        static const int use_vmm = lb_tunable("LB_VMM", 1);
        if (use_vmm) (void)h->vmm.init(device); // on failure: geometric hipMalloc + copy
    } catch (const HipErr &e) {
        st(e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP);
        lb_gpu_index_free(h);
        return nullptr;
    }
    st(LB_OK);
    return h;
}

void lb_gpu_index_free(lb_gpu_index *h)
{
    if (!h) return;
    {
        std::unique_lock<std::shared_mutex> g(h->mu);
        h->closed = true;
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        {
            std::lock_guard<std::mutex> g2(h->ws_mu);
            h->ws_free.clear();
            h->hs_free.clear();
        }
        if (h->vmm.ok) h->vmm.destroy();
        else if (h->d_X) (void)hipFree(h->d_X);
        h->d_X = nullptr;
        if (h->d_norm2) (void)hipFree(h->d_norm2);
        if (h->d_rnorm) (void)hipFree(h->d_rnorm);
        if (h->d_ids) (void)hipFree(h->d_ids);
        if (h->d_mask) (void)hipFree(h->d_mask);
        if (h->d_rowmap) (void)hipFree(h->d_rowmap);
        if (h->d_smap) (void)hipFree(h->d_smap);
        if (h->d_cscratch) (void)hipFree(h->d_cscratch);
        if (h->d_maxnorm2) (void)hipFree(h->d_maxnorm2);
        if (h->d_Xs) (void)hipFree(h->d_Xs);
        if (h->d_Xh) (void)hipFree(h->d_Xh);
        if (h->d_norm2c) (void)hipFree(h->d_norm2c);
        if (h->d_center) (void)hipFree(h->d_center);
        if (h->d_cstats) (void)hipFree(h->d_cstats);
        if (h->d_xh_rho2) (void)hipFree(h->d_xh_rho2);
        for (int i = 0; i < 2; i++) {
            if (h->h_stage[i]) (void)hipHostFree(h->h_stage[i]);
            if (h->stage_ev[i]) (void)hipEventDestroy(h->stage_ev[i]);
        }
        if (h->add_stream) (void)hipStreamDestroy(h->add_stream);
    }
    delete h;
}

const char *lb_gpu_last_error(const lb_gpu_index *h)
{
    if (!h) return "null handle";
    std::lock_guard<std::mutex> g(h->err_mu);
    return h->last_error.c_str();
}

int lb_gpu_index_set_order(lb_gpu_index *h, int order)
{
    if (!h || (order != LB_ORDER_SEQ && order != LB_ORDER_UNROLL4)) return LB_ERR_INVALID_ARG;
    h->order.store(order);
    return LB_OK;
}

int lb_gpu_index_set_candidate_mode(lb_gpu_index *h, int mode)
{
    if (!h || mode < LB_CAND_F32_MFMA || mode > LB_CAND_F16) return LB_ERR_INVALID_ARG;
    std::unique_lock<std::shared_mutex> g(h->mu);
    if (h->closed) return LB_ERR_CLOSED;
    if ((mode == LB_CAND_SPLIT_BF16 || mode == LB_CAND_SPLIT_BF16_INREG) && h->dim % 32 != 0) {
        h->set_error("split-bf16 candidates need dim %% 32 == 0 (dim = %d)", h->dim);
        return LB_ERR_UNSUPPORTED;
    }
    try {
        LB_HIP(hipSetDevice(h->device));
        h->cand_mode.store(mode);
        if (mode == 1) sync_split_image(h);
        else if (h->d_Xs) { (void)hipFree(h->d_Xs); h->d_Xs = nullptr; h->xs_rows = 0; }
        sync_f16_image(h);
    } catch (const HipErr &e) {
        h->cand_mode.store(LB_CAND_AUTO);
        return fail_hip(h, e);
    }
    return LB_OK;
}

int lb_gpu_index_set_f16_image(lb_gpu_index *h, int mode)
{
    if (!h || mode < 0 || mode > 1) return LB_ERR_INVALID_ARG;
    try {
        std::unique_lock<std::shared_mutex> g(h->mu);
        if (h->closed) return LB_ERR_CLOSED;
        if (hipSetDevice(h->device) != hipSuccess) { (void)hipGetLastError(); return LB_ERR_HIP; }
        h->xh_mode.store(mode);
        if (mode) { h->xh_failed = h->xh_shed = false; h->xh_declined_n = 0; }
        sync_f16_image(h);
    } catch (...) {
        return LB_ERR_INTERNAL;
    }
    return LB_OK;
}

int64_t lb_gpu_index_f16_image_bytes(const lb_gpu_index *h)
{
    if (!h) return 0;
    std::shared_lock<std::shared_mutex> g(const_cast<lb_gpu_index *>(h)->mu); // (the copy is built and dropped under the writer lock)
    const int pd = corpus_f16_plane_dims();
    return h->d_Xh ? (int64_t)h->xh_cap * ((h->dim + pd - 1) / pd * pd) * 2 : 0;
}

int64_t lb_gpu_index_ntotal(const lb_gpu_index *h)
{
    if (!h) return 0;
    std::shared_lock<std::shared_mutex> g(const_cast<lb_gpu_index *>(h)->mu); // (Add commits its rows under the writer lock)
    return h->n;
}
int lb_gpu_index_dim(const lb_gpu_index *h) { return h ? h->dim : 0; }
int lb_gpu_index_device(const lb_gpu_index *h) { return h ? h->device : -1; }

int lb_gpu_index_reserve(lb_gpu_index *h, int64_t n_total)
{
    if (!h || n_total < 0) return LB_ERR_INVALID_ARG;
    std::unique_lock<std::shared_mutex> g(h->mu);
    if (h->closed) return LB_ERR_CLOSED;
    try {
        LB_HIP(hipSetDevice(h->device));
        return grow_or_shed(h, n_total);
    } catch (const HipErr &e) {
        return fail_hip(h, e);
    }
}

int lb_gpu_index_add(lb_gpu_index *h, int64_t n, const float *vectors, const int64_t *ids)
{
    if (!h || n < 0 || (n > 0 && !vectors)) return LB_ERR_INVALID_ARG;
    std::unique_lock<std::shared_mutex> g(h->mu);
    if (h->closed) { h->set_error("index is closed"); return LB_ERR_CLOSED; }
    if (n == 0) return LB_OK;
    if (h->n + n > (int64_t)0xffffffffll) { h->set_error("more than 2^32 rows per device"); return LB_ERR_UNSUPPORTED; }
    try {
        LB_HIP(hipSetDevice(h->device));
        grow_or_shed(h, h->n + n);
        for (int i = 0; i < 2; i++) {
            if (!h->h_stage[i]) LB_HIP(hipHostMalloc(&h->h_stage[i], kStageBytes, hipHostMallocDefault));
            if (!h->stage_ev[i]) LB_HIP(hipEventCreateWithFlags(&h->stage_ev[i], hipEventDisableTiming));
        }
        const size_t total = (size_t)n * h->dim * sizeof(float);
        const char *src = reinterpret_cast<const char *>(vectors);
        char *dst = reinterpret_cast<char *>(h->d_X + (size_t)h->n * h->dim);
        size_t off = 0;
        // Large batches: pin the caller's buffer (the Arrow values buffer) for the duration of the call and
        // DMA straight out of it -- no host-side copy (SURVEY 8b ownership row: "the shim hipHostRegisters
        // the Arrow values buffer for the duration of the call").  Small batches, or a buffer the driver
        // will not pin, go through the double-buffered pinned slabs below.
        if (total >= (size_t)g_add_register_min.load() && g_add_register_min.load() > 0) {
            void *reg = const_cast<float *>(vectors);
            if (hipHostRegister(reg, total, hipHostRegisterDefault) == hipSuccess) {
                hipError_t e = hipSuccess;
                const size_t piece = (size_t)256 << 20;
                for (size_t o = 0; o < total && e == hipSuccess; o += piece)
                    e = hipMemcpyAsync(dst + o, src + o, std::min(piece, total - o), hipMemcpyHostToDevice, h->add_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->add_stream);
                (void)hipHostUnregister(reg);
                if (e != hipSuccess) throw HipErr{e, "hipMemcpyAsync (registered add)"};
                off = total;
            } else {
                (void)hipGetLastError();
            }
        }
        // host buffer -> pinned slab (memcpy) -> HBM (async DMA), two slabs in flight
        int slab = 0;
        bool used[2] = {false, false};
        while (off < total) {
            const size_t len = std::min(kStageBytes, total - off);
            if (used[slab]) LB_HIP(hipEventSynchronize(h->stage_ev[slab]));
            std::memcpy(h->h_stage[slab], src + off, len);
            LB_HIP(hipMemcpyAsync(dst + off, h->h_stage[slab], len, hipMemcpyHostToDevice, h->add_stream));
            LB_HIP(hipEventRecord(h->stage_ev[slab], h->add_stream));
            used[slab] = true;
            slab ^= 1;
            off += len;
        }
        finish_add(h, n, ids, false);
    } catch (const HipErr &e) {
        return fail_hip(h, e);
    }
    return LB_OK;
}

int lb_gpu_index_add_device(lb_gpu_index *h, int64_t n, const float *d_vectors, const int64_t *d_ids)
{
    if (!h || n < 0 || (n > 0 && !d_vectors)) return LB_ERR_INVALID_ARG;
    std::unique_lock<std::shared_mutex> g(h->mu);
    if (h->closed) { h->set_error("index is closed"); return LB_ERR_CLOSED; }
    if (n == 0) return LB_OK;
    if (h->n + n > (int64_t)0xffffffffll) { h->set_error("more than 2^32 rows per device"); return LB_ERR_UNSUPPORTED; }
    try {
        LB_HIP(hipSetDevice(h->device));
        grow_or_shed(h, h->n + n);
        LB_HIP(hipMemcpyAsync(h->d_X + (size_t)h->n * h->dim, d_vectors, (size_t)n * h->dim * sizeof(float),
                              hipMemcpyDeviceToDevice, h->add_stream));
        finish_add(h, n, d_ids, true);
    } catch (const HipErr &e) {
        return fail_hip(h, e);
    }
    return LB_OK;
}

int lb_gpu_index_set_filter(lb_gpu_index *h, const uint8_t *mask, int64_t n)
{
    if (!h) return LB_ERR_INVALID_ARG;
    std::unique_lock<std::shared_mutex> g(h->mu);
    if (h->closed) return LB_ERR_CLOSED;
    if (!mask) { h->has_mask = false; h->rowmap_on = false; h->smap_valid = false; return LB_OK; }
    if (n != h->n) { h->set_error("filter mask has %lld bytes, index has %lld rows", (long long)n, (long long)h->n); return LB_ERR_INVALID_ARG; }
    try {
        LB_HIP(hipSetDevice(h->device));
        if (n > 0) LB_HIP(hipMemcpy(h->d_mask, mask, (size_t)n, hipMemcpyHostToDevice));
        h->has_mask = true;
        rebuild_rowmap(h);
    } catch (const HipErr &e) {
        return fail_hip(h, e);
    }
    return LB_OK;
}

int lb_gpu_index_filter_int64(lb_gpu_index *h, const int64_t *column, int64_t n, int64_t value, int op,
                              const uint8_t *validity, int64_t validity_offset, int combine)
{
    return filter_column<int64_t>(h, column, n, value, op, validity, validity_offset, combine);
}

int lb_gpu_index_filter_float32(lb_gpu_index *h, const float *column, int64_t n, float value, int op,
                                const uint8_t *validity, int64_t validity_offset, int combine)
{
    return filter_column<float>(h, column, n, value, op, validity, validity_offset, combine);
}

int lb_simd_match_int64(int device, const int64_t *src, int64_t n, int64_t value, int op, uint8_t *dst)
{
    return match_host<int64_t>(device, src, n, value, op, dst);
}

int lb_simd_match_float32(int device, const float *src, int64_t n, float value, int op, uint8_t *dst)
{
    return match_host<float>(device, src, n, value, op, dst);
}

int lb_simd_and_bytes(int device, uint8_t *dst, const uint8_t *src, int64_t n)
{
    if (n < 0) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    if (!dst || !src) return LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    try {
        LB_HIP(hipSetDevice(device));
        Lease da(device, (size_t)n), db(device, (size_t)n);
        LB_HIP(hipMemcpy(da.p, dst, (size_t)n, hipMemcpyHostToDevice));
        LB_HIP(hipMemcpy(db.p, src, (size_t)n, hipMemcpyHostToDevice));
        launch_and_bytes(da.as<uint8_t>(), db.as<uint8_t>(), n, nullptr);
        LB_LAUNCH_CHECK();
        LB_HIP(hipMemcpy(dst, da.p, (size_t)n, hipMemcpyDeviceToHost));
    } catch (const HipErr &e) {
        return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
    }
    return LB_OK;
}

int64_t lb_gpu_index_last_fallbacks(const lb_gpu_index *h) { return h ? h->last_fallbacks.load() : 0; }
int64_t lb_gpu_index_fused_giveups(const lb_gpu_index *h) { return h ? h->fused_giveups.load() : 0; }
int lb_gpu_index_last_route(const lb_gpu_index *h) { return h ? h->last_route.load() : 0; }

int lb_gpu_index_search_device_ctx(lb_gpu_index *h, int64_t nq, const float *d_queries, int k, float *d_dist,
                                   int64_t *d_labels, void *stream, const lb_cancel *ctx)
{
    if (!h || nq < 0 || k <= 0 || (nq > 0 && (!d_queries || !d_dist || !d_labels))) return LB_ERR_INVALID_ARG;
    std::shared_lock<std::shared_mutex> g(h->mu);
    if (h->closed) { h->set_error("index is closed"); return LB_ERR_CLOSED; }
    if (nq == 0) return LB_OK;
    if (k > LB_MAX_K) { h->set_error("k=%d exceeds the supported maximum %d", k, LB_MAX_K); return LB_ERR_UNSUPPORTED; }
    if (const int st = ctx_state(ctx)) { h->set_error(st == LB_ERR_CANCELLED ? "context canceled" : "context deadline exceeded"); return st; }
#ifdef LB_DIAG
    if (g_search_fail_next.exchange(0) != 0) { h->set_error("search failure forced by lb_debug_search_fail_next"); return LB_ERR_INTERNAL; }
#endif
    std::unique_ptr<Workspace> w;
    hipStream_t s = nullptr;
    try {
        LB_HIP(hipSetDevice(h->device));
        int kc;
        uint32_t cap;
        cand_geometry(k, kc, cap);
        w = acquire_ws(h, (int)std::min<int64_t>(nq, kMaxBatch), cap);
        s = stream ? (hipStream_t)stream : w->stream;
        const bool prof = h->profiling.load() != 0;
        w->ev_used = 0;
        w->ctx = ctx;
        int64_t fallbacks = 0;
        if (h->n == 0) {
            launch_fill_empty(d_dist, d_labels, nq * k, s);
        } else {
            for (int64_t q0 = 0; q0 < nq; q0 += kMaxBatch) {
                const int bq = (int)std::min<int64_t>(kMaxBatch, nq - q0);
                int rc = search_batch_device(h, w.get(), s, bq, d_queries + (size_t)q0 * h->dim, k,
                                             d_dist + (size_t)q0 * k, d_labels + (size_t)q0 * k, kc, prof, fallbacks);
                if (rc != LB_OK) { w->ctx = nullptr; release_ws(h, std::move(w)); return rc; }
            }
        }
        LB_LAUNCH_CHECK();
        LB_HIP(hipStreamSynchronize(s));
        h->last_fallbacks.store(fallbacks);
        if (prof) finish_profile(h, w.get());
        w->ctx = nullptr;
        release_ws(h, std::move(w));
    } catch (const HipErr &e) {
        return fail_hip(h, e);
    } catch (const CtxErr &c) {
        // stop enqueuing, let what is already on the stream finish (it writes into the caller's buffers), report.
        // The output buffers hold unspecified values, as after any failed call.
        (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
        if (w) {
            w->ctx = nullptr;
            w->ev_used = 0;
            // a fused launch may have been skipped between its host-side bookkeeping and the device: re-base the ticket
            (void)hipMemsetAsync(w->d_fsync, 0, sizeof(uint32_t), s);
            (void)hipStreamSynchronize(s);
            w->fs_base = 0;
            release_ws(h, std::move(w));
        }
        h->set_error(c.code == LB_ERR_CANCELLED ? "context canceled" : "context deadline exceeded");
        return c.code;
    }
    return LB_OK;
}

int lb_gpu_index_search_device(lb_gpu_index *h, int64_t nq, const float *d_queries, int k, float *d_dist,
                               int64_t *d_labels, void *stream)
{
    return lb_gpu_index_search_device_ctx(h, nq, d_queries, k, d_dist, d_labels, stream, nullptr);
}

lb_cancel *lb_cancel_new(void) { return new (std::nothrow) lb_cancel(); }
void lb_cancel_free(lb_cancel *c) { delete c; }
void lb_cancel_fire(lb_cancel *c) { if (c) c->fired.store(1); }
void lb_cancel_set_deadline_ms(lb_cancel *c, int64_t ms_from_now)
{
    if (!c) return;
    if (ms_from_now < 0) { c->deadline_ns.store(0); return; }
    const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    long long d = now + (long long)ms_from_now * 1000000ll;
    if (d == 0) d = 1;
    c->deadline_ns.store(d);
}
int lb_cancel_state(const lb_cancel *c) { return ctx_state(c); }

// Host-pointer search of one or several requests with the same k as ONE device batch: borrowed host buffers -> pooled pinned
// slab -> HBM (async DMA on the call's own stream), and back; every request gets its rows of the result.
static int host_search_multi(lb_gpu_index *h, HostReq *const *reqs, int nreq, int k, const lb_cancel *ctx)
{
    int64_t nq = 0;
    for (int i = 0; i < nreq; i++) nq += reqs[i]->nq;
    // (before any staging is sized: nq * k * 12 bytes of pinned + device memory per call)
    if (k > LB_MAX_K) { h->set_error("k=%d exceeds the supported maximum %d", k, LB_MAX_K); return LB_ERR_UNSUPPORTED; }
    if (nq > ((int64_t)1 << 40) / ((int64_t)h->dim + 3 * (int64_t)k)) { h->set_error("batch too large"); return LB_ERR_INVALID_ARG; }
    const size_t qb = (size_t)nq * h->dim * sizeof(float);
    const size_t db = (((size_t)nq * k * sizeof(float)) + 15) & ~(size_t)15;
    const size_t lb_ = (size_t)nq * k * sizeof(int64_t);
    const size_t qoff = 0, doff = (qb + 15) & ~(size_t)15, loff = doff + db, total = loff + lb_;
    std::unique_ptr<HostStage> st;
    int rc = LB_OK;
    try {
        LB_HIP(hipSetDevice(h->device));
        {
            std::lock_guard<std::mutex> g(h->ws_mu);
            for (size_t i = 0; i < h->hs_free.size(); i++)
                if (h->hs_free[i]->bytes >= total) {
                    st = std::move(h->hs_free[i]);
                    h->hs_free.erase(h->hs_free.begin() + (long)i);
                    break;
                }
        }
        if (!st) {
            st = std::make_unique<HostStage>();
            st->device = h->device;
            st->bytes = std::max<size_t>(total, 1u << 20);
            LB_HIP(hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking));
            LB_HIP(hipMalloc(&st->d_buf, st->bytes));
            LB_HIP(hipHostMalloc(&st->h_buf, st->bytes, hipHostMallocDefault));
        }
        char *hb = static_cast<char *>(st->h_buf), *dbuf = static_cast<char *>(st->d_buf);
        {
            size_t off = qoff;
            for (int i = 0; i < nreq; i++) {
                const size_t b = (size_t)reqs[i]->nq * h->dim * sizeof(float);
                std::memcpy(hb + off, reqs[i]->q, b);
                off += b;
            }
        }
        LB_HIP(hipMemcpyAsync(dbuf + qoff, hb + qoff, qb, hipMemcpyHostToDevice, st->stream));
        // Small results (a few queries: the latency path) are written by the last kernel straight into the pinned slab -- no
        // device-to-host copy and no second wait behind it (~20 us of a 0.35 ms call); the search's own stream
        // synchronisation is what makes them visible.  Large results go through HBM and one DMA.
        const bool direct = db + lb_ <= ((size_t)64 << 10);
        char *obuf = direct ? hb : dbuf;
        rc = lb_gpu_index_search_device_ctx(h, nq, reinterpret_cast<const float *>(dbuf + qoff), k,
                                            reinterpret_cast<float *>(obuf + doff), reinterpret_cast<int64_t *>(obuf + loff),
                                            st->stream, ctx);
        if (rc == LB_OK) {
            if (!direct) {
                LB_HIP(hipMemcpyAsync(hb + doff, dbuf + doff, db + lb_, hipMemcpyDeviceToHost, st->stream));
                LB_HIP(hipStreamSynchronize(st->stream));
            }
            size_t row = 0;
            for (int i = 0; i < nreq; i++) {
                const size_t n = (size_t)reqs[i]->nq * k;
                std::memcpy(reqs[i]->dist, hb + doff + row * sizeof(float), n * sizeof(float));
                std::memcpy(reqs[i]->labels, hb + loff + row * sizeof(int64_t), n * sizeof(int64_t));
                row += n;
            }
        }
        std::lock_guard<std::mutex> g(h->ws_mu);
        if (h->hs_free.size() < 8) h->hs_free.push_back(std::move(st));
    } catch (const HipErr &e) {
        rc = fail_hip(h, e);
    } catch (...) {
        h->set_error("internal error (exception)");
        rc = LB_ERR_INTERNAL;
    }
    return rc;
}

int lb_gpu_index_search_ctx(lb_gpu_index *h, int64_t nq, const float *queries, int k, float *dist, int64_t *labels,
                            const lb_cancel *ctx)
{
    if (!h || nq < 0 || k <= 0 || (nq > 0 && (!queries || !dist || !labels))) return LB_ERR_INVALID_ARG;
    if (nq == 0) return LB_OK;
    {
        std::shared_lock<std::shared_mutex> g(h->mu);
        if (h->closed) { h->set_error("index is closed"); return LB_ERR_CLOSED; }
    }
    // (a call with a cancellation context is searched on its own: its deadline is not its neighbours')
    HostReq me{queries, nq, dist, labels, k};
    if (!ctx && nq <= SearchCombiner::kMaxNq && k <= LB_MAX_K && h->combiner.on.load() != 0)
        return h->combiner.search(me, [h](HostReq *const *reqs, int n, int kk) { return host_search_multi(h, reqs, n, kk, nullptr); });
    HostReq *one = &me;
    return host_search_multi(h, &one, 1, k, ctx);
}

int lb_gpu_index_set_search_combining(lb_gpu_index *h, int enable)
{
    if (!h) return LB_ERR_INVALID_ARG;
    h->combiner.on.store(enable ? 1 : 0);
    return LB_OK;
}

int lb_gpu_index_combining_stats(const lb_gpu_index *h, int64_t out[2])
{
    if (!h || !out) return LB_ERR_INVALID_ARG;
    out[0] = h->combiner.batches.load();
    out[1] = h->combiner.requests.load();
    return LB_OK;
}

int lb_gpu_index_search(lb_gpu_index *h, int64_t nq, const float *queries, int k, float *dist, int64_t *labels)
{
    return lb_gpu_index_search_ctx(h, nq, queries, k, dist, labels, nullptr);
}

int lb_gpu_index_set_profiling(lb_gpu_index *h, int enable)
{
    if (!h) return LB_ERR_INVALID_ARG;
    h->profiling.store(enable ? 1 : 0);
    return LB_OK;
}

int lb_gpu_index_last_timing(const lb_gpu_index *hc, float ms[5], int n_launch[5])
{
    if (!hc || !ms || !n_launch) return LB_ERR_INVALID_ARG;
    auto *h = const_cast<lb_gpu_index *>(hc);
    std::lock_guard<std::mutex> g(h->prof_mu);
    for (int i = 0; i < 5; i++) { ms[i] = h->prof_ms[i]; n_launch[i] = h->prof_n[i]; }
    return LB_OK;
}

#ifdef LB_DIAG
// Diagnostic build only (python -m longbow_amd.build --diag -> liblongbow_gpu_diag.so; the tests that force a
// fallback path load that library): none of these symbols exists in liblongbow_gpu.so.
// Test hooks (both settings are exact; they only choose between two schedules / expose host logic).
void lb_debug_set_sample_tau(int v) { g_sample_tau.store(v); } // 0: classic bootstrap schedule only
void lb_debug_vmm_fail_next(int v) { g_vmm_fail_next.store(v); } // the next in-place growth is refused (-> hipMalloc + copy)
void lb_debug_fused_fail_next(int v) { g_fused_fail_next.store(v); } // the next fused sample launch counts as timed out (-> exact path for the batch)
int lb_debug_last_route(void) { return g_last_route.load(); } // RouteKind * 10 + split of the most recent batched search
void lb_debug_search_fail_next(int v) { g_search_fail_next.store(v); } // the next search in this process returns LB_ERR_INTERNAL
void lb_debug_set_add_register_min(long long bytes) { g_add_register_min.store(bytes); } // ingest A/B (tools/bench_add.py)
// host-only: the sampled-threshold plan for a view of n rows (tests check its invariants without a GPU);
// out = {on, span, count, m}
void lb_debug_sample_plan(long long n, int keep, unsigned cap, unsigned count_max, long long *out)
{
    const SamplePlan p = sample_plan((int64_t)n, keep, cap, count_max ? count_max : 8192u);
    out[0] = p.on ? 1 : 0;
    out[1] = p.span;
    out[2] = p.count;
    out[3] = p.m;
}
// timing-only ablations whose results are wrong by design, the in-kernel clock probe, staging A/B
void lb_debug_set_gemm_ablation(int v) { lb::g_gemm_ablation = v; }
void lb_debug_set_gemm_glds(int v) { lb::g_gemm_glds = v; }
void lb_debug_set_adc_ablation(int v) { lb::g_adc_ablation = v; }
int lb_debug_gemm_occupancy(void) { return lb::debug_gemm_occupancy(); }
void lb_debug_read_clock_probe(unsigned long long *out, int reset) { lb::read_clock_probe(out, reset != 0); }
void lb_debug_read_fused_probe(unsigned long long *out, int reset) { lb::read_fused_probe(out, reset != 0); }
void lb_debug_read_tall2_probe(unsigned long long *out, int reset) { lb::read_tall2_probe(out, reset != 0); }
void lb_debug_read_tall16_probe(unsigned long long *out, int reset) { lb::read_tall16_probe(out, reset != 0); }
void lb_debug_read_finish_probe(unsigned long long *out, int reset) { lb::read_finish_probe(out, reset != 0); }
#endif

// ---- candidate re-rank (processChunkInternal) ------------------------------------------
int lb_gpu_index_rerank_device(lb_gpu_index *h, const float *d_query, const int64_t *d_rows, int64_t n, int order,
                               float *d_dist, float *d_score, void *stream)
{
    if (!h || n < 0 || (order != -1 && order != LB_ORDER_SEQ && order != LB_ORDER_UNROLL4)) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    if (!d_query || !d_rows || !d_dist) return LB_ERR_INVALID_ARG;
    if (n > (int64_t)0x7fffffff) return LB_ERR_INVALID_ARG;
    std::shared_lock<std::shared_mutex> g(h->mu);
    if (h->closed) { h->set_error("index is closed"); return LB_ERR_CLOSED; }
    try {
        LB_HIP(hipSetDevice(h->device));
        const int ord = order == -1 ? h->order.load() : order;
        Lease map(h->device, (size_t)n * sizeof(uint32_t)), qna(h->device, 16);
        // a private stream per call when the caller gave none (the null stream would serialise callers)
        hipStream_t s = (hipStream_t)stream;
        std::unique_ptr<Workspace> w;
        if (!s) {
            int kc; uint32_t cap;
            cand_geometry(1, kc, cap);
            w = acquire_ws(h, 1, cap);
            s = w->stream;
        }
        const unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(rerank_map_kernel, dim3(blocks), dim3(256), 0, s, d_rows, n, h->n, map.as<uint32_t>());
        if (h->n > 0) {
            if (h->metric == LB_METRIC_COSINE) launch_query_norms(ord, d_query, nullptr, 1, h->dim, qna.as<float>(), s);
            CandState cs{};
            launch_scan(h->metric, ord, /*raw_dot=*/false, h->d_X, 0, n, h->dim, d_query, nullptr, 1, qna.as<float>(), nullptr,
                        map.as<uint32_t>(), cs, false, d_dist, n, s);
        }
        hipLaunchKernelGGL(rerank_score_kernel, dim3(blocks), dim3(256), 0, s, d_rows, n, h->n, d_dist, d_score);
        LB_LAUNCH_CHECK();
        LB_HIP(hipStreamSynchronize(s));
        if (w) release_ws(h, std::move(w));
    } catch (const HipErr &e) {
        return fail_hip(h, e);
    }
    return LB_OK;
}

int lb_gpu_index_rerank(lb_gpu_index *h, const float *query, const int64_t *rows, int64_t n, int order, float *dist,
                        float *score)
{
    if (!h || n < 0) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    if (!query || !rows || !dist) return LB_ERR_INVALID_ARG;
    try {
        LB_HIP(hipSetDevice(h->device));
        // [query | rows] up through one pinned block, [dist | score] back through another
        const size_t qb = ((size_t)h->dim * 4 + 15) & ~(size_t)15, rb = (size_t)n * 8, ob = (size_t)n * 4;
        Lease hin(h->device, qb + rb, true), din(h->device, qb + rb), dout(h->device, 2 * ob), hout(h->device, 2 * ob, true);
        std::memcpy(hin.p, query, (size_t)h->dim * 4);
        std::memcpy(hin.as<char>() + qb, rows, rb);
        LB_HIP(hipMemcpy(din.p, hin.p, qb + rb, hipMemcpyHostToDevice));
        const int rc = lb_gpu_index_rerank_device(h, din.as<float>(), reinterpret_cast<const int64_t *>(din.as<char>() + qb), n,
                                                  order, dout.as<float>(), dout.as<float>() + n, nullptr);
        if (rc != LB_OK) return rc;
        LB_HIP(hipMemcpy(hout.p, dout.p, 2 * ob, hipMemcpyDeviceToHost));
        std::memcpy(dist, hout.p, ob);
        if (score) std::memcpy(score, hout.as<char>() + ob, ob);
    } catch (const HipErr &e) {
        return fail_hip(h, e);
    }
    return LB_OK;
}

// ---- simd batch interface ---------------------------------------------------------
int lb_simd_distance_batch_flat_device(int device, int metric, int order, const float *d_query,
                                       const float *d_flat, int64_t n, int dims, float *d_results, void *stream)
{
    if (metric < 0 || metric > 2 || (order != 0 && order != 1) || n < 0 || dims < 0) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK; // batch_operations.go:65-67
    if (!d_query || !d_flat || !d_results || dims == 0) return LB_ERR_INVALID_ARG;
    if (dims > LB_MAX_DIM) return LB_ERR_UNSUPPORTED;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    try {
        LB_HIP(hipSetDevice(device));
        hipStream_t s = (hipStream_t)stream;
        Lease qna(device, 16);
        if (metric == LB_METRIC_COSINE) launch_query_norms(order, d_query, nullptr, 1, dims, qna.as<float>(), s);
        CandState cs{};
        launch_scan(metric, order, /*raw_dot=*/true, d_flat, 0, n, dims, d_query, nullptr, 1, qna.as<float>(), nullptr, nullptr,
                    cs, false, d_results, n, s);
        LB_LAUNCH_CHECK();
        LB_HIP(hipStreamSynchronize(s));
    } catch (const HipErr &e) {
        return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
    }
    return LB_OK;
}

int lb_simd_distance_batch_flat(int device, int metric, int order, const float *query, const float *flat,
                                int64_t n, int dims, float *results)
{
    if (metric < 0 || metric > 2 || (order != 0 && order != 1) || n < 0 || dims < 0) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK;
    if (!query || !flat || !results || dims == 0) return LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    try {
        LB_HIP(hipSetDevice(device));
        Lease dq(device, (size_t)dims * 4), dx(device, (size_t)n * dims * 4), dr(device, (size_t)n * 4);
        LB_HIP(hipMemcpy(dq.p, query, (size_t)dims * 4, hipMemcpyHostToDevice));
        LB_HIP(hipMemcpy(dx.p, flat, (size_t)n * dims * 4, hipMemcpyHostToDevice));
        const int rc = lb_simd_distance_batch_flat_device(device, metric, order, dq.as<float>(), dx.as<float>(), n, dims,
                                                          dr.as<float>(), nullptr);
        if (rc != LB_OK) return rc;
        LB_HIP(hipMemcpy(results, dr.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    } catch (const HipErr &e) {
        return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
    }
    return LB_OK;
}

// simd.EuclideanDistanceBatch / CosineDistanceBatch / DotProductBatch over [][]float32
// (internal/simd/batch_operations.go:29-60,131-157; per-vector rules in include/longbow_gpu.h)
int lb_simd_distance_batch(int device, int metric, int order, const float *query, int dims, const float *const *vectors,
                           const int *lens, int64_t n, float *results)
{
    if (metric < 0 || metric > 2 || (order != 0 && order != 1) || n < 0 || dims < 0) return LB_ERR_INVALID_ARG;
    if (n == 0) return LB_OK; // batch_operations.go:33-35,132-134
    if (!vectors || !lens || !results || (dims > 0 && !query)) return LB_ERR_INVALID_ARG;
    if (dims > LB_MAX_DIM) return LB_ERR_UNSUPPORTED;
    // which vectors are scored
    std::vector<int64_t> live;
    try {
        live.reserve((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            const bool ok = vectors[i] != nullptr && lens[i] == dims;
            if (metric == LB_METRIC_EUCLIDEAN) {
                if (ok) live.push_back(i);
                else results[i] = FLT_MAX; // math.MaxFloat32 (batch_operations.go:39-42,51)
            } else {
                if (vectors[i] == nullptr) continue; // skipped: results[i] keeps the caller's value (simd.go:243-245,256-258)
                if (lens[i] != dims) break;          // the loop returns its error here and the wrapper swallows it (:140,155)
                live.push_back(i);
            }
        }
    } catch (...) {
        return LB_ERR_OOM;
    }
    const int64_t m = (int64_t)live.size();
    if (m == 0) return LB_OK;
    if (dims == 0) { // len 0: 0 (Euclidean, dot) / 1.0 (cosine)  (simd.go:131-163)
        for (int64_t i : live) results[i] = metric == LB_METRIC_COSINE ? 1.0f : 0.0f;
        return LB_OK;
    }
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    try {
        LB_HIP(hipSetDevice(device));
        const size_t row = (size_t)dims * 4;
        Lease hx(device, (size_t)m * row, true), dq(device, row), dx(device, (size_t)m * row), dr(device, (size_t)m * 4),
            hr(device, (size_t)m * 4, true);
        for (int64_t j = 0; j < m; j++) std::memcpy(hx.as<char>() + (size_t)j * row, vectors[live[(size_t)j]], row);
        LB_HIP(hipMemcpy(dq.p, query, row, hipMemcpyHostToDevice));
        LB_HIP(hipMemcpy(dx.p, hx.p, (size_t)m * row, hipMemcpyHostToDevice));
        const int rc = lb_simd_distance_batch_flat_device(device, metric, order, dq.as<float>(), dx.as<float>(), m, dims,
                                                          dr.as<float>(), nullptr);
        if (rc != LB_OK) return rc;
        LB_HIP(hipMemcpy(hr.p, dr.p, (size_t)m * 4, hipMemcpyDeviceToHost));
        for (int64_t j = 0; j < m; j++) results[live[(size_t)j]] = hr.as<float>()[j];
    } catch (const HipErr &e) {
        (void)hipGetLastError();
        return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
    } catch (...) {
        return LB_ERR_INTERNAL;
    }
    return LB_OK;
}

// ---- merge / fill -----------------------------------------------------------------
int lb_gpu_merge_topk_device(int device, int nshards, int64_t nq, int k, const float *d_dist_in,
                             const int64_t *d_labels_in, float *d_dist_out, int64_t *d_labels_out, void *stream)
{
    if (nshards <= 0 || nq < 0 || k <= 0 || (int64_t)nshards * k > 16384) return LB_ERR_INVALID_ARG;
    if (nq == 0) return LB_OK;
    if (!d_dist_in || !d_labels_in || !d_dist_out || !d_labels_out) return LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LB_ERR_HIP;
    launch_merge_topk(nshards, nq, k, d_dist_in, d_labels_in, nq * k, nq * k, d_dist_out, d_labels_out,
                      (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return LB_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? LB_OK : LB_ERR_HIP;
}

int lb_gpu_merge_topk_packed_device(int device, int nshards, int64_t nq, int k, const void *d_packed,
                                    float *d_dist_out, int64_t *d_labels_out, void *stream)
{
    if (nshards <= 0 || nq < 0 || k <= 0 || (int64_t)nshards * k > 16384) return LB_ERR_INVALID_ARG;
    if (nq == 0) return LB_OK;
    if (!d_packed || !d_dist_out || !d_labels_out) return LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LB_ERR_HIP;
    // per shard: nq*k int64 labels followed by nq*k f32 distances (padded to 8 bytes)
    const int64_t nk = nq * k;
    const int64_t block_bytes = nk * 8 + ((nk * 4 + 7) / 8) * 8;
    const char *base = reinterpret_cast<const char *>(d_packed);
    launch_merge_topk(nshards, nq, k, reinterpret_cast<const float *>(base + nk * 8),
                      reinterpret_cast<const int64_t *>(base), block_bytes / 4, block_bytes / 8, d_dist_out,
                      d_labels_out, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return LB_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? LB_OK : LB_ERR_HIP;
}

int lb_gpu_rrf_fuse_device(int device, int64_t nq, int kd, const int64_t *d_dense_ids, int ks,
                           const int64_t *d_sparse_ids, int k, int limit, int64_t *d_out_ids,
                           float *d_out_scores, void *stream)
{
    if (nq < 0 || kd < 0 || ks < 0 || limit <= 0 || kd + ks > 8192) return LB_ERR_INVALID_ARG;
    if (nq == 0) return LB_OK;
    if ((kd > 0 && !d_dense_ids) || (ks > 0 && !d_sparse_ids) || !d_out_ids || !d_out_scores) return LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LB_ERR_HIP;
    launch_rrf(nq, kd, d_dense_ids, ks, d_sparse_ids, k <= 0 ? 60 : k, limit, d_out_ids, d_out_scores, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return LB_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? LB_OK : LB_ERR_HIP;
}

int lb_gpu_rrf_fuse(int device, int64_t nq, int kd, const int64_t *dense_ids, int ks, const int64_t *sparse_ids,
                    int k, int limit, int64_t *out_ids, float *out_scores)
{
    if (nq < 0 || kd < 0 || ks < 0 || limit <= 0 || kd + ks > 8192) return LB_ERR_INVALID_ARG;
    if (nq == 0) return LB_OK;
    if ((kd > 0 && !dense_ids) || (ks > 0 && !sparse_ids) || !out_ids || !out_scores) return LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    try {
        LB_HIP(hipSetDevice(device));
        const size_t bd = (size_t)nq * std::max(kd, 1) * 8, bs = (size_t)nq * std::max(ks, 1) * 8;
        Lease dd(device, bd), ds(device, bs), dout(device, (size_t)nq * limit * 8), dsc(device, (size_t)nq * limit * 4);
        if (kd > 0) LB_HIP(hipMemcpy(dd.p, dense_ids, (size_t)nq * kd * 8, hipMemcpyHostToDevice));
        if (ks > 0) LB_HIP(hipMemcpy(ds.p, sparse_ids, (size_t)nq * ks * 8, hipMemcpyHostToDevice));
        const int rc = lb_gpu_rrf_fuse_device(device, nq, kd, dd.as<int64_t>(), ks, ds.as<int64_t>(), k, limit,
                                              dout.as<int64_t>(), dsc.as<float>(), nullptr);
        if (rc != LB_OK) return rc;
        LB_HIP(hipMemcpy(out_ids, dout.p, (size_t)nq * limit * 8, hipMemcpyDeviceToHost));
        LB_HIP(hipMemcpy(out_scores, dsc.p, (size_t)nq * limit * 4, hipMemcpyDeviceToHost));
    } catch (const HipErr &e) {
        return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
    }
    return LB_OK;
}

int lb_gpu_fill_uniform_device(int device, float *d_dst, int64_t n, uint64_t seed, int64_t offset, void *stream)
{
    if (n < 0 || (n > 0 && !d_dst)) return LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LB_ERR_HIP;
    launch_fill_uniform(d_dst, n, seed, offset, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return LB_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? LB_OK : LB_ERR_HIP;
}

int lb_gpu_fill_uniform_rows_device(int device, float *d_dst, const int64_t *d_ids, int64_t nrows, int dim, uint64_t seed,
                                    void *stream)
{
    if (nrows < 0 || dim <= 0 || (nrows > 0 && (!d_dst || !d_ids))) return LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LB_ERR_HIP;
    launch_fill_uniform_rows(d_dst, d_ids, nrows, dim, seed, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return LB_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? LB_OK : LB_ERR_HIP;
}

double lb_gpu_shader_clock_mhz(int device, int spin_us)
{
    if (spin_us <= 0 || spin_us > 1000000) return -(double)LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return -(double)LB_ERR_NO_DEVICE;
    try {
        LB_HIP(hipSetDevice(device));
        int cus = 0;
        LB_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        if (cus <= 0) cus = 1;
        Lease d(device, 2 * sizeof(unsigned long long)), hbuf(device, 2 * sizeof(unsigned long long), true);
        LB_HIP(hipMemset(d.p, 0, 2 * sizeof(unsigned long long)));
        hipLaunchKernelGGL(clock_probe_kernel, dim3((unsigned)cus), dim3(64), 0, nullptr, d.as<unsigned long long>(),
                           (unsigned long long)spin_us * 100ull);
        LB_LAUNCH_CHECK();
        LB_HIP(hipMemcpy(hbuf.p, d.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        const unsigned long long *v = hbuf.as<unsigned long long>();
        if (v[1] == 0) return -(double)LB_ERR_INTERNAL;
        return 100.0 * (double)v[0] / (double)v[1];
    } catch (const HipErr &e) {
        (void)hipGetLastError();
        return -(double)(e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP);
    } catch (...) {
        return -(double)LB_ERR_INTERNAL;
    }
}

int lb_gpu_fill_codes_device(int device, uint8_t *d_dst, int64_t n, uint64_t seed, int64_t offset, void *stream)
{
    if (n < 0 || (n > 0 && !d_dst)) return LB_ERR_INVALID_ARG;
    if (!device_ok(device)) return LB_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LB_ERR_HIP;
    launch_fill_codes(d_dst, n, seed, offset, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return LB_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? LB_OK : LB_ERR_HIP;
}

} // extern "C"
