// comm.hip -- multi-GPU search through the C ABI: shard search + ONE all-gather of the packed per-shard
// top-k + device merge, without torch.
//
// Semantics: ShardedHNSW.SearchVectors (internal/store/sharded_hnsw.go:414-503: per-shard search, concat,
// sort by Score, truncate) / GlobalSearch + MergeSortedStreams (internal/store/global_search.go:88-251,
// result_merger.go:34-101).  The reference has no collective library at all; here the
// exchange is RCCL's all-gather over xGMI (B*k*12 bytes per rank: latency-bound), or a host-supplied
// all-gather for hosts that bring their own transport (gloo in the CPU-side tests, gRPC in a cluster).
//
// Three ways to stand a communicator up:
//   lb_gpu_comm_init_all   one process drives all GPUs of the node (the Go server): ncclCommInitAll
//   lb_gpu_comm_init_rank  one process (or thread) per GPU: ncclCommInitRank with a unique id the host ships
//   lb_gpu_comm_init_host  any number of ranks, exchange through a host callback (staged via pinned memory)
// RCCL is bound with dlopen at first use: the library itself has no link-time dependency on it.
#include "../../include/longbow_gpu.h"
#include "lb_device.h"
#include "lb_host.h"

#include <dlfcn.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace lb;

namespace {

// ---- the slice of the RCCL API this file uses (nccl.h is not included: no link-time dependency) ----
struct NcclUniqueId { char internal[128]; };
typedef void *NcclComm;
enum { kNcclSuccess = 0, kNcclChar = 0 };
struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*CommInitAll)(NcclComm *, int, const int *) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr; // optional (point-to-point)
    int (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
        if (!r.so) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.so, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.so, "ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.so, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.so, "ncclAllGather"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.so, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.so, "ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.so, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.so, "ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd;
    });
    return r;
}

inline size_t block_bytes(int64_t nq, int k)
{
    const int64_t nk = nq * k;
    return (size_t)(nk * 8 + ((nk * 4 + 7) / 8) * 8);
}

// What travels per rank: the packed block (lb_gpu_merge_topk_packed_device's layout) + one 8-byte status word
// (0 = the rank's shard search succeeded).  A rank whose local search fails still takes part in the exchange, with
// the canonical empty block (label -1 / FLT_MAX) and its lb_status in the word, so no peer is left waiting in the
// collective and every rank reports the failure.
constexpr size_t kStatusBytes = 8;

// per-device state of a communicator (one entry for init_rank / init_host, ndev entries for init_all)
struct Peer {
    int device = 0;
    NcclComm comm = nullptr;
    hipStream_t stream = nullptr;
    void *d_mine = nullptr, *d_all = nullptr; // packed blocks
    size_t mine_bytes = 0, all_bytes = 0;
    float *d_q = nullptr, *d_dist = nullptr;  // init_all: host-pointer staging
    int64_t *d_lab = nullptr;
    size_t q_bytes = 0, out_n = 0;
};

} // namespace

struct lb_gpu_comm {
    int nranks = 1, rank = 0;
    int mode = 0; // 0 = RCCL one rank per handle, 1 = host transport, 2 = RCCL all devices in this process
    lb_allgather_fn fn = nullptr;
    void *fn_ctx = nullptr;
    std::vector<Peer> peers;
    void *h_send = nullptr, *h_recv = nullptr; // pinned (host transport)
    uint64_t *h_status = nullptr;              // pinned: this rank's status word on its way into the packed block
    size_t h_send_bytes = 0, h_recv_bytes = 0;
    std::mutex mu; // one search at a time per communicator (a collective is an ordered sequence)
    std::string last_error;
};

namespace {

void set_err(lb_gpu_comm *c, const char *what, int code)
{
    char buf[256];
    const char *s = (rccl().ok && rccl().GetErrorString) ? rccl().GetErrorString(code) : "";
    snprintf(buf, sizeof buf, "%s failed (%d %s)", what, code, s ? s : "");
    c->last_error = buf;
}

void ensure_blocks(Peer &p, int nranks, int64_t nq, int k)
{
    const size_t bb = block_bytes(nq, k) + kStatusBytes;
    if (p.mine_bytes < bb) {
        if (p.d_mine) (void)hipFree(p.d_mine);
        p.d_mine = nullptr;
        p.mine_bytes = 0;
        LB_HIP(hipMalloc(&p.d_mine, bb));
        p.mine_bytes = bb;
    }
    if (p.all_bytes < bb * nranks) {
        if (p.d_all) (void)hipFree(p.d_all);
        p.d_all = nullptr;
        p.all_bytes = 0;
        LB_HIP(hipMalloc(&p.d_all, bb * nranks));
        p.all_bytes = bb * nranks;
    }
}

void free_peer(Peer &p)
{
    (void)hipSetDevice(p.device);
    if (p.comm && rccl().ok) (void)rccl().CommDestroy(p.comm);
    if (p.d_mine) (void)hipFree(p.d_mine);
    if (p.d_all) (void)hipFree(p.d_all);
    if (p.d_q) (void)hipFree(p.d_q);
    if (p.d_dist) (void)hipFree(p.d_dist);
    if (p.d_lab) (void)hipFree(p.d_lab);
    if (p.stream) (void)hipStreamDestroy(p.stream);
}

lb_gpu_comm *new_comm(int nranks, int rank, int mode, int ndev, const int *devices, int *out_status)
{
    auto st = [&](int v) { if (out_status) *out_status = v; };
    auto *c = new (std::nothrow) lb_gpu_comm();
    if (!c) { st(LB_ERR_OOM); return nullptr; }
    c->nranks = nranks; c->rank = rank; c->mode = mode;
    c->peers.resize((size_t)ndev);
    for (int i = 0; i < ndev; i++) {
        c->peers[(size_t)i].device = devices[i];
        if (hipSetDevice(devices[i]) != hipSuccess ||
            hipStreamCreateWithFlags(&c->peers[(size_t)i].stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            st(LB_ERR_HIP);
            lb_gpu_comm_free(c);
            return nullptr;
        }
    }
    st(LB_OK);
    return c;
}

} // namespace

extern "C" {

int lb_gpu_comm_get_unique_id(void *out128)
{
    if (!out128) return LB_ERR_INVALID_ARG;
    if (!rccl().ok) return LB_ERR_UNSUPPORTED; // RCCL not loadable here
    NcclUniqueId id;
    if (rccl().GetUniqueId(&id) != kNcclSuccess) return LB_ERR_HIP;
    std::memcpy(out128, &id, sizeof id);
    return LB_OK;
}

lb_gpu_comm *lb_gpu_comm_init_rank(int device, int nranks, int rank, const void *unique_id, int *out_status)
{
    auto st = [&](int v) { if (out_status) *out_status = v; };
    if (nranks <= 0 || rank < 0 || rank >= nranks || !unique_id) { st(LB_ERR_INVALID_ARG); return nullptr; }
    if (!device_ok(device)) { st(LB_ERR_NO_DEVICE); return nullptr; }
    if (!rccl().ok) { st(LB_ERR_UNSUPPORTED); return nullptr; }
    lb_gpu_comm *c = new_comm(nranks, rank, 0, 1, &device, out_status);
    if (!c) return nullptr;
    NcclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    (void)hipSetDevice(device);
    const int rc = rccl().CommInitRank(&c->peers[0].comm, nranks, id, rank);
    if (rc != kNcclSuccess) {
        c->peers[0].comm = nullptr;
        st(LB_ERR_HIP);
        lb_gpu_comm_free(c);
        return nullptr;
    }
    return c;
}

lb_gpu_comm *lb_gpu_comm_init_host(int device, int nranks, int rank, lb_allgather_fn fn, void *ctx, int *out_status)
{
    auto st = [&](int v) { if (out_status) *out_status = v; };
    if (nranks <= 0 || rank < 0 || rank >= nranks || (nranks > 1 && !fn)) { st(LB_ERR_INVALID_ARG); return nullptr; }
    if (!device_ok(device)) { st(LB_ERR_NO_DEVICE); return nullptr; }
    lb_gpu_comm *c = new_comm(nranks, rank, 1, 1, &device, out_status);
    if (!c) return nullptr;
    c->fn = fn;
    c->fn_ctx = ctx;
    return c;
}

lb_gpu_comm *lb_gpu_comm_init_all(int ndev, const int *devices, int *out_status)
{
    auto st = [&](int v) { if (out_status) *out_status = v; };
    if (ndev <= 0 || ndev > 64) { st(LB_ERR_INVALID_ARG); return nullptr; }
    std::vector<int> devs((size_t)ndev);
    for (int i = 0; i < ndev; i++) {
        devs[(size_t)i] = devices ? devices[i] : i;
        if (!device_ok(devs[(size_t)i])) { st(LB_ERR_NO_DEVICE); return nullptr; }
        for (int j = 0; j < i; j++)
            if (devs[(size_t)j] == devs[(size_t)i]) { st(LB_ERR_INVALID_ARG); return nullptr; } // one rank per GPU
    }
    if (ndev > 1 && !rccl().ok) { st(LB_ERR_UNSUPPORTED); return nullptr; }
    lb_gpu_comm *c = new_comm(ndev, 0, 2, ndev, devs.data(), out_status);
    if (!c) return nullptr;
    if (ndev > 1) {
        std::vector<NcclComm> comms((size_t)ndev, nullptr);
        const int rc = rccl().CommInitAll(comms.data(), ndev, devs.data());
        if (rc != kNcclSuccess) {
            st(LB_ERR_HIP);
            lb_gpu_comm_free(c);
            return nullptr;
        }
        for (int i = 0; i < ndev; i++) c->peers[(size_t)i].comm = comms[(size_t)i];
    }
    return c;
}

void lb_gpu_comm_free(lb_gpu_comm *c)
{
    if (!c) return;
    for (auto &p : c->peers) free_peer(p);
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    if (c->h_status) (void)hipHostFree(c->h_status);
    delete c;
}

int lb_gpu_comm_nranks(const lb_gpu_comm *c) { return c ? c->nranks : 0; }
int lb_gpu_comm_rank(const lb_gpu_comm *c) { return c ? c->rank : -1; }
const char *lb_gpu_comm_last_error(const lb_gpu_comm *c) { return c ? c->last_error.c_str() : "null handle"; }

// pinned host blocks of the host transport, sized for a block of `bs` bytes per rank
static void ensure_host_blocks(lb_gpu_comm *c, size_t bs)
{
    if (c->h_send_bytes < bs) {
        if (c->h_send) (void)hipHostFree(c->h_send);
        c->h_send = nullptr;
        c->h_send_bytes = 0;
        LB_HIP(hipHostMalloc(&c->h_send, bs, hipHostMallocDefault));
        c->h_send_bytes = bs;
    }
    if (c->h_recv_bytes < bs * (size_t)c->nranks) {
        if (c->h_recv) (void)hipHostFree(c->h_recv);
        c->h_recv = nullptr;
        c->h_recv_bytes = 0;
        LB_HIP(hipHostMalloc(&c->h_recv, bs * (size_t)c->nranks, hipHostMallocDefault));
        c->h_recv_bytes = bs * (size_t)c->nranks;
    }
}

int lb_gpu_comm_prepare(lb_gpu_comm *c, int64_t nq_max, int k_max)
{
    if (!c || nq_max <= 0 || k_max <= 0) return LB_ERR_INVALID_ARG;
    if (k_max > 2048 || (int64_t)c->nranks * k_max > 16384) return LB_ERR_UNSUPPORTED;
    if (nq_max > ((int64_t)1 << 32) / k_max) return LB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    try {
        const size_t bs = block_bytes(nq_max, k_max) + kStatusBytes;
        for (size_t i = 0; i < c->peers.size(); i++) {
            Peer &p = c->peers[i];
            LB_HIP(hipSetDevice(p.device));
            ensure_blocks(p, (c->mode == 2 && i != 0) ? 1 : c->nranks, nq_max, k_max);
        }
        if (c->mode == 1) ensure_host_blocks(c, bs);
        if (!c->h_status) LB_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_status), sizeof(uint64_t), hipHostMallocDefault));
    } catch (const HipErr &e) {
        (void)hipGetLastError();
        c->last_error = std::string("HIP error in ") + e.what + " (lb_gpu_comm_prepare)";
        return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
    } catch (...) {
        return LB_ERR_INTERNAL;
    }
    return LB_OK;
}

int lb_gpu_comm_search_device(lb_gpu_comm *c, lb_gpu_index *h, int64_t nq, const float *d_queries, int k, float *d_dist,
                              int64_t *d_labels, void *stream)
{
    if (!c || !h || nq < 0 || k <= 0 || (nq > 0 && (!d_queries || !d_dist || !d_labels))) return LB_ERR_INVALID_ARG;
    if (c->mode == 2) return LB_ERR_INVALID_ARG; // use lb_gpu_comm_search_all
    if (nq == 0) return LB_OK;
    // (argument errors that every rank sees alike may return before the exchange; anything that can differ from
    // rank to rank -- a failed shard search, an allocation -- must not: the peers would wait in the all-gather forever)
    if ((int64_t)c->nranks * k > 16384) { c->last_error = "nranks * k exceeds 16384"; return LB_ERR_INVALID_ARG; }
    if (k > 2048) { c->last_error = "k exceeds the supported maximum 2048"; return LB_ERR_UNSUPPORTED; }
    std::lock_guard<std::mutex> g(c->mu);
    Peer &p = c->peers[0];
    int local_rc = LB_OK;
    try {
        LB_HIP(hipSetDevice(p.device));
        hipStream_t s = stream ? (hipStream_t)stream : p.stream;
        const int64_t nk = nq * k;
        const size_t bb = block_bytes(nq, k), bs = bb + kStatusBytes;
        try { // (no-ops after lb_gpu_comm_prepare with this size: nothing below allocates before the exchange)
            ensure_blocks(p, c->nranks, nq, k);
            if (c->mode == 1 && c->nranks > 1) ensure_host_blocks(c, bs);
            if (!c->h_status) LB_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_status), sizeof(uint64_t), hipHostMallocDefault));
        } catch (const HipErr &e) {
            // no room for the exchange buffers of a search larger than what lb_gpu_comm_prepare sized: this rank cannot take
            // part at all.  The peers' all-gather will fail or time out at the transport level; nothing here can repair that.
            (void)hipGetLastError();
            c->last_error = std::string("HIP error in ") + e.what + " (exchange buffers: call lb_gpu_comm_prepare after init)";
            return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
        }
        char *mine = static_cast<char *>(p.d_mine);
        // local shard: labels | distances written straight into this rank's packed block
        local_rc = lb_gpu_index_search_device(h, nq, d_queries, k, reinterpret_cast<float *>(mine + nk * 8),
                                              reinterpret_cast<int64_t *>(mine), s);
        if (local_rc != LB_OK) {
            // still take part in the exchange, with the canonical empty block and the status word set
            c->last_error = lb_gpu_last_error(h);
            launch_fill_empty(reinterpret_cast<float *>(mine + nk * 8), reinterpret_cast<int64_t *>(mine), nk, s);
        }
        *c->h_status = (uint64_t)(uint32_t)local_rc; // (pinned word: the copy below stages nothing)
        LB_HIP(hipMemcpyAsync(mine + bb, c->h_status, sizeof(uint64_t), hipMemcpyHostToDevice, s));
        if (c->nranks == 1) {
            LB_HIP(hipMemcpyAsync(p.d_all, p.d_mine, bs, hipMemcpyDeviceToDevice, s));
        } else if (c->mode == 0) { // RCCL over xGMI: the one exchange step of the path
            const int nrc = rccl().AllGather(p.d_mine, p.d_all, bs, kNcclChar, p.comm, s);
            if (nrc != kNcclSuccess) { set_err(c, "ncclAllGather", nrc); return LB_ERR_HIP; }
        } else { // host transport: D2H, the host's all-gather, H2D
            LB_HIP(hipMemcpyAsync(c->h_send, p.d_mine, bs, hipMemcpyDeviceToHost, s));
            LB_HIP(hipStreamSynchronize(s));
            const int frc = c->fn(c->fn_ctx, c->h_send, c->h_recv, bs);
            if (frc != 0) { set_err(c, "host all-gather callback", frc); return LB_ERR_INTERNAL; }
            LB_HIP(hipMemcpyAsync(p.d_all, c->h_recv, bs * c->nranks, hipMemcpyHostToDevice, s));
        }
        const char *all = static_cast<const char *>(p.d_all);
        launch_merge_topk(c->nranks, nq, k, reinterpret_cast<const float *>(all + nk * 8), reinterpret_cast<const int64_t *>(all),
                          (int64_t)(bs / 4), (int64_t)(bs / 8), d_dist, d_labels, s);
        LB_LAUNCH_CHECK();
        // every rank's status word: a failure anywhere fails the search everywhere (results would lack a shard)
        std::vector<uint64_t> st((size_t)c->nranks, 0);
        LB_HIP(hipMemcpy2DAsync(st.data(), sizeof(uint64_t), all + bb, bs, sizeof(uint64_t), (size_t)c->nranks,
                                hipMemcpyDeviceToHost, s));
        LB_HIP(hipStreamSynchronize(s));
        if (local_rc != LB_OK) return local_rc;
        for (int r = 0; r < c->nranks; r++)
            if (st[(size_t)r] != 0) {
                char buf[96];
                snprintf(buf, sizeof buf, "shard search failed on rank %d (status %d)", r, (int)st[(size_t)r]);
                c->last_error = buf;
                return (int)st[(size_t)r];
            }
    } catch (const HipErr &e) {
        (void)hipGetLastError();
        c->last_error = std::string("HIP error in ") + e.what;
        return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
    } catch (...) {
        c->last_error = "internal error (exception)";
        return LB_ERR_INTERNAL;
    }
    return LB_OK;
}

int lb_gpu_comm_search_all(lb_gpu_comm *c, lb_gpu_index *const *shards, int64_t nq, const float *queries, int k, float *dist,
                           int64_t *labels)
{
    if (!c || !shards || nq < 0 || k <= 0 || (nq > 0 && (!queries || !dist || !labels))) return LB_ERR_INVALID_ARG;
    if (c->mode != 2) return LB_ERR_INVALID_ARG;
    if (nq == 0) return LB_OK;
    const int nd = c->nranks;
    if ((int64_t)nd * k > 16384) { c->last_error = "ndev * k exceeds 16384"; return LB_ERR_INVALID_ARG; }
    if (k > 2048) { c->last_error = "k exceeds the supported maximum 2048"; return LB_ERR_UNSUPPORTED; }
    for (int i = 0; i < nd; i++)
        if (!shards[i]) return LB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    const int dim = lb_gpu_index_dim(shards[0]);
    for (int i = 0; i < nd; i++) { // shard i must live on the communicator's i-th device and share the dimension
        if (lb_gpu_index_device(shards[i]) != c->peers[(size_t)i].device) {
            char buf[128];
            snprintf(buf, sizeof buf, "shard %d lives on device %d, the communicator's rank %d is device %d", i,
                     lb_gpu_index_device(shards[i]), i, c->peers[(size_t)i].device);
            c->last_error = buf;
            return LB_ERR_INVALID_ARG;
        }
        if (lb_gpu_index_dim(shards[i]) != dim) { c->last_error = "shards differ in dimension"; return LB_ERR_INVALID_ARG; }
    }
    const int64_t nk = nq * k;
    const size_t bb = block_bytes(nq, k);
    const size_t qb = (size_t)nq * dim * sizeof(float);
    std::vector<int> rcs((size_t)nd, LB_OK);
    try {
        // one host thread per device: upload the queries, search the shard into its packed block
        auto local = [&](int i) {
            Peer &p = c->peers[(size_t)i];
            try {
                LB_HIP(hipSetDevice(p.device));
                ensure_blocks(p, i == 0 ? nd : 1, nq, k); // only the merging device holds every block
                if (p.q_bytes < qb) {
                    if (p.d_q) (void)hipFree(p.d_q);
                    p.d_q = nullptr;
                    p.q_bytes = 0;
                    LB_HIP(hipMalloc(&p.d_q, qb));
                    p.q_bytes = qb;
                }
                LB_HIP(hipMemcpyAsync(p.d_q, queries, qb, hipMemcpyHostToDevice, p.stream));
                char *mine = static_cast<char *>(p.d_mine);
                rcs[(size_t)i] = lb_gpu_index_search_device(shards[i], nq, p.d_q, k, reinterpret_cast<float *>(mine + nk * 8),
                                                            reinterpret_cast<int64_t *>(mine), p.stream);
            } catch (const HipErr &e) {
                (void)hipGetLastError();
                rcs[(size_t)i] = e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
            } catch (...) {
                rcs[(size_t)i] = LB_ERR_INTERNAL;
            }
        };
        {
            std::vector<std::thread> th;
            for (int i = 1; i < nd; i++) th.emplace_back(local, i);
            local(0);
            for (auto &t : th) t.join();
        }
        // all shard searches are host-synchronous and this one process drives every device: a failure is known HERE,
        // before anything collective has been issued, so returning now strands nobody
        for (int i = 0; i < nd; i++)
            if (rcs[(size_t)i] != LB_OK) {
                c->last_error = std::string("shard search failed: ") + lb_gpu_last_error(shards[i]);
                return rcs[(size_t)i];
            }
        Peer &p0 = c->peers[0];
        LB_HIP(hipSetDevice(p0.device));
        const size_t bs = bb + kStatusBytes; // (block stride of d_all, as in search_device; the status words stay unused)
        LB_HIP(hipMemcpyAsync(p0.d_all, p0.d_mine, bb, hipMemcpyDeviceToDevice, p0.stream));
        if (nd > 1) {
            // The answer is needed on ONE device (host results out), so the blocks are gathered to it and nowhere else:
            // grouped ncclSend / ncclRecv over xGMI, (nd - 1) x nq*k*12 bytes in all (an all-gather would move nd times
            // that and leave nd - 1 copies unread).  RCCL builds without point-to-point: the all-gather.
            int nrc = rccl().GroupStart();
            if (rccl().Send && rccl().Recv) {
                for (int i = 1; i < nd && nrc == kNcclSuccess; i++) {
                    Peer &p = c->peers[(size_t)i];
                    nrc = rccl().Send(p.d_mine, bb, kNcclChar, 0, p.comm, p.stream);
                    if (nrc == kNcclSuccess)
                        nrc = rccl().Recv(static_cast<char *>(p0.d_all) + (size_t)i * bs, bb, kNcclChar, i, p0.comm, p0.stream);
                }
            } else {
                for (int i = 0; i < nd && nrc == kNcclSuccess; i++) {
                    Peer &p = c->peers[(size_t)i];
                    if (i != 0) { // every device needs room for every block in this form
                        LB_HIP(hipSetDevice(p.device));
                        ensure_blocks(p, nd, nq, k);
                    }
                    nrc = rccl().AllGather(p.d_mine, p.d_all, bs, kNcclChar, p.comm, p.stream);
                }
                LB_HIP(hipSetDevice(p0.device));
            }
            const int erc = rccl().GroupEnd();
            if (nrc != kNcclSuccess || erc != kNcclSuccess) { set_err(c, "RCCL gather", nrc != kNcclSuccess ? nrc : erc); return LB_ERR_HIP; }
        }
        if (p0.out_n < (size_t)nk) {
            if (p0.d_dist) (void)hipFree(p0.d_dist);
            if (p0.d_lab) (void)hipFree(p0.d_lab);
            p0.d_dist = nullptr;
            p0.d_lab = nullptr;
            p0.out_n = 0;
            LB_HIP(hipMalloc(&p0.d_dist, (size_t)nk * sizeof(float)));
            LB_HIP(hipMalloc(&p0.d_lab, (size_t)nk * sizeof(int64_t)));
            p0.out_n = (size_t)nk;
        }
        const char *all = static_cast<const char *>(p0.d_all);
        launch_merge_topk(nd, nq, k, reinterpret_cast<const float *>(all + nk * 8), reinterpret_cast<const int64_t *>(all),
                          (int64_t)(bs / 4), (int64_t)(bs / 8), p0.d_dist, p0.d_lab, p0.stream);
        LB_LAUNCH_CHECK();
        LB_HIP(hipMemcpyAsync(dist, p0.d_dist, (size_t)nk * sizeof(float), hipMemcpyDeviceToHost, p0.stream));
        LB_HIP(hipMemcpyAsync(labels, p0.d_lab, (size_t)nk * sizeof(int64_t), hipMemcpyDeviceToHost, p0.stream));
        LB_HIP(hipStreamSynchronize(p0.stream));
        for (int i = 1; i < nd; i++) { // leave no work in flight on the other devices
            LB_HIP(hipSetDevice(c->peers[(size_t)i].device));
            LB_HIP(hipStreamSynchronize(c->peers[(size_t)i].stream));
        }
    } catch (const HipErr &e) {
        (void)hipGetLastError();
        c->last_error = std::string("HIP error in ") + e.what;
        return e.e == hipErrorOutOfMemory ? LB_ERR_OOM : LB_ERR_HIP;
    } catch (...) {
        c->last_error = "internal error (exception)";
        return LB_ERR_INTERNAL;
    }
    return LB_OK;
}

} // extern "C"
