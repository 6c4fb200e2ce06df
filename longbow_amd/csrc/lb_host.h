// lb_host.h -- host-side helpers shared by the C-ABI translation units (index.hip, pq.hip, comm.hip):
// HIP error plumbing and a pooled device / pinned-host buffer cache.
//
// Why a pool: hipMalloc costs 0.1-0.3 ms and hipFree synchronises the whole device, so an entry
// point that allocates per call stalls every concurrent search on the same GPU.  The host-pointer
// entry points (lb_simd_*, lb_gpu_index_filter_*, lb_gpu_rrf_fuse, lb_gpu_index_rerank, PQ) borrow
// their staging buffers from this cache instead.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <mutex>
#include <vector>

namespace lb {

struct HipErr {
    hipError_t e;
    const char *what;
};

#define LB_HIP(call)                                          \
    do {                                                      \
        hipError_t _e = (call);                               \
        if (_e != hipSuccess) throw ::lb::HipErr{_e, #call};  \
    } while (0)

// A kernel launch with an illegal configuration (too much LDS, bad grid) is only reported by
// hipGetLastError; stream synchronisation returns success.  Every public entry point calls this
// once after its launches and before it reports LB_OK.
#define LB_LAUNCH_CHECK()                                                  \
    do {                                                                   \
        hipError_t _e = hipGetLastError();                                 \
        if (_e != hipSuccess) throw ::lb::HipErr{_e, "kernel launch"};     \
    } while (0)

class BufPool {
  public:
    // device memory (pinned == false) on `device`, or pinned host memory; at least `bytes` long
    void *get(int device, size_t bytes, bool pinned)
    {
        if (bytes == 0) bytes = 1;
        {
            std::lock_guard<std::mutex> g(mu_);
            size_t best = (size_t)-1;
            for (size_t i = 0; i < free_.size(); i++) {
                const Ent &e = free_[i];
                if (e.device != device || e.pinned != pinned || e.bytes < bytes || e.bytes > 4 * bytes + (1u << 20)) continue;
                if (best == (size_t)-1 || e.bytes < free_[best].bytes) best = i;
            }
            if (best != (size_t)-1) {
                Ent e = free_[best];
                free_.erase(free_.begin() + (long)best);
                live_.push_back(e);
                return e.p;
            }
        }
        Ent e;
        e.device = device;
        e.pinned = pinned;
        e.bytes = (bytes + 65535) & ~(size_t)65535;
        e.p = nullptr;
        hipError_t rc = pinned ? hipHostMalloc(&e.p, e.bytes, hipHostMallocDefault) : hipMalloc(&e.p, e.bytes);
        if (rc != hipSuccess) {
            trim(device); // give cached buffers back and retry once
            rc = pinned ? hipHostMalloc(&e.p, e.bytes, hipHostMallocDefault) : hipMalloc(&e.p, e.bytes);
            if (rc != hipSuccess) throw HipErr{rc, pinned ? "hipHostMalloc (pool)" : "hipMalloc (pool)"};
        }
        std::lock_guard<std::mutex> g(mu_);
        live_.push_back(e);
        return e.p;
    }
    void put(void *p)
    {
        if (!p) return;
        Ent e{};
        bool drop = false;
        {
            std::lock_guard<std::mutex> g(mu_);
            size_t i = 0;
            for (; i < live_.size(); i++)
                if (live_[i].p == p) break;
            if (i == live_.size()) return; // not ours
            e = live_[i];
            live_.erase(live_.begin() + (long)i);
            size_t cached = 0, cached_bytes = 0;
            for (const Ent &f : free_)
                if (f.device == e.device && f.pinned == e.pinned) { cached++; cached_bytes += f.bytes; }
            // bounded cache: at most 16 buffers / 2 GiB per (device, kind)
            drop = cached >= 16 || cached_bytes + e.bytes > ((size_t)2 << 30);
            if (!drop) free_.push_back(e);
        }
        if (drop) {
            if (e.pinned) (void)hipHostFree(e.p);
            else (void)hipFree(e.p);
        }
    }
    // release every cached (not in use) buffer of `device` (-1: all devices)
    void trim(int device)
    {
        std::vector<Ent> out;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t i = 0; i < free_.size();) {
                if (device < 0 || free_[i].device == device) {
                    out.push_back(free_[i]);
                    free_.erase(free_.begin() + (long)i);
                } else {
                    i++;
                }
            }
        }
        for (const Ent &e : out) {
            if (e.pinned) (void)hipHostFree(e.p);
            else (void)hipFree(e.p);
        }
    }

  private:
    struct Ent {
        int device;
        bool pinned;
        void *p;
        size_t bytes;
    };
    std::mutex mu_;
    std::vector<Ent> free_, live_;
};

BufPool &buf_pool(); // one per process (index.hip)

// RAII lease from the pool
struct Lease {
    void *p = nullptr;
    Lease() = default;
    Lease(int device, size_t bytes, bool pinned = false) : p(buf_pool().get(device, bytes, pinned)) {}
    Lease(const Lease &) = delete;
    Lease &operator=(const Lease &) = delete;
    ~Lease() { buf_pool().put(p); }
    void reset(int device, size_t bytes, bool pinned = false)
    {
        buf_pool().put(p);
        p = nullptr;
        p = buf_pool().get(device, bytes, pinned);
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

} // namespace lb

// context.Context's cancellation half for ONE call (include/longbow_gpu.h: lb_cancel_*).  The reference checks
// ctx.Err() every 1000 rows of its scan (internal/store/adaptive_index.go:182); here the host checks before every
// kernel launch of a search (a launch covers at most one pass over <= 2.5M rows / one PQ query), stops enqueuing,
// drains the stream and returns LB_ERR_CANCELLED / LB_ERR_DEADLINE.
struct lb_cancel {
    std::atomic<int> fired{0};
    std::atomic<long long> deadline_ns{0}; // steady clock; 0 = none
};

namespace lb {

struct CtxErr {
    int code;
};
inline int ctx_state(const lb_cancel *c)
{
    if (!c) return 0;
    if (c->fired.load(std::memory_order_relaxed)) return 8; // LB_ERR_CANCELLED
    const long long d = c->deadline_ns.load(std::memory_order_relaxed);
    if (d != 0 && std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() >= d)
        return 9; // LB_ERR_DEADLINE
    return 0;
}
inline void ctx_check(const lb_cancel *c)
{
    const int st = ctx_state(c);
    if (st) throw CtxErr{st};
}

// ---- combining of concurrent host-pointer searches (lb_gpu_index_search, lb_gpu_pq_search) --------------------------------
// The reference's gpu.Index.Search is ONE query per call, from many goroutines (internal/gpu/faiss_gpu.go:108-145).  Served as
// they come, T overlapping calls stream the corpus T times.  Instead: one or two callers search at once; a caller that arrives
// while kLanes searches are on the device queues, and when one of them ends the first in the queue answers everybody who queued
// with its k by ONE batched search (at most kBatch queries), then hands the device on.  A batch's lists are the single searches'
// lists bit for bit, so nobody can tell except by the clock.  No allocation after the enqueue, nothing thrown.
struct HostReq {
    const float *q;
    int64_t nq;
    float *dist;
    int64_t *labels;
    int k;
    int rc = 0;
    bool done = false; // served by another caller's batch
    bool lead = false; // promoted: run the next batch (its own request included)
    std::condition_variable cv; // this caller's own wake-up (no herd: a finishing search wakes its batch and one successor)
    HostReq(const float *q_, int64_t nq_, float *dist_, int64_t *labels_, int k_) : q(q_), nq(nq_), dist(dist_), labels(labels_), k(k_) {}
};

class SearchCombiner {
  public:
    // requests of at most kMaxNq queries take part; a combined batch holds at most kBatch queries
    static constexpr int64_t kMaxNq = 16, kBatch = 256;
    // Two searches at a time while nothing is queued: one's host work (staging, copies, wake-up: ~80 us around a 0.3 ms search)
    // runs under the other's device work, and two callers alone are served as without combining.
    static constexpr int kLanes = 2;
    std::atomic<int> on{1};
    std::atomic<int64_t> batches{0}, requests{0}; // combined batches run / requests served by them

    // run(reqs, n, k) -> rc searches n requests with the same k as one device batch and fills every request's buffers
    template <typename Run>
    int search(HostReq &me, Run &&run)
    {
        HostReq *batch[kBatch]; // (every request holds at least one query)
        int nb = 0;
        {
            std::unique_lock<std::mutex> lk(mu_);
            // (two lanes while callers come one or two at a time; ONE under load: two batches side by side would halve each
            // other's share of the HBM stream, one batch of everybody does not)
            const int lanes = (recent_ > 1 || !wait_.empty()) ? 1 : kLanes;
            if (!gatherer_ && active_ < lanes) {
                active_++;
            } else {
                try {
                    wait_.push_back(&me);
                } catch (...) { // (out of memory: search alone, beside whoever holds the device)
                    lk.unlock();
                    HostReq *one = &me;
                    return run(&one, 1, me.k);
                }
                if (gatherer_) gatherer_->cv.notify_one(); // (the gathering caller counts the queue)
                me.cv.wait(lk, [&] { return me.done || me.lead; });
                if (me.done) return me.rc;
                for (auto it = wait_.begin(); it != wait_.end(); ++it) // (promoted: the lane is mine; leave the queue)
                    if (*it == &me) { wait_.erase(it); break; }
            }
            // Under load the callers of a combined batch come back together: the first of them would search alone and the
            // rest would wait out its whole search (8 callers: batches of 1 and 7 in turn, half the possible rate).  After a
            // combined batch the caller that gets the lane therefore gives the others a moment -- until as many have queued as
            // the last batch held, at most kGatherUs -- and takes them along.  A lone caller (the last batch was its own)
            // never waits.
            if (recent_ > 1 && (int)wait_.size() < recent_ - 1) {
                gatherer_ = &me;
                const int want = recent_ - 1;
                me.cv.wait_for(lk, std::chrono::microseconds(kGatherUs), [&] { return (int)wait_.size() >= want; });
                if (gatherer_ == &me) gatherer_ = nullptr;
            }
            nb = take_same_k(me, batch);
        }
        batch[nb++] = &me;
        const int rc = run(batch, nb, me.k);
        if (nb > 1) {
            batches.fetch_add(1);
            requests.fetch_add((int64_t)nb);
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            recent_ = nb;
            for (int i = 0; i + 1 < nb; i++) {
                batch[i]->rc = rc;
                batch[i]->done = true;
                batch[i]->cv.notify_one(); // (the waiter re-checks under the lock we hold: it cannot be gone before we let go)
            }
            // hand the device to the next waiting caller, or mark the lane free
            HostReq *next = nullptr;
            for (HostReq *r : wait_)
                if (!r->done && !r->lead) { next = r; break; }
            if (next && active_ > 1) next = nullptr; // (callers are queueing: one lane -- the other search's caller serves them)
            if (next) {
                next->lead = true; // (stays queued until it wakes; the lane is its)
                next->cv.notify_one();
            } else {
                active_--;
            }
        }
        return rc;
    }

  private:
    static constexpr int kGatherUs = 50;
    // every queued request with me's k (at most kBatch queries with me's) leaves the queue for me's batch
    int take_same_k(HostReq &me, HostReq **batch)
    {
        int nb = 0;
        int64_t total = me.nq;
        for (auto it = wait_.begin(); it != wait_.end();) {
            HostReq *r = *it;
            if (!r->lead && r->k == me.k && total + r->nq <= kBatch) {
                batch[nb++] = r;
                total += r->nq;
                it = wait_.erase(it);
            } else {
                ++it;
            }
        }
        return nb;
    }
    std::mutex mu_;
    int active_ = 0;              // searches of this kind on the device right now (at most kLanes)
    int recent_ = 1;              // requests in the batch that ended last
    HostReq *gatherer_ = nullptr; // the caller that holds the lane and is gathering the others of the last batch
    std::deque<HostReq *> wait_;
};

inline bool device_ok(int device)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return false;
    return device >= 0 && device < cnt;
}

} // namespace lb
