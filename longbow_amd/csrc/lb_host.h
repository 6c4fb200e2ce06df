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
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>

#include "lb_combine.h"

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

inline bool device_ok(int device)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return false;
    return device >= 0 && device < cnt;
}

} // namespace lb
