// lb_combine.h -- combining of concurrent host-pointer searches (lb_gpu_index_search, lb_gpu_pq_search).  Standard library
// only: tests/cpp/combiner_tsan.cpp builds it under ThreadSanitizer without HIP.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>

namespace lb {

// The reference's gpu.Index.Search is ONE query per call, from many goroutines (internal/gpu/faiss_gpu.go:108-145).  Served as
// they come, T overlapping calls stream the corpus T times.  Instead: one or two callers search at once (two lanes while nothing
// is queued, one under load); a caller that finds the lanes taken queues, and when a search ends the first in the queue answers
// everybody who queued with its k by ONE batched search (at most kBatch queries), then hands the lane on.  A batch's lists are the
// single searches' lists bit for bit, so nobody can tell except by the clock.  No allocation after the enqueue, nothing thrown:
// an exception out of run() is caught here and becomes kRunThrew (the lane is always handed on).  A combined batch that FAILS
// is not reported to its callers as one: the leader searches every request of it again alone, so that each caller gets the
// status of its own query (a device error in one caller's batch does not fail up to 255 others).  The handle's last_error /
// last_fallbacks / last_route describe the last device BATCH, which under combining may hold other callers' queries.
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
    static constexpr int kRunThrew = 7; // (== LB_ERR_INTERNAL: run() threw)
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
            // (the size of the last batch is evidence of load only for a moment: after kRecentUs without a search a lone
            // caller neither loses the second lane nor pays the gather wait)
            if (recent_ > 1 && std::chrono::steady_clock::now() - last_end_ > std::chrono::microseconds(kRecentUs)) recent_ = 1;
            const int lanes = (recent_ > 1 || !wait_.empty()) ? 1 : kLanes;
            if (!gatherer_ && active_ < lanes) {
                active_++;
            } else {
                try {
                    wait_.push_back(&me);
                } catch (...) { // (out of memory: search alone, beside whoever holds the device)
                    lk.unlock();
                    HostReq *one = &me;
                    return guarded(run, &one, 1, me.k);
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
#ifdef __SANITIZE_THREAD__ // (gcc 11's libtsan does not intercept pthread_cond_clockwait, which a steady-clock wait uses)
                me.cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(kGatherUs),
                                 [&] { return (int)wait_.size() >= want; });
#else
                me.cv.wait_for(lk, std::chrono::microseconds(kGatherUs), [&] { return (int)wait_.size() >= want; });
#endif
                if (gatherer_ == &me) gatherer_ = nullptr;
            }
            nb = take_same_k(me, batch);
        }
        batch[nb++] = &me;
        int rc = guarded(run, batch, nb, me.k);
        if (nb > 1) {
            batches.fetch_add(1);
            requests.fetch_add((int64_t)nb);
            if (rc != 0) { // every request again, alone: each caller gets its own query's status
                retried.fetch_add(1);
                for (int i = 0; i < nb; i++) {
                    HostReq *one = batch[i];
                    one->rc = guarded(run, &one, 1, one->k);
                }
                rc = me.rc;
            } else {
                for (int i = 0; i < nb; i++) batch[i]->rc = 0;
            }
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            recent_ = nb;
            last_end_ = std::chrono::steady_clock::now();
            for (int i = 0; i + 1 < nb; i++) {
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

    std::atomic<int64_t> retried{0}; // combined batches that failed and were searched again request by request

  private:
    static constexpr int kGatherUs = 50, kRecentUs = 2000;
    template <typename Run>
    static int guarded(Run &run, HostReq *const *reqs, int n, int k)
    {
        try {
            return run(reqs, n, k);
        } catch (...) {
            return kRunThrew;
        }
    }
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
    std::chrono::steady_clock::time_point last_end_{}; // when it ended
    HostReq *gatherer_ = nullptr; // the caller that holds the lane and is gathering the others of the last batch
    std::deque<HostReq *> wait_;
};

} // namespace lb
