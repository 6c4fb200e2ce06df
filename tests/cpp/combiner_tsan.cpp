// SearchCombiner (longbow_amd/csrc/lb_combine.h) under ThreadSanitizer, host only: T threads of single-query "searches" whose
// device part is a stub (sleeps like a corpus pass, then writes results that are a function of each request's query alone).
// Checks: every caller gets ITS results and return code, requests with another k are never mixed into a batch, a failing batch
// is searched again request by request so that ONLY the poisoned request fails, a run() that throws becomes an error code and
// the lane is handed on, batches do get combined under load, a lone caller is not delayed, nothing deadlocks.
//   g++ -std=c++17 -O1 -g -fsanitize=thread tests/cpp/combiner_tsan.cpp -Ilongbow_amd/csrc -lpthread -o /tmp/combiner_tsan
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "lb_combine.h"

using lb::HostReq;
using lb::SearchCombiner;

static std::atomic<long long> g_runs{0}, g_mixed_k{0}, g_max_batch{0};

static int stub_run(HostReq *const *reqs, int n, int k)
{
    g_runs++;
    long long tot = 0;
    int rc = 0;
    for (int i = 0; i < n; i++) {
        if (reqs[i]->k != k) g_mixed_k++;
        tot += reqs[i]->nq;
        if (reqs[i]->q[0] < 0.f) rc = 4; // a poisoned request fails its whole batch (as a device error would)
        if (reqs[i]->q[0] < -1.5f) throw 1; // (run() must not throw; the combiner survives one that does)
    }
    long long m = g_max_batch.load();
    while (tot > m && !g_max_batch.compare_exchange_weak(m, tot)) {}
    std::this_thread::sleep_for(std::chrono::microseconds(150)); // "the pass"
    if (rc) return rc;
    for (int i = 0; i < n; i++)
        for (int64_t j = 0; j < reqs[i]->nq * k; j++) {
            reqs[i]->dist[j] = reqs[i]->q[0] * 1000.f + (float)j;
            reqs[i]->labels[j] = (int64_t)(reqs[i]->q[0] * 7.f) + j;
        }
    return 0;
}

int main()
{
    SearchCombiner cb;
    const int T = 12, REPS = 300;
    std::atomic<int> bad{0}, failed_ok{0}, poisoned{0};
    std::vector<std::thread> ths;
    for (int t = 0; t < T; t++)
        ths.emplace_back([&, t] {
            const int k = (t % 4 == 3) ? 5 : 10; // a quarter of the callers use another k
            std::vector<float> d(16 * 10);
            std::vector<int64_t> l(16 * 10);
            for (int r = 0; r < REPS; r++) {
                const int64_t nq = 1 + (r + t) % 3;
                const bool poison = (t == 5 && r % 50 == 7), thrower = (t == 6 && r % 100 == 9);
                float q = thrower ? -2.f : (poison ? -1.f : (float)(t * 1000 + r));
                HostReq me{&q, nq, d.data(), l.data(), k};
                const int rc = cb.search(me, stub_run);
                if (poison || thrower) {
                    poisoned++;
                    if (rc != (thrower ? SearchCombiner::kRunThrew : 4)) bad++;
                    else failed_ok++;
                    continue;
                }
                if (rc != 0) { // (a poisoned request in the same batch must not fail this one)
                    bad++;
                    continue;
                }
                for (int64_t j = 0; j < nq * k; j++)
                    if (d[j] != q * 1000.f + (float)j || l[j] != (int64_t)(q * 7.f) + j) { bad++; break; }
                if (r % 40 == 0) std::this_thread::sleep_for(std::chrono::microseconds(300 + 10 * t)); // fall out of step
            }
        });
    for (auto &th : ths) th.join();
    // a lone caller afterwards: its second call must not be delayed by a gather window's worth per call
    float q = 1.f;
    std::vector<float> d(10);
    std::vector<int64_t> l(10);
    HostReq warm{&q, 1, d.data(), l.data(), 10};
    (void)cb.search(warm, stub_run);
    const long long before = cb.batches.load();
    for (int i = 0; i < 20; i++) {
        HostReq me{&q, 1, d.data(), l.data(), 10};
        if (cb.search(me, stub_run) != 0) bad++;
    }
    const bool lone_uncombined = cb.batches.load() == before;
    std::printf("runs %lld, combined batches %lld holding %lld requests, largest batch %lld queries, mixed-k %lld, wrong results %d, "
                "failed calls %d (batches searched again %lld), lone caller uncombined %d\n",
                g_runs.load(), (long long)cb.batches.load(), (long long)cb.requests.load(), g_max_batch.load(), g_mixed_k.load(), bad.load(),
                failed_ok.load(), (long long)cb.retried.load(), (int)lone_uncombined);
    const bool ok = bad.load() == 0 && g_mixed_k.load() == 0 && cb.batches.load() > 0 && failed_ok.load() == poisoned.load() && poisoned.load() == 9 && lone_uncombined &&
                    g_max_batch.load() <= SearchCombiner::kBatch;
    std::printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
