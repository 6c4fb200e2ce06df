// T native threads of single-query host-pointer searches on one index (what goroutines calling gpu.Index.Search do), with and
// without request combining.  Build and run on the GPU box:
//   hipcc -O2 -std=c++17 tools/probe/concurrent_single.cpp -Iinclude -Llongbow_amd -llongbow_gpu -Wl,-rpath,$PWD/longbow_amd -lpthread -o /tmp/cs && /tmp/cs
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

#include "longbow_gpu.h"

int main()
{
    const int64_t rows = 1000000;
    const int D = 768, K = 100;
    int st = 0;
    lb_gpu_index *h = lb_gpu_index_new(0, D, 1, &st);
    if (!h) { std::printf("index_new failed: %d\n", st); return 1; }
    float *dX = nullptr;
    if (hipMalloc(&dX, (size_t)rows * D * 4) != hipSuccess) return 1;
    lb_gpu_fill_uniform_device(0, dX, rows * D, 12345, 0, nullptr);
    lb_gpu_index_reserve(h, rows);
    if (lb_gpu_index_add_device(h, rows, dX, nullptr) != LB_OK) { std::printf("add failed\n"); return 1; }
    (void)hipFree(dX);
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> uni(0.f, 1.f);
    std::vector<std::vector<float>> Q(64, std::vector<float>(D));
    for (auto &q : Q)
        for (auto &v : q) v = uni(rng);
    for (int comb = 0; comb < 2; comb++) {
        lb_gpu_index_set_search_combining(h, comb);
        for (int T : {1, 2, 4, 8, 16, 32, 64}) {
            std::vector<std::vector<double>> lat(T);
            std::atomic<int> ready{0};
            std::atomic<bool> go{false};
            std::vector<std::thread> ths;
            for (int t = 0; t < T; t++)
                ths.emplace_back([&, t] {
                    std::vector<float> d(K);
                    std::vector<int64_t> l(K);
                    for (int i = 0; i < 3; i++) lb_gpu_index_search(h, 1, Q[t % 64].data(), K, d.data(), l.data());
                    ready++;
                    while (!go.load()) std::this_thread::yield();
                    const auto t0 = std::chrono::steady_clock::now();
                    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.5) {
                        const auto a = std::chrono::steady_clock::now();
                        lb_gpu_index_search(h, 1, Q[t % 64].data(), K, d.data(), l.data());
                        lat[t].push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count());
                    }
                });
            while (ready.load() < T) std::this_thread::yield();
            go = true;
            for (auto &th : ths) th.join();
            std::vector<double> all;
            for (auto &v : lat) all.insert(all.end(), v.begin(), v.end());
            std::sort(all.begin(), all.end());
            int64_t stats[2] = {0, 0};
            lb_gpu_index_combining_stats(h, stats);
            std::printf("combining=%d threads=%3d: %9.0f queries/s  p50 %.3f ms  p99 %.3f ms  (batches %lld, requests %lld)\n", comb, T,
                        all.size() / 1.5, all[all.size() / 2] * 1e3, all[(size_t)(all.size() * 0.99)] * 1e3, (long long)stats[0],
                        (long long)stats[1]);
            std::fflush(stdout);
        }
    }
    lb_gpu_index_free(h);
    return 0;
}
