// longbow_gpu.hpp -- header-only C++ host mirror of Longbow's gpu.Index over the C ABI.
//
// The reference's host language (Go) is not available in this build image; this is the
// compiled-language host side above the C ABI, with the reference's names and argument meaning:
//   gpu.GPUConfig / gpu.Index{Add, Search, Close} / NewIndex / NewIndexWithConfig
//   (internal/gpu/interface.go:3-19, gpu_enabled.go:8-21, faiss_gpu.go:44-167).
// Errors are reported the way the Go code reports them (error value != nil): methods return an
// Error whose ok() is false and whose message is the reference's text where it has one.
#pragma once
#include <cstdint>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <utility>
#include <vector>

#include "longbow_gpu.h"

namespace longbow {
namespace gpu {

struct Error {
    int code = LB_OK;
    std::string message;
    bool ok() const { return code == LB_OK; }
    explicit operator bool() const { return !ok(); } // `if (err)` reads like Go's `if err != nil`
};

inline Error ErrGPUNotAvailable() { return {LB_ERR_NO_DEVICE, "GPU support not enabled in this build"}; }

struct GPUConfig {
    int DeviceID = 0;
    int Dimension = 128;
    int Metric = LB_METRIC_EUCLIDEAN; // simd.MetricType value
};

class Index {
  public:
    Index() = default;
    Index(const Index &) = delete;
    Index &operator=(const Index &) = delete;
    ~Index() { Close(); }

    // Add(ids []int64, vectors []float32) error
    Error Add(const std::vector<int64_t> &ids, const std::vector<float> &vectors)
    {
        std::unique_lock<std::shared_mutex> g(mu_);
        if (closed_) return {LB_ERR_CLOSED, "index is closed"};
        if (vectors.size() % (size_t)dim_ != 0)
            return {LB_ERR_INVALID_ARG, "vector data length " + std::to_string(vectors.size()) +
                                            " not divisible by dimension " + std::to_string(dim_)};
        const size_t n = vectors.size() / (size_t)dim_;
        if (ids.size() != n)
            return {LB_ERR_INVALID_ARG, "id count " + std::to_string(ids.size()) +
                                            " does not match vector count " + std::to_string(n)};
        if (n == 0) return {};
        return wrap("add", lb_gpu_index_add(h_, (int64_t)n, vectors.data(), ids.data()));
    }

    // Search(vector []float32, k int) (ids []int64, distances []float32, err error)
    Error Search(const std::vector<float> &vector, int k, std::vector<int64_t> &ids, std::vector<float> &distances)
    {
        std::shared_lock<std::shared_mutex> g(mu_);
        if (closed_) return {LB_ERR_CLOSED, "index is closed"};
        if ((int)vector.size() != dim_)
            return {LB_ERR_INVALID_ARG, "query vector dimension " + std::to_string(vector.size()) +
                                            " does not match index dimension " + std::to_string(dim_)};
        ids.assign((size_t)k, -1);
        distances.assign((size_t)k, 0.f);
        Error e = wrap("search", lb_gpu_index_search(h_, 1, vector.data(), k, distances.data(), ids.data()));
        if (e) return e;
        size_t n = (size_t)k;
        while (n > 0 && ids[n - 1] < 0) n--; // trim FAISS-style padding: min(k, N) results
        ids.resize(n);
        distances.resize(n);
        return {};
    }

    // batched superset: nq row-major queries -> nq*k results (padding kept)
    Error SearchBatch(const float *queries, int64_t nq, int k, int64_t *ids, float *distances)
    {
        std::shared_lock<std::shared_mutex> g(mu_);
        if (closed_) return {LB_ERR_CLOSED, "index is closed"};
        return wrap("search", lb_gpu_index_search(h_, nq, queries, k, distances, ids));
    }

    Error Close() // idempotent
    {
        std::unique_lock<std::shared_mutex> g(mu_);
        if (closed_) return {};
        if (h_) lb_gpu_index_free(h_);
        h_ = nullptr;
        closed_ = true;
        return {};
    }

    friend std::pair<Index *, Error> NewIndexWithConfig(const GPUConfig &cfg);

  private:
    Error wrap(const char *op, int rc)
    {
        if (rc == LB_OK) return {};
        if (rc == LB_ERR_NO_DEVICE) return ErrGPUNotAvailable();
        return {rc, std::string("GPU index ") + op + " failed with code " + std::to_string(rc) + " (" +
                        lb_gpu_status_string(rc) + ": " + lb_gpu_last_error(h_) + ")"};
    }
    lb_gpu_index *h_ = nullptr;
    int dim_ = 0;
    bool closed_ = true;
    std::shared_mutex mu_;
};

// NewIndexWithConfig(cfg GPUConfig) (Index, error)
inline std::pair<Index *, Error> NewIndexWithConfig(const GPUConfig &cfg)
{
    if (cfg.Dimension <= 0)
        return {nullptr, {LB_ERR_INVALID_ARG, "dimension must be positive, got " + std::to_string(cfg.Dimension)}};
    int st = LB_OK;
    lb_gpu_index *h = lb_gpu_index_new(cfg.DeviceID, cfg.Dimension, cfg.Metric, &st);
    if (!h) {
        if (st == LB_ERR_NO_DEVICE) return {nullptr, ErrGPUNotAvailable()};
        return {nullptr, {st, std::string("failed to create GPU index: ") + lb_gpu_status_string(st)}};
    }
    auto *idx = new Index();
    idx->h_ = h;
    idx->dim_ = cfg.Dimension;
    idx->closed_ = false;
    return {idx, {}};
}

// NewIndex(): device 0, dimension 128 (gpu_enabled.go:8-14)
inline std::pair<Index *, Error> NewIndex() { return NewIndexWithConfig(GPUConfig{}); }

} // namespace gpu
} // namespace longbow
