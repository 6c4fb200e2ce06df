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

    // SearchBatch under a cancellation context (the ctx of SearchVectors, internal/store/adaptive_index.go:182):
    // LB_ERR_CANCELLED / LB_ERR_DEADLINE when `ctx` fired or its deadline passed before the search finished
    Error SearchBatch(const float *queries, int64_t nq, int k, int64_t *ids, float *distances, const lb_cancel *ctx)
    {
        std::shared_lock<std::shared_mutex> g(mu_);
        if (closed_) return {LB_ERR_CLOSED, "index is closed"};
        return wrap("search", lb_gpu_index_search_ctx(h_, nq, queries, k, distances, ids, ctx));
    }

    // lb_candidate_mode: LB_CAND_AUTO (default), LB_CAND_F32_MFMA (strict), LB_CAND_SPLIT_BF16[_INREG]; same results
    Error SetCandidateMode(int mode)
    {
        std::unique_lock<std::shared_mutex> g(mu_);
        if (closed_) return {LB_ERR_CLOSED, "index is closed"};
        return wrap("set_candidate_mode", lb_gpu_index_set_candidate_mode(h_, mode));
    }

    // the fp16 route's own copy of the corpus (default on while memory allows); results do not depend on it
    Error SetF16Image(bool on)
    {
        std::unique_lock<std::shared_mutex> g(mu_);
        if (closed_) return {LB_ERR_CLOSED, "index is closed"};
        return wrap("set_f16_image", lb_gpu_index_set_f16_image(h_, on ? 1 : 0));
    }

    // concurrent single-query Search calls answered by one batched search (default on; lists identical either way)
    Error SetSearchCombining(bool on)
    {
        std::unique_lock<std::shared_mutex> g(mu_);
        if (closed_) return {LB_ERR_CLOSED, "index is closed"};
        return wrap("set_search_combining", lb_gpu_index_set_search_combining(h_, on ? 1 : 0));
    }

    // The distance step of processChunkInternal (internal/store/parallel_search.go:274-364) on rows that are
    // resident on the GPU: dist[i] as simd.EuclideanDistanceBatchFlat computes it (4-accumulator order),
    // score[i] = 1/(1+dist[i]).  rows are row positions.
    Error Rerank(const std::vector<float> &query, const std::vector<int64_t> &rows, std::vector<float> &dist,
                 std::vector<float> &score)
    {
        std::shared_lock<std::shared_mutex> g(mu_);
        if (closed_) return {LB_ERR_CLOSED, "index is closed"};
        if ((int)query.size() != dim_)
            return {LB_ERR_INVALID_ARG, "query vector dimension " + std::to_string(query.size()) +
                                            " does not match index dimension " + std::to_string(dim_)};
        dist.assign(rows.size(), 0.f);
        score.assign(rows.size(), 0.f);
        if (rows.empty()) return {};
        return wrap("rerank", lb_gpu_index_rerank(h_, query.data(), rows.data(), (int64_t)rows.size(), LB_ORDER_UNROLL4,
                                                  dist.data(), score.data()));
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

// ---- internal/simd: MetricType, KernelRegistry, dispatch (internal/simd/registry.go:8-124, dispatch.go:264-302) ----
namespace simd {

enum MetricType { MetricEuclidean = 0, MetricCosine = 1, MetricDotProduct = 2 };
inline const char *String(MetricType m) // MetricType.String(), registry.go:17-28
{
    switch (m) {
    case MetricEuclidean: return "euclidean";
    case MetricCosine: return "cosine";
    case MetricDotProduct: return "dot";
    default: return "unknown";
    }
}
enum SIMDDataType { DataTypeFloat32 = 0, DataTypeFloat16 = 1 /* ... registry.go:32-47 */ };

// core.DistanceMetric strings (internal/core/enums.go:6-13) and MetricType.String()'s "dot"
inline bool MetricFromCore(const std::string &name, MetricType &out)
{
    if (name == "euclidean" || name.empty()) { out = MetricEuclidean; return true; }
    if (name == "cosine") { out = MetricCosine; return true; }
    if (name == "dot_product" || name == "dot") { out = MetricDotProduct; return true; }
    return false;
}

// one query x n rows of a flat buffer -> results (simd.EuclideanDistanceBatchFlat's signature, batch_operations.go:64-87)
using BatchFlatFunc = int (*)(const float *query, const float *flat, int64_t n, int dims, float *results);
constexpr int BatchFlatDims = -1; // KernelKey.Dims of batch kernels (go/internal/simd/hip_kernels.go)

struct KernelKey {
    int Metric, DataType, Dims;
    bool operator<(const KernelKey &o) const
    {
        if (Metric != o.Metric) return Metric < o.Metric;
        if (DataType != o.DataType) return DataType < o.DataType;
        return Dims < o.Dims;
    }
};

// KernelRegistry.Register / Get: exact (metric, type, dims) first, then the generic dims = 0 entry, else null
class KernelRegistry {
  public:
    void Register(MetricType m, SIMDDataType dt, int dims, BatchFlatFunc k)
    {
        std::unique_lock<std::shared_mutex> g(mu_);
        for (auto &e : kernels_)
            if (!(e.first < KernelKey{m, dt, dims}) && !(KernelKey{m, dt, dims} < e.first)) { e.second = k; return; }
        kernels_.emplace_back(KernelKey{m, dt, dims}, k);
    }
    BatchFlatFunc Get(MetricType m, SIMDDataType dt, int dims) const
    {
        std::shared_lock<std::shared_mutex> g(mu_);
        BatchFlatFunc generic = nullptr;
        for (const auto &e : kernels_) {
            if (e.first.Metric != m || e.first.DataType != dt) continue;
            if (e.first.Dims == dims) return e.second;
            if (e.first.Dims == 0) generic = e.second;
        }
        return generic;
    }

  private:
    mutable std::shared_mutex mu_;
    std::vector<std::pair<KernelKey, BatchFlatFunc>> kernels_;
};

inline int HIPDevice = 0;
template <int METRIC>
inline int hipBatchFlat(const float *q, const float *flat, int64_t n, int dims, float *results)
{
    return lb_simd_distance_batch_flat(HIPDevice, METRIC, LB_ORDER_UNROLL4, q, flat, n, dims, results);
}
// the registry with the HIP batch kernels plugged in, as go/internal/simd/hip_kernels.go's init() does
inline KernelRegistry &Registry()
{
    static KernelRegistry *r = [] {
        auto *x = new KernelRegistry();
        x->Register(MetricEuclidean, DataTypeFloat32, BatchFlatDims, hipBatchFlat<LB_METRIC_EUCLIDEAN>);
        x->Register(MetricCosine, DataTypeFloat32, BatchFlatDims, hipBatchFlat<LB_METRIC_COSINE>);
        x->Register(MetricDotProduct, DataTypeFloat32, BatchFlatDims, hipBatchFlat<LB_METRIC_DOT>);
        return x;
    }();
    return *r;
}
// DispatchDistance's rule for one query against n rows: 0 = ok, else an lb_status
inline int DispatchBatchFlat(MetricType m, const float *q, const float *flat, int64_t n, int dims, float *results)
{
    BatchFlatFunc k = Registry().Get(m, DataTypeFloat32, BatchFlatDims);
    if (!k) return LB_ERR_UNSUPPORTED; // "simd: no kernel found"
    return k(q, flat, n, dims, results);
}

} // namespace simd

// ---- internal/pq: PQEncoder's query side + codec (internal/pq/encoder.go:76-158, adc_table.go:15-72) ----
namespace pq {

class PQEncoder {
  public:
    PQEncoder(const PQEncoder &) = delete;
    PQEncoder &operator=(const PQEncoder &) = delete;
    ~PQEncoder() { if (p_) lb_gpu_pq_free(p_); }
    // DeserializePQEncoder (persistence.go:38-73) onto the GPU; null + status on error
    static PQEncoder *Deserialize(const std::vector<uint8_t> &blob, int device, int &status)
    {
        lb_gpu_pq *p = lb_gpu_pq_new(device, blob.data(), blob.size(), &status);
        if (!p) return nullptr;
        auto *e = new PQEncoder();
        e->p_ = p;
        e->M = lb_gpu_pq_m(p);
        e->Dims = lb_gpu_pq_dims(p);
        return e;
    }
    int Encode(const std::vector<float> &vectors, std::vector<uint8_t> &codes) // n x Dims -> n x M
    {
        if (vectors.size() % (size_t)Dims != 0) return LB_ERR_INVALID_ARG; // "vector dimension mismatch"
        const int64_t n = (int64_t)(vectors.size() / (size_t)Dims);
        codes.assign((size_t)n * M, 0);
        return lb_gpu_pq_encode(p_, n, vectors.data(), codes.data());
    }
    int Decode(const std::vector<uint8_t> &codes, std::vector<float> &vectors)
    {
        if (codes.size() % (size_t)M != 0) return LB_ERR_INVALID_ARG; // "code length mismatch"
        const int64_t n = (int64_t)(codes.size() / (size_t)M);
        vectors.assign((size_t)n * Dims, 0.f);
        return lb_gpu_pq_decode(p_, n, codes.data(), vectors.data());
    }
    int BuildADCTable(const std::vector<float> &query, std::vector<float> &table)
    {
        if ((int)query.size() != Dims) return LB_ERR_INVALID_ARG; // "query dimension mismatch"
        table.assign((size_t)M * 256, 0.f);
        return lb_gpu_pq_build_adc_table(p_, query.data(), table.data());
    }
    int M = 0, Dims = 0;

  private:
    PQEncoder() = default;
    lb_gpu_pq *p_ = nullptr;
};

} // namespace pq
} // namespace longbow
