"""Mirror of internal/simd's metric interface, computed by HIP kernels.

MetricType / names: internal/simd/registry.go:8-29.  Batch entry points:
internal/simd/batch_operations.go:29-157.  Argument meaning and error behaviour
follow the Go functions (ValueError where Go returns an error).
"""
import ctypes as C
import enum

import numpy as np

from . import _lib


class MetricType(enum.IntEnum):
    Euclidean = 0   # MetricEuclidean
    Cosine = 1      # MetricCosine
    DotProduct = 2  # MetricDotProduct

    def __str__(self):  # MetricType.String(), registry.go:17-28
        return {0: "euclidean", 1: "cosine", 2: "dot"}[int(self)]


class Order(enum.IntEnum):
    Seq = 0      # simd_test.go:13-33 / simd.go:138-163
    Unroll4 = 1  # simd.go:365-479


def _batch_flat(metric, query, flat, num_vectors, dims, results, order, device):
    lib = _lib.require_gpu(device)
    query = np.ascontiguousarray(query, np.float32)
    flat = np.ascontiguousarray(flat, np.float32).reshape(-1)
    if num_vectors == 0:
        return  # batch_operations.go:65-67
    if results.shape[0] != num_vectors:
        raise ValueError("simd: results length mismatch")       # simd.go:204-206
    if flat.size < num_vectors * dims:
        raise ValueError("simd: flatVectors too small")          # simd.go:207-209
    if query.size != dims:
        raise ValueError("simd: query dimension mismatch")       # simd.go:215-217
    if results.dtype != np.float32 or not results.flags.c_contiguous:
        raise ValueError("simd: results must be contiguous float32")
    rc = lib.lb_simd_distance_batch_flat(device, int(metric), int(order), query.ctypes.data,
                                         flat.ctypes.data, num_vectors, dims, results.ctypes.data)
    _lib.check(rc)


def EuclideanDistanceBatchFlat(query, flatVectors, numVectors, dims, results, order=Order.Unroll4, device=0):
    """simd.EuclideanDistanceBatchFlat (batch_operations.go:64-87).  The reference runs the
    4-accumulator order here (simd.go:221), hence the default."""
    _batch_flat(MetricType.Euclidean, query, flatVectors, numVectors, dims, results, order, device)


def _is_ragged(vectors):
    """a Go [][]float32 with nil or odd-length members, as opposed to a rectangular 2-D array"""
    if isinstance(vectors, np.ndarray) and vectors.dtype != object:
        return False
    return any(v is None for v in vectors) or len({len(v) for v in vectors if v is not None}) > 1


def _batch_slices(metric, query, vectors, results, order, device):
    """[][]float32 form through lb_simd_distance_batch: nil / length-mismatched members follow the reference's
    per-vector rules (batch_operations.go:39-42,51; simd.go:241-267)."""
    lib = _lib.require_gpu(device)
    query = np.ascontiguousarray(query, np.float32).reshape(-1)
    n = len(vectors)
    keep = [None if v is None else np.ascontiguousarray(v, np.float32).reshape(-1) for v in vectors]
    ptrs = (C.c_void_p * n)(*[None if v is None else v.ctypes.data for v in keep])
    lens = (C.c_int * n)(*[0 if v is None else v.size for v in keep])
    if results.dtype != np.float32 or not results.flags.c_contiguous:
        raise ValueError("simd: results must be contiguous float32")
    _lib.check(lib.lb_simd_distance_batch(device, int(metric), int(order), query.ctypes.data, query.size, ptrs, lens, n,
                                          results.ctypes.data))


def _batch(metric, query, vectors, results, order, device):
    if len(vectors) == 0:
        return
    if results.shape[0] < len(vectors):
        raise ValueError("simd: results slice too small")  # batch_operations.go:135-137
    if _is_ragged(vectors) or np.asarray(query).size != np.asarray(vectors[0]).size:
        return _batch_slices(metric, query, vectors, results, order, device)
    vectors = np.ascontiguousarray(vectors, np.float32)
    n, dims = vectors.shape
    _batch_flat(metric, query, vectors, n, dims, results[:n], order, device)


def EuclideanDistanceBatch(query, vectors, results, order=Order.Seq, device=0):
    """simd.EuclideanDistanceBatch (batch_operations.go:29-60): `vectors` is a 2-D array or a list of vectors; a None
    or length-mismatched member yields math.MaxFloat32 in its slot."""
    if len(vectors) != results.shape[0]:
        raise ValueError("simd: vectors and results length mismatch")  # batch_operations.go:30-32
    _batch(MetricType.Euclidean, query, vectors, results, order, device)


def CosineDistanceBatch(query, vectors, results, order=Order.Seq, device=0):
    """simd.CosineDistanceBatch (batch_operations.go:131-143): None members are skipped (their slot keeps its value);
    a length mismatch stops the batch at that member, error swallowed, as the reference does."""
    _batch(MetricType.Cosine, query, vectors, results, order, device)


def DotProductBatch(query, vectors, results, order=Order.Seq, device=0):
    """raw dot products (not negated), as simd.DotProductBatch; member rules as CosineDistanceBatch"""
    _batch(MetricType.DotProduct, query, vectors, results, order, device)


def EuclideanDistance(a, b, order=Order.Seq, device=0):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.size != b.size:
        raise ValueError("simd: vector length mismatch")  # distance_functions.go:18-20
    if a.size == 0:
        return np.float32(0)
    r = np.empty(1, np.float32)
    _batch_flat(MetricType.Euclidean, a, b, 1, a.size, r, order, device)
    return r[0]


def CosineDistance(a, b, order=Order.Seq, device=0):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.size != b.size:
        raise ValueError("simd: vector length mismatch")
    if a.size == 0:
        return np.float32(1.0)  # distance_functions.go:51-53
    r = np.empty(1, np.float32)
    _batch_flat(MetricType.Cosine, a, b, 1, a.size, r, order, device)
    return r[0]


def DotProduct(a, b, order=Order.Seq, device=0):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.size != b.size:
        raise ValueError("simd: vector length mismatch")
    if a.size == 0:
        return np.float32(0)
    r = np.empty(1, np.float32)
    _batch_flat(MetricType.DotProduct, a, b, 1, a.size, r, order, device)
    return r[0]


class CompareOp(enum.IntEnum):
    """simd.CompareOp (internal/simd/simd.go:38-45)"""
    Eq = 0
    Neq = 1
    Gt = 2
    Ge = 3
    Lt = 4
    Le = 5


_OPERATORS = {"=": CompareOp.Eq, "eq": CompareOp.Eq, "==": CompareOp.Eq, "!=": CompareOp.Neq, "neq": CompareOp.Neq,
              ">": CompareOp.Gt, "gt": CompareOp.Gt, ">=": CompareOp.Ge, "ge": CompareOp.Ge,
              "<": CompareOp.Lt, "lt": CompareOp.Lt, "<=": CompareOp.Le, "le": CompareOp.Le}


def parse_operator(op):
    """operator strings of query.Filter (internal/query/filter_evaluator.go:80-93)"""
    if isinstance(op, str):
        if op not in _OPERATORS:
            raise ValueError(f"unsupported operator {op!r}")
        return _OPERATORS[op]
    return CompareOp(int(op))


def MatchInt64(src, val, op, dst, device=0):
    """simd.MatchInt64: dst[i] = 1 if src[i] OP val else 0"""
    lib = _lib.require_gpu(device)
    src = np.ascontiguousarray(src, np.int64)
    if src.size != dst.size:
        raise ValueError("simd: length mismatch")  # simd.go:573-575
    _lib.check(lib.lb_simd_match_int64(device, src.ctypes.data, src.size, int(val), int(parse_operator(op)),
                                       dst.ctypes.data))


def MatchFloat32(src, val, op, dst, device=0):
    lib = _lib.require_gpu(device)
    src = np.ascontiguousarray(src, np.float32)
    if src.size != dst.size:
        raise ValueError("simd: length mismatch")
    _lib.check(lib.lb_simd_match_float32(device, src.ctypes.data, src.size, float(val), int(parse_operator(op)),
                                         dst.ctypes.data))


def AndBytes(dst, src, device=0):
    """simd.AndBytes: dst[i] &= src[i]"""
    lib = _lib.require_gpu(device)
    src = np.ascontiguousarray(src, np.uint8)
    if src.size != dst.size:
        raise ValueError("simd: length mismatch")  # simd.go:120-122
    _lib.check(lib.lb_simd_and_bytes(device, dst.ctypes.data, src.ctypes.data, dst.size))


# ---------------------------------------------------------------------------------------------------
# internal/simd's dispatch surface (a13): SIMDDataType, KernelKey, KernelRegistry.Register/Get,
# DispatchDistance, core.DistanceMetric strings -- with the HIP kernels registered the way
# go/internal/simd/hip_kernels.go registers them in the Go tree.
# ---------------------------------------------------------------------------------------------------
class SIMDDataType(enum.IntEnum):
    """simd.SIMDDataType (internal/simd/registry.go:32-47)"""
    Float32 = 0
    Float16 = 1
    Int8 = 2
    Uint8 = 3
    Int16 = 4
    Uint16 = 5
    Int32 = 6
    Uint32 = 7
    Int64 = 8
    Uint64 = 9
    Float64 = 10
    Complex64 = 11
    Complex128 = 12

    def __str__(self):  # SIMDDataType.String(), registry.go:49-79
        return self.name.lower()


BatchFlatDims = -1  # KernelKey.Dims of the batch-flat kernels (no vector has a negative length)


class KernelRegistry:
    """simd.KernelRegistry (internal/simd/registry.go:82-124): kernels keyed by (metric, data type, dims);
    Get tries the exact dims first and then the generic dims = 0 entry."""

    def __init__(self):
        import threading
        self._mu = threading.RLock()
        self._kernels = {}

    def Register(self, metric, dt, dims, kernel):
        with self._mu:
            self._kernels[(int(metric), int(dt), int(dims))] = kernel

    def Get(self, metric, dt, dims):
        with self._mu:
            k = self._kernels.get((int(metric), int(dt), int(dims)))
            if k is not None:
                return k
            return self._kernels.get((int(metric), int(dt), 0))


Registry = KernelRegistry()
# per-pair kernels at the generic key, as dispatch.go:221-234 registers the CPU ones
Registry.Register(MetricType.Euclidean, SIMDDataType.Float32, 0, EuclideanDistance)
Registry.Register(MetricType.Cosine, SIMDDataType.Float32, 0, CosineDistance)
Registry.Register(MetricType.DotProduct, SIMDDataType.Float32, 0, DotProduct)


def _hip_batch_flat(metric):
    def kernel(query, flatVectors, numVectors, dims, results, order=Order.Unroll4, device=0):
        _batch_flat(metric, query, flatVectors, numVectors, dims, results, order, device)
    kernel.metric = metric
    return kernel


for _m in MetricType:  # go/internal/simd/hip_kernels.go: init()
    Registry.Register(_m, SIMDDataType.Float32, BatchFlatDims, _hip_batch_flat(_m))


def DispatchDistance(metric, a, b):
    """simd.DispatchDistance (internal/simd/dispatch.go:264-302)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.size != b.size:
        raise ValueError(f"simd: dimension mismatch: {a.size} != {b.size}")
    if a.size == 0:
        return np.float32(0)
    kernel = Registry.Get(metric, SIMDDataType.Float32, a.size)
    if kernel is None:
        raise ValueError(f"simd: no kernel found for {MetricType(metric)}/{SIMDDataType.Float32} dims={a.size}")
    return kernel(a, b)


def DispatchBatchFlat(metric, query, flatVectors, numVectors, dims, results, **kw):
    """one query x n rows through the registry: the HIP batch kernel when registered (exact key), else the
    generic per-pair kernel row by row (go/internal/simd/hip_kernels.go: DispatchBatchFlat)"""
    kernel = Registry.Get(metric, SIMDDataType.Float32, BatchFlatDims)
    if kernel is None:
        raise ValueError(f"simd: no kernel found for {MetricType(metric)}/{SIMDDataType.Float32}")
    if getattr(kernel, "metric", None) is not None:
        return kernel(query, flatVectors, numVectors, dims, results, **kw)
    flat = np.ascontiguousarray(flatVectors, np.float32).reshape(-1)
    for i in range(numVectors):
        results[i] = kernel(query, flat[i * dims:(i + 1) * dims])


_CORE_METRICS = {"euclidean": MetricType.Euclidean, "": MetricType.Euclidean, "cosine": MetricType.Cosine,
                 "dot_product": MetricType.DotProduct, "dot": MetricType.DotProduct}


def MetricFromCore(name):
    """core.DistanceMetric strings (internal/core/enums.go:6-13) and MetricType.String()'s "dot" -> MetricType"""
    if isinstance(name, MetricType):
        return name
    try:
        return _CORE_METRICS[str(name)]
    except KeyError:
        raise ValueError(f"simd: unknown distance metric {name!r}") from None
