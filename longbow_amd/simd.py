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


def _batch(metric, query, vectors, results, order, device):
    vectors = np.ascontiguousarray(vectors, np.float32)
    if vectors.size == 0:
        return
    n, dims = vectors.shape
    if results.shape[0] < n:
        raise ValueError("simd: results slice too small")  # batch_operations.go:135-137
    _batch_flat(metric, query, vectors, n, dims, results[:n], order, device)


def EuclideanDistanceBatch(query, vectors, results, order=Order.Seq, device=0):
    vectors = np.ascontiguousarray(vectors, np.float32)
    if vectors.shape[0] != results.shape[0]:
        raise ValueError("simd: vectors and results length mismatch")  # batch_operations.go:30-32
    _batch(MetricType.Euclidean, query, vectors, results, order, device)


def CosineDistanceBatch(query, vectors, results, order=Order.Seq, device=0):
    _batch(MetricType.Cosine, query, vectors, results, order, device)


def DotProductBatch(query, vectors, results, order=Order.Seq, device=0):
    """raw dot products (not negated), as simd.DotProductBatch"""
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
