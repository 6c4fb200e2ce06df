"""ctypes binding of oracle/liblongbow_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/longbow_oracle.h for what each function
restates (reference file:line).  Never imported by longbow_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblongbow_oracle.so")

EUCLIDEAN, COSINE, DOT = 0, 1, 2
SEQ, UNROLL4 = 0, 1

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("longbow_oracle.c", "cpu_baseline.c", "longbow_oracle.h")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        for name in ("lbo_l2sq", "lbo_euclidean", "lbo_cosine", "lbo_dot"):
            f = getattr(L, name)
            f.restype = C.c_float
            f.argtypes = [_f32p, _f32p, C.c_int, C.c_int]
        L.lbo_distance.restype = C.c_float
        L.lbo_distance.argtypes = [C.c_int, _f32p, _f32p, C.c_int, C.c_int]
        L.lbo_batch_flat.restype = None
        L.lbo_batch_flat.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.c_int64, C.c_int, _f32p]
        L.lbo_bruteforce_goheap.restype = C.c_int
        L.lbo_bruteforce_goheap.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.c_int64, C.c_int,
                                            C.c_int, _i64p, _f32p]
        L.lbo_topk_canonical.restype = C.c_int
        L.lbo_topk_canonical.argtypes = [_f32p, C.c_int64, C.c_int, _i64p, _f32p]
        L.lbo_search_batch.restype = None
        L.lbo_search_batch.argtypes = [C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int64, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p, _i64p, _f32p, C.c_int]
        L.lbo_build_adc_table.restype = None
        L.lbo_build_adc_table.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p]
        L.lbo_adc_batch.restype = None
        L.lbo_adc_batch.argtypes = [_f32p, _u8p, C.c_int, C.c_int64, _f32p]
        L.lbo_adc_single.restype = C.c_float
        L.lbo_adc_single.argtypes = [_f32p, _u8p, C.c_int, C.c_int]
        L.lbo_pq_encode.restype = None
        L.lbo_pq_encode.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, _u8p]
        L.lbo_pq_decode.restype = None
        L.lbo_pq_decode.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _u8p, _f32p]
        L.lbo_pq_parse_blob.restype = C.c_int
        L.lbo_pq_parse_blob.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lbo_fnv1a32.restype = C.c_uint32
        L.lbo_fnv1a32.argtypes = [C.c_char_p, C.c_size_t]
        L.lbo_ring_build.restype = C.c_int
        L.lbo_ring_build.argtypes = [C.c_int, C.c_int, _u32p, _i32p]
        L.lbo_ring_get_shard.restype = C.c_int
        L.lbo_ring_get_shard.argtypes = [_u32p, _i32p, C.c_int, C.c_uint64]
        L.lbo_merge_sorted_streams.restype = C.c_int
        L.lbo_merge_sorted_streams.argtypes = [_i64p, _f32p, _i32p, C.c_int, C.c_int, _i64p, _f32p]
        L.lbo_match_int64.restype = None
        L.lbo_match_int64.argtypes = [_i64p, C.c_int64, C.c_int64, C.c_int, _u8p]
        L.lbo_match_float32.restype = None
        L.lbo_match_float32.argtypes = [_f32p, C.c_int64, C.c_float, C.c_int, _u8p]
        L.lbo_and_bytes.restype = None
        L.lbo_and_bytes.argtypes = [_u8p, _u8p, C.c_int64]
        L.lbo_adaptive_limit.restype = C.c_int
        L.lbo_adaptive_limit.argtypes = [C.c_int, C.c_uint64, C.c_int]
        L.lbo_rrf.restype = C.c_int
        L.lbo_rrf.argtypes = [_i64p, C.c_int, _i64p, C.c_int, C.c_int, C.c_int, _i64p, _f32p]
        L.lbo_fill_uniform.restype = None
        L.lbo_fill_uniform.argtypes = [_f32p, C.c_int64, C.c_uint64, C.c_int64]
        L.lbo_fill_codes.restype = None
        L.lbo_fill_codes.argtypes = [_u8p, C.c_int64, C.c_uint64, C.c_int64]
        L.lbo_cpu_baseline.restype = C.c_double
        L.lbo_cpu_baseline.argtypes = [C.c_int, _f32p, C.c_int, _f32p, C.c_int64, C.c_int, C.c_int,
                                       C.c_int, C.c_int, _i64p, _f32p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def l2sq(a, b, order=SEQ):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().lbo_l2sq(a, b, a.size, order))


def euclidean(a, b, order=SEQ):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().lbo_euclidean(a, b, a.size, order))


def cosine(a, b, order=SEQ):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().lbo_cosine(a, b, a.size, order))


def dot(a, b, order=SEQ):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().lbo_dot(a, b, a.size, order))


def batch_flat(metric, q, flat, order=SEQ):
    q, flat = _f32(q), _f32(flat)
    n, dims = flat.shape
    out = np.empty(n, np.float32)
    lib().lbo_batch_flat(metric, order, q, flat, n, dims, out)
    return out


def bruteforce_goheap(metric, q, flat, k, order=SEQ):
    q, flat = _f32(q), _f32(flat)
    n, dims = flat.shape if flat.ndim == 2 else (0, q.size)
    ids = np.empty(max(k, 1), np.int64)
    dist = np.empty(max(k, 1), np.float32)
    cnt = lib().lbo_bruteforce_goheap(metric, order, q, flat.reshape(-1), n, dims, k, ids, dist)
    return ids[:cnt].copy(), dist[:cnt].copy()


def topk_canonical(dist, k):
    dist = _f32(dist)
    ids = np.empty(max(k, 1), np.int64)
    out = np.empty(max(k, 1), np.float32)
    cnt = lib().lbo_topk_canonical(dist, dist.size, k, ids, out)
    return ids[:k], out[:k], cnt


def search_batch(metric, queries, flat, k, order=SEQ, mask=None, ids=None, nthreads=1):
    queries, flat = _f32(queries), _f32(flat)
    nq, dims = queries.shape
    n = flat.shape[0]
    out_ids = np.empty((nq, k), np.int64)
    out_dist = np.empty((nq, k), np.float32)
    mptr = None
    iptr = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
        mptr = mask.ctypes.data
    if ids is not None:
        ids = np.ascontiguousarray(ids, np.int64)
        iptr = ids.ctypes.data
    lib().lbo_search_batch(metric, order, queries, nq, flat.reshape(-1), n, dims, k, mptr, iptr,
                           out_ids, out_dist, nthreads)
    return out_ids, out_dist


def build_adc_table(codebooks, query):
    cb = _f32(codebooks)
    M, K, sub = cb.shape
    table = np.empty(M * K, np.float32)
    lib().lbo_build_adc_table(cb, M, K, sub, _f32(query), table)
    return table


def adc_batch(table, codes):
    codes = np.ascontiguousarray(codes, np.uint8)
    n, m = codes.shape
    out = np.empty(n, np.float32)
    lib().lbo_adc_batch(_f32(table), codes, m, n, out)
    return out


def adc_single(table, code, K):
    code = np.ascontiguousarray(code, np.uint8)
    return np.float32(lib().lbo_adc_single(_f32(table), code, code.size, K))


def pq_encode(codebooks, vec):
    cb = _f32(codebooks)
    M, K, sub = cb.shape
    codes = np.empty(M, np.uint8)
    lib().lbo_pq_encode(cb, M, K, sub, _f32(vec), codes)
    return codes


def pq_decode(codebooks, codes):
    cb = _f32(codebooks)
    M, K, sub = cb.shape
    vec = np.empty(M * sub, np.float32)
    lib().lbo_pq_decode(cb, M, K, sub, np.ascontiguousarray(codes, np.uint8), vec)
    return vec


def pq_parse_blob(blob):
    d, m, k = C.c_int(), C.c_int(), C.c_int()
    rc = lib().lbo_pq_parse_blob(bytes(blob), len(blob), C.byref(d), C.byref(m), C.byref(k))
    return rc, d.value, m.value, k.value


def fnv1a32(b):
    return int(lib().lbo_fnv1a32(bytes(b), len(b)))


class Ring:
    """RingSharder(num_shards, vnodes) -- internal/store/sharding_strategy.go:40-127."""

    def __init__(self, num_shards, vnodes=40):
        v = vnodes if vnodes > 0 else 20
        self.hashes = np.empty(num_shards * v, np.uint32)
        self.owners = np.empty(num_shards * v, np.int32)
        self.n = lib().lbo_ring_build(num_shards, vnodes, self.hashes, self.owners)

    def get_shard(self, vid):
        return lib().lbo_ring_get_shard(self.hashes, self.owners, self.n, int(vid))


def merge_sorted_streams(lists, k):
    """lists: [(ids, scores), ...] each ascending by score."""
    lens = np.array([len(x[0]) for x in lists], np.int32)
    ids = np.concatenate([np.asarray(x[0], np.int64) for x in lists]) if lists else np.empty(0, np.int64)
    sc = np.concatenate([np.asarray(x[1], np.float32) for x in lists]) if lists else np.empty(0, np.float32)
    total = int(lens.sum())
    oi = np.empty(max(total, 1), np.int64)
    os_ = np.empty(max(total, 1), np.float32)
    cnt = lib().lbo_merge_sorted_streams(np.ascontiguousarray(ids), np.ascontiguousarray(sc), lens,
                                         len(lists), k, oi, os_)
    return oi[:cnt].copy(), os_[:cnt].copy()


def match_int64(src, val, op):
    src = np.ascontiguousarray(src, np.int64)
    dst = np.empty(src.size, np.uint8)
    lib().lbo_match_int64(src, src.size, int(val), int(op), dst)
    return dst


def match_float32(src, val, op):
    src = _f32(src)
    dst = np.empty(src.size, np.uint8)
    lib().lbo_match_float32(src, src.size, float(val), int(op), dst)
    return dst


def and_bytes(dst, src):
    dst = np.ascontiguousarray(dst, np.uint8).copy()
    src = np.ascontiguousarray(src, np.uint8)
    lib().lbo_and_bytes(dst, src, dst.size)
    return dst


def rrf(dense_ids, sparse_ids, k=60, limit=0):
    d = np.ascontiguousarray(dense_ids, np.int64)
    s_ = np.ascontiguousarray(sparse_ids, np.int64)
    cap = max(d.size + s_.size, 1)
    oi = np.empty(cap, np.int64)
    os_ = np.empty(cap, np.float32)
    cnt = lib().lbo_rrf(d if d.size else np.zeros(1, np.int64), d.size, s_ if s_.size else np.zeros(1, np.int64), s_.size,
                        k, limit, oi, os_)
    return oi[:cnt].copy(), os_[:cnt].copy()


def fill_uniform(n, seed, offset=0):
    out = np.empty(int(n), np.float32)
    lib().lbo_fill_uniform(out, out.size, seed, offset)
    return out


def fill_codes(n, seed, offset=0):
    out = np.empty(int(n), np.uint8)
    lib().lbo_fill_codes(out, out.size, seed, offset)
    return out


def cpu_baseline(metric, queries, flat, k, nthreads, simd=1):
    queries, flat = _f32(queries), _f32(flat)
    nq, dims = queries.shape
    oi = np.empty((nq, k), np.int64)
    od = np.empty((nq, k), np.float32)
    secs = lib().lbo_cpu_baseline(metric, queries, nq, flat.reshape(-1), flat.shape[0], dims, k,
                                  nthreads, simd, oi, od)
    return secs, oi, od


def adaptive_limit(k, matches, total):
    """calculateAdaptiveLimit (internal/store/adaptive_search.go:7-39)"""
    return int(lib().lbo_adaptive_limit(int(k), int(matches), int(total)))
