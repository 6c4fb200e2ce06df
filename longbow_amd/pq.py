"""Mirror of internal/pq's query-side interface (the ADC path), computed by HIP kernels.

PQEncoder.{BuildADCTable, ADCDistanceBatch, Serialize/Deserialize blob}
(internal/pq/adc_table.go:15-72, persistence.go:9-73) and Encode / Decode (encoder.go:76-158).
Training (k-means, unseeded) is an offline step in the reference and stays out of the GPU path:
codebooks are an input.
"""
import ctypes as C
import struct

import numpy as np

from . import _lib


def serialize_codebooks(codebooks):
    """PQEncoder.Serialize layout (persistence.go:9-35): u32 LE dims, M, K + M*K*SubDim f32 LE."""
    cb = np.ascontiguousarray(codebooks, np.float32)
    M, K, sub = cb.shape
    return struct.pack("<III", M * sub, M, K) + cb.astype("<f4").tobytes()


class PQEncoder:
    """Query-side PQEncoder on the GPU, built from the reference's serialised blob."""

    def __init__(self, blob, device=0, lib=None):
        lib = lib or _lib.require_gpu(device)
        st = C.c_int(0)
        buf = bytes(blob)
        h = lib.lb_gpu_pq_new(device, buf, len(buf), C.byref(st))
        if not h:
            if st.value == 1:
                raise ValueError("invalid PQ data")  # persistence.go:39-56 error cases
            _lib.check(st.value or 7)
        self._lib = lib
        self._h = C.c_void_p(h)
        self.M = lib.lb_gpu_pq_m(self._h)
        self.Dims = lib.lb_gpu_pq_dims(self._h)
        self.K = 256
        self.SubDim = self.Dims // self.M

    def add_codes(self, codes):
        codes = np.ascontiguousarray(codes, np.uint8).reshape(-1)
        if codes.size % self.M:
            raise ValueError("code length mismatch")
        _lib.check(self._lib.lb_gpu_pq_add_codes(self._h, codes.size // self.M, codes.ctypes.data), self._h, pq=True, lib=self._lib)

    def add_codes_device(self, n, d_codes):
        _lib.check(self._lib.lb_gpu_pq_add_codes_device(self._h, n, d_codes), self._h, pq=True, lib=self._lib)

    @property
    def ntotal(self):
        return int(self._lib.lb_gpu_pq_ntotal(self._h))

    def reserve(self, n_total):
        _lib.check(self._lib.lb_gpu_pq_reserve(self._h, n_total), self._h, pq=True, lib=self._lib)

    def get_codes(self, row0=0, n=None):
        """stored codes rows [row0, row0+n) -> uint8 [n, M]"""
        n = self.ntotal - row0 if n is None else n
        out = np.empty((n, self.M), np.uint8)
        _lib.check(self._lib.lb_gpu_pq_get_codes(self._h, row0, n, out.ctypes.data), self._h, pq=True, lib=self._lib)
        return out

    def Encode(self, vector):
        """PQEncoder.Encode (encoder.go:76-89): one vector -> M bytes; a 2-D array encodes row by row."""
        v = np.ascontiguousarray(vector, np.float32)
        single = v.ndim == 1
        v = v.reshape(-1, v.shape[-1])
        if v.shape[1] != self.Dims:
            raise ValueError("vector dimension mismatch")  # encoder.go:77-79
        codes = np.empty((v.shape[0], self.M), np.uint8)
        _lib.check(self._lib.lb_gpu_pq_encode(self._h, v.shape[0], v.ctypes.data, codes.ctypes.data), self._h, pq=True, lib=self._lib)
        return codes[0] if single else codes

    def Decode(self, codes):
        """PQEncoder.Decode (encoder.go:139-158)"""
        c = np.ascontiguousarray(codes, np.uint8)
        single = c.ndim == 1
        c = c.reshape(-1, c.shape[-1])
        if c.shape[1] != self.M:
            raise ValueError("code length mismatch")  # encoder.go:140-142
        out = np.empty((c.shape[0], self.Dims), np.float32)
        _lib.check(self._lib.lb_gpu_pq_decode(self._h, c.shape[0], c.ctypes.data, out.ctypes.data), self._h, pq=True, lib=self._lib)
        return out[0] if single else out

    def encode_device(self, n, d_vectors, d_codes, stream=None):
        _lib.check(self._lib.lb_gpu_pq_encode_device(self._h, n, d_vectors, d_codes, stream), self._h, pq=True, lib=self._lib)

    def add_vectors_device(self, n, d_vectors):
        """encode n device-resident vectors and append their codes"""
        _lib.check(self._lib.lb_gpu_pq_add_vectors_device(self._h, n, d_vectors), self._h, pq=True, lib=self._lib)

    def Rerank(self, query, rows):
        """processChunkInternal's PQ branch (parallel_search.go:292-345): ADC distance of the stored code rows
        + Score = 1/(1+d)"""
        query = np.ascontiguousarray(query, np.float32).reshape(-1)
        if query.size != self.Dims:
            raise ValueError("query dimension mismatch")
        rows = np.ascontiguousarray(rows, np.int64).reshape(-1)
        dist = np.empty(rows.size, np.float32)
        score = np.empty(rows.size, np.float32)
        _lib.check(self._lib.lb_gpu_pq_rerank(self._h, query.ctypes.data, rows.ctypes.data, rows.size, dist.ctypes.data,
                                              score.ctypes.data), self._h, pq=True, lib=self._lib)
        return dist, score

    def BuildADCTable(self, query):
        query = np.ascontiguousarray(query, np.float32).reshape(-1)
        if query.size != self.Dims:
            raise ValueError("query dimension mismatch")  # adc_table.go:16-18
        table = np.empty(self.M * self.K, np.float32)
        _lib.check(self._lib.lb_gpu_pq_build_adc_table(self._h, query.ctypes.data, table.ctypes.data), self._h, pq=True, lib=self._lib)
        return table

    def ADCDistanceBatch(self, table, results, row0=0):
        """distances of stored codes [row0, row0+len(results)) -> results (adc_table.go:57-72)"""
        if results.size == 0:
            return
        table = np.ascontiguousarray(table, np.float32)
        if table.size != self.M * self.K:
            raise ValueError("invalid table size")  # adc_table.go:64-66
        if row0 + results.size > self.ntotal:
            raise ValueError("flatCodes buffer too small")  # adc_table.go:61-63
        _lib.check(self._lib.lb_gpu_pq_adc_distance_batch(self._h, table.ctypes.data, row0, results.size,
                                                           results.ctypes.data), self._h, pq=True, lib=self._lib)

    def set_prefilter(self, on):
        """True (default): byte-table prefilter + exact survivors; False: exact f32-table pass only.  Same results."""
        _lib.check(self._lib.lb_gpu_pq_set_prefilter(self._h, 1 if on else 0), self._h, pq=True, lib=self._lib)

    def set_search_combining(self, enable):
        """1 (default): concurrent Search calls of a few queries each are answered by one batch (pairs of queries share a pass
        over the codes); 0: every call on its own.  Same results."""
        _lib.check(self._lib.lb_gpu_pq_set_search_combining(self._h, 1 if enable else 0), self._h, pq=True, lib=self._lib)

    @property
    def combining_stats(self):
        """(combined batches run, requests they answered)"""
        import ctypes as C
        out = (C.c_int64 * 2)()
        _lib.check(self._lib.lb_gpu_pq_combining_stats(self._h, out), self._h, pq=True, lib=self._lib)
        return int(out[0]), int(out[1])

    def Search(self, queries, k, ctx=None):
        queries = np.ascontiguousarray(queries, np.float32)
        if queries.ndim == 1:
            queries = queries[None, :]
        if queries.shape[1] != self.Dims:
            raise ValueError("query dimension mismatch")
        nq = queries.shape[0]
        dist = np.empty((nq, k), np.float32)
        labels = np.empty((nq, k), np.int64)
        _lib.check(self._lib.lb_gpu_pq_search_ctx(self._h, nq, queries.ctypes.data, k, dist.ctypes.data,
                                                  labels.ctypes.data, ctx._h if ctx is not None else None),
                   self._h, pq=True, lib=self._lib)
        return labels, dist

    def search_device(self, nq, d_queries, k, d_dist, d_labels, stream=None, ctx=None):
        _lib.check(self._lib.lb_gpu_pq_search_device_ctx(self._h, nq, d_queries, k, d_dist, d_labels, stream,
                                                         ctx._h if ctx is not None else None),
                   self._h, pq=True, lib=self._lib)

    def set_profiling(self, on):
        _lib.check(self._lib.lb_gpu_pq_set_profiling(self._h, 1 if on else 0), self._h, pq=True, lib=self._lib)

    def last_timing(self):
        """(ms of the last query's pass over the codes, ms of the whole search on the device)"""
        ms = (C.c_float * 2)()
        _lib.check(self._lib.lb_gpu_pq_last_timing(self._h, ms), self._h, pq=True, lib=self._lib)
        return float(ms[0]), float(ms[1])

    def Close(self):
        if self._h:
            self._lib.lb_gpu_pq_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass
