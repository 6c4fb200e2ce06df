"""Mirror of internal/gpu: Index{Add, Search, Close}, GPUConfig, NewIndex, NewIndexWithConfig
(internal/gpu/interface.go:3-19, gpu_enabled.go:8-21, faiss_gpu.go:44-167), over the C ABI.

Superset used by the batched path: SearchBatch, metric selection, device-resident
add/search (pointers are plain ints, e.g. torch_tensor.data_ptr()).
"""
import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import Canceled, DeadlineExceeded, GPUNotAvailable, LongbowGPUError  # noqa: F401
from .simd import MetricType, Order

ErrGPUNotAvailable = GPUNotAvailable


class GPUConfig:
    def __init__(self, DeviceID=0, Dimension=128, Metric=MetricType.Euclidean):
        self.DeviceID = DeviceID
        self.Dimension = Dimension
        self.Metric = Metric


class Cancel:
    """The cancellation half of a context.Context for ONE search call (lb_cancel): fire() from any thread, or
    set a deadline; pass as ctx= to Search / SearchBatch / search_device."""

    def __init__(self, deadline_ms=None, lib=None):
        self._lib = lib or _lib.load()
        self._h = C.c_void_p(self._lib.lb_cancel_new())
        if deadline_ms is not None:
            self.set_deadline_ms(deadline_ms)

    def fire(self):
        self._lib.lb_cancel_fire(self._h)

    def set_deadline_ms(self, ms):
        self._lib.lb_cancel_set_deadline_ms(self._h, int(ms))

    @property
    def state(self):
        return int(self._lib.lb_cancel_state(self._h))

    def close(self):
        if self._h:
            self._lib.lb_cancel_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """gpu.Index backed by lb_gpu_index (HIP)."""

    def __init__(self, cfg: GPUConfig, lib=None):
        if cfg.Dimension <= 0:
            raise ValueError(f"dimension must be positive, got {cfg.Dimension}")  # faiss_gpu.go:46-48
        if lib is None:
            lib = _lib.require_gpu(cfg.DeviceID)
        elif lib.lb_gpu_device_count() <= cfg.DeviceID:
            raise GPUNotAvailable(3, f"device {cfg.DeviceID} requested")
        st = C.c_int(0)
        h = lib.lb_gpu_index_new(cfg.DeviceID, cfg.Dimension, int(cfg.Metric), C.byref(st))
        if not h:
            _lib.check(st.value or 7)
        self._lib = lib
        self._h = C.c_void_p(h)
        self.dim = cfg.Dimension
        self.device = cfg.DeviceID
        self.metric = MetricType(int(cfg.Metric))
        self._lock = threading.Lock()
        self._closed = False

    # -- gpu.Index ---------------------------------------------------------------
    def Add(self, ids, vectors):
        """Add(ids []int64, vectors []float32) error  (faiss_gpu.go:75-104)"""
        self._live()
        vectors = np.ascontiguousarray(vectors, np.float32).reshape(-1)
        if vectors.size % self.dim != 0:
            raise ValueError(f"vector data length {vectors.size} not divisible by dimension {self.dim}")
        n = vectors.size // self.dim
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, np.int64)
            if ids.size != n:
                raise ValueError(f"id count {ids.size} does not match vector count {n}")
            idp = ids.ctypes.data
        _lib.check(self._lib.lb_gpu_index_add(self._h, n, vectors.ctypes.data, idp), self._h, lib=self._lib)

    def Search(self, vector, k, ctx=None):
        """Search(vector []float32, k int) (ids []int64, distances []float32, err)  (faiss_gpu.go:107-144)"""
        self._live()
        vector = np.ascontiguousarray(vector, np.float32).reshape(-1)
        if vector.size != self.dim:
            raise ValueError(f"query vector dimension {vector.size} does not match index dimension {self.dim}")
        ids, dist = self.SearchBatch(vector[None, :], k, ctx=ctx)
        return ids[0], dist[0]

    def Close(self):
        """idempotent (faiss_gpu.go:147-167)"""
        with self._lock:
            if self._closed:
                return
            h, self._h = self._h, None  # no other thread may be inside a call on this index (as for the C handle)
            self._closed = True
            self._lib.lb_gpu_index_free(h)

    # -- superset ----------------------------------------------------------------
    def SearchBatch(self, queries, k, ctx=None):
        self._live()
        queries = np.ascontiguousarray(queries, np.float32)
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise ValueError(f"query vector dimension {queries.shape[-1]} does not match index dimension {self.dim}")
        nq = queries.shape[0]
        dist = np.empty((nq, k), np.float32)
        labels = np.empty((nq, k), np.int64)
        _lib.check(self._lib.lb_gpu_index_search_ctx(self._h, nq, queries.ctypes.data, k, dist.ctypes.data,
                                                     labels.ctypes.data, ctx._h if ctx is not None else None), self._h, lib=self._lib)
        return labels, dist

    def Rerank(self, query, rows, order=Order.Unroll4, want_score=True):
        """The distance step of processChunkInternal (internal/store/parallel_search.go:274-364): distances of
        the resident rows `rows` (positions) to `query` + Score = 1/(1+d).  The reference runs
        simd.EuclideanDistanceBatchFlat here (4-accumulator order), hence the default order."""
        self._live()
        query = np.ascontiguousarray(query, np.float32).reshape(-1)
        if query.size != self.dim:
            raise ValueError(f"query vector dimension {query.size} does not match index dimension {self.dim}")
        rows = np.ascontiguousarray(rows, np.int64).reshape(-1)
        dist = np.empty(rows.size, np.float32)
        score = np.empty(rows.size, np.float32) if want_score else None
        _lib.check(self._lib.lb_gpu_index_rerank(self._h, query.ctypes.data, rows.ctypes.data, rows.size,
                                                 -1 if order is None else int(Order(order)), dist.ctypes.data,
                                                 score.ctypes.data if want_score else None), self._h, lib=self._lib)
        return (dist, score) if want_score else dist

    def rerank_device(self, d_query, d_rows, n, d_dist, d_score=None, order=Order.Unroll4, stream=None):
        self._live()
        _lib.check(self._lib.lb_gpu_index_rerank_device(self._h, d_query, d_rows, n, -1 if order is None else int(Order(order)),
                                                        d_dist, d_score, stream), self._h, lib=self._lib)

    def add_device(self, n, d_vectors, d_ids=None):
        self._live()
        _lib.check(self._lib.lb_gpu_index_add_device(self._h, n, d_vectors, d_ids), self._h, lib=self._lib)

    def search_device(self, nq, d_queries, k, d_dist, d_labels, stream=None, ctx=None):
        self._live()
        _lib.check(self._lib.lb_gpu_index_search_device_ctx(self._h, nq, d_queries, k, d_dist, d_labels, stream,
                                                            ctx._h if ctx is not None else None), self._h, lib=self._lib)

    def reserve(self, n_total):
        self._live()
        _lib.check(self._lib.lb_gpu_index_reserve(self._h, n_total), self._h, lib=self._lib)

    def set_order(self, order):
        self._live()
        _lib.check(self._lib.lb_gpu_index_set_order(self._h, int(Order(order))), self._h, lib=self._lib)

    def set_candidate_mode(self, mode):
        """lb_candidate_mode: 3 = AUTO (default: cheapest exact route), 0 = strict f32 MFMA beyond 384 queries,
        1 = split-bf16 corpus image, 2 = split in registers, 4 = one fp16 product; results identical in every mode"""
        self._live()
        _lib.check(self._lib.lb_gpu_index_set_candidate_mode(self._h, int(mode)), self._h, lib=self._lib)

    def set_f16_image(self, mode):
        """1 (default): keep an fp16 copy of the corpus for the single-product route while it pays and fits; 0: never"""
        self._live()
        _lib.check(self._lib.lb_gpu_index_set_f16_image(self._h, int(mode)), self._h, lib=self._lib)

    @property
    def f16_image_bytes(self):
        self._live()
        return int(self._lib.lb_gpu_index_f16_image_bytes(self._h))

    def set_search_combining(self, enable):
        """1 (default): concurrent host-pointer searches of a few queries each are answered by one batched search (lists identical
        to the single searches'); 0: every call searches on its own"""
        self._live()
        _lib.check(self._lib.lb_gpu_index_set_search_combining(self._h, 1 if enable else 0), self._h, lib=self._lib)

    @property
    def combining_stats(self):
        """(combined batches run, requests they answered)"""
        self._live()
        out = (C.c_int64 * 2)()
        _lib.check(self._lib.lb_gpu_index_combining_stats(self._h, out), self._h, lib=self._lib)
        return int(out[0]), int(out[1])

    def set_filter(self, mask):
        self._live()
        if mask is None:
            _lib.check(self._lib.lb_gpu_index_set_filter(self._h, None, 0), self._h, lib=self._lib)
            return
        mask = np.ascontiguousarray(mask, np.uint8)
        _lib.check(self._lib.lb_gpu_index_set_filter(self._h, mask.ctypes.data, mask.size), self._h, lib=self._lib)

    def filter_column(self, column, operator, value, validity=None, validity_offset=0, combine=False):
        """Evaluate `column OP value` on the device into the row mask (query.Filter semantics,
        internal/query/filter_evaluator.go:79-115,205-241).  column: int64 or float32, ntotal values;
        validity: Arrow LSB validity bitmap bytes or None; combine=True ANDs into the current mask."""
        from .simd import parse_operator
        self._live()
        op = int(parse_operator(operator))
        column = np.ascontiguousarray(column)
        vptr = None
        if validity is not None:
            validity = np.ascontiguousarray(np.frombuffer(validity, np.uint8) if not isinstance(validity, np.ndarray) else validity, np.uint8)
            vptr = validity.ctypes.data
        if column.dtype == np.int64:
            rc = self._lib.lb_gpu_index_filter_int64(self._h, column.ctypes.data, column.size, int(value), op, vptr,
                                                     validity_offset, 1 if combine else 0)
        elif column.dtype == np.float32:
            rc = self._lib.lb_gpu_index_filter_float32(self._h, column.ctypes.data, column.size, float(value), op,
                                                       vptr, validity_offset, 1 if combine else 0)
        else:
            raise TypeError(f"unsupported filter column type {column.dtype} (int64 / float32)")
        _lib.check(rc, self._h, lib=self._lib)

    def set_profiling(self, on):
        self._live()
        _lib.check(self._lib.lb_gpu_index_set_profiling(self._h, 1 if on else 0), self._h, lib=self._lib)

    def last_timing(self):
        """{class: (ms, launches)} of the last search: gemm, select, rerank, scan, total"""
        self._live()
        ms = (C.c_float * 5)()
        n = (C.c_int * 5)()
        _lib.check(self._lib.lb_gpu_index_last_timing(self._h, ms, n), self._h, lib=self._lib)
        names = ["gemm", "select", "rerank", "scan", "total"]
        return {names[i]: (float(ms[i]), int(n[i])) for i in range(5)}

    @property
    def ntotal(self):
        self._live()
        return int(self._lib.lb_gpu_index_ntotal(self._h))

    ROUTE_NAMES = {0: "exact scan", 1: "narrow 256x32", 2: "narrow 128x64", 4: "wide 128x128 f32 MFMA",
                   5: "tall 256x256 split-bf16", 6: "tall 256x256 fp16 single product",
                   7: "one query tile (64/128) fp16 single product over the fp16 copy"}

    @property
    def last_route(self):
        """(kind, operand form, name) of the kernel that generated the last batched search's candidates"""
        self._live()
        v = int(self._lib.lb_gpu_index_last_route(self._h))
        return v // 10, v % 10, self.ROUTE_NAMES.get(v // 10, "?")

    @property
    def fused_giveups(self):
        """searches whose in-launch threshold hand-off gave up (~1 ms) and were redone exactly (cumulative)"""
        self._live()
        return int(self._lib.lb_gpu_index_fused_giveups(self._h))

    @property
    def last_fallbacks(self):
        self._live()
        return int(self._lib.lb_gpu_index_last_fallbacks(self._h))

    def _live(self):
        if self._closed:
            raise LongbowGPUError(2)  # "index is closed"

    def __del__(self):  # runtime.SetFinalizer(idx, Close)  (faiss_gpu.go:69)
        try:
            self.Close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.Close()


def NewIndex():
    """gpu.NewIndex(): device 0, dimension 128 (gpu_enabled.go:8-14)"""
    return NewIndexWithConfig(GPUConfig(DeviceID=0, Dimension=128))


def NewIndexWithConfig(cfg: GPUConfig, lib=None):
    """gpu.NewIndexWithConfig (gpu_enabled.go:17-21).  Raises ErrGPUNotAvailable without a device.
    lib: another build of the library (tests: _lib.load_diag())."""
    return Index(cfg, lib=lib)
