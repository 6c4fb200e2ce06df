"""Arrow on either side of the k-NN path (f-1 ingestion, f-2 DoExchange framing).

f-1  GPUDataset.add_record_batch / add_ipc_stream: the "vector" column must be
     FixedSizeList<float32>[dim] (internal/store/store_lifecycle.go:66-69); its values buffer is
     handed to lb_gpu_index_add as is (zero-copy view, no repacking; the library pins + DMAs).
     Other element types are up-cast to float32 like ExtractVectorFromArrow
     (internal/store/arrow_utils.go:198-260).  The "id" column (uint32 / uint64 / int64) supplies
     the reported ids, truncated to the reference's uint32 VectorID exactly as
     mapInternalToUserIDsLocked does (internal/store/store_query.go:459-530); without an "id"
     column the row position is the id.
f-2  handle_vector_search_exchange: request RecordBatch {dataset utf8, k int32 (default 10),
     ef int32 (ignored, as in the reference), query_vector FixedSizeList|List<float32>}, ROW 0 ONLY
     -> response RecordBatch {id uint64, score float32}, as Arrow IPC stream bytes
     (internal/store/vector_search_exchange.go:31-217).  Error codes/messages follow the gRPC
     statuses of the reference.
     handle_vector_search_action: the JSON `VectorSearchRequest` with `vector` and/or `vectors`
     (internal/query/requests.go:4-20): one result batch per query
     (internal/store/vector_search_action.go:25-231) -- computed as ONE batched GPU search.
"""
import json

import numpy as np
import pyarrow as pa

from . import gpu
from .simd import MetricType

RESPONSE_SCHEMA = pa.schema([pa.field("id", pa.uint64()), pa.field("score", pa.float32())])


class ExchangeError(Exception):
    """gRPC status of the reference handler: code in {InvalidArgument, NotFound, FailedPrecondition, Internal}"""

    def __init__(self, code, message):
        self.code = code
        super().__init__(f"{code}: {message}")


def _vector_values(col, dim_expected=None):
    """(n, dim) float32 view/copy of a FixedSizeList column"""
    if isinstance(col, pa.ChunkedArray):
        col = col.combine_chunks()
    t = col.type
    if not pa.types.is_fixed_size_list(t):
        raise ExchangeError("InvalidArgument", f"'vector' must be FixedSizeList, got {t}")
    dim = t.list_size
    if dim_expected is not None and dim != dim_expected:
        raise ExchangeError("InvalidArgument", f"dimension mismatch: expected {dim_expected}, got {dim}")
    if col.null_count:
        raise ExchangeError("InvalidArgument", "null vectors are not supported")
    values = col.flatten()  # accounts for the array offset
    arr = values.to_numpy(zero_copy_only=False)
    if arr.dtype != np.float32:
        arr = arr.astype(np.float32)  # ExtractVectorFromArrow up-cast
    return np.ascontiguousarray(arr).reshape(len(col), dim)


def _ids_from(batch):
    idx = batch.schema.get_field_index("id")
    if idx == -1:
        return None
    col = batch.column(idx)
    t = col.type
    if not (pa.types.is_uint32(t) or pa.types.is_uint64(t) or pa.types.is_int64(t)):
        return None  # store_query.go:505-530 handles these three; anything else keeps the internal id
    raw = col.to_numpy(zero_copy_only=False)
    return (raw.astype(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.int64)  # VectorID is uint32


class GPUDataset:
    def __init__(self, name, dim, metric=MetricType.Euclidean, device=0):
        self.name = name
        self.dim = dim
        self.index = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=device, Dimension=dim, Metric=metric))
        self._has_ids = False
        self._rows = 0

    def add_record_batch(self, batch):
        vidx = batch.schema.get_field_index("vector")
        if vidx == -1:
            raise ExchangeError("InvalidArgument", "missing 'vector' column")
        X = _vector_values(batch.column(vidx), self.dim)
        ids = _ids_from(batch)
        if ids is None and self._has_ids:
            ids = np.arange(self._rows, self._rows + len(X), dtype=np.int64)
        self.index.Add(ids, X.reshape(-1))
        self._has_ids = self._has_ids or ids is not None
        self._rows += len(X)
        return len(X)

    def add_ipc_stream(self, data):
        n = 0
        with pa.ipc.open_stream(data) as reader:
            for batch in reader:
                n += self.add_record_batch(batch)
        return n

    def close(self):
        self.index.Close()


def _result_batch(ids, scores):
    keep = ids >= 0  # fewer than k hits: the reference returns min(k, N) rows
    return pa.record_batch([pa.array(ids[keep].astype(np.uint64), pa.uint64()),
                            pa.array(scores[keep], pa.float32())], schema=RESPONSE_SCHEMA)


def _to_ipc(batch):
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, batch.schema) as w:
        w.write_batch(batch)
    return sink.getvalue().to_pybytes()


def handle_vector_search_exchange(datasets, request_ipc):
    """datasets: {name: GPUDataset}.  request_ipc: Arrow IPC stream bytes with one request batch."""
    try:
        reader = pa.ipc.open_stream(request_ipc)
        rec = reader.read_next_batch()
    except StopIteration:
        raise ExchangeError("InvalidArgument", "empty search request")
    except Exception as e:
        raise ExchangeError("Internal", f"failed to create record reader: {e}")
    if rec.num_rows == 0:
        raise ExchangeError("InvalidArgument", "empty search request parameters")

    def col(name):
        i = rec.schema.get_field_index(name)
        return None if i == -1 else rec.column(i)

    c = col("dataset")
    if c is None:
        raise ExchangeError("InvalidArgument", "missing 'dataset' column")
    name = c[0].as_py()
    ck = col("k")
    k = 10 if ck is None else int(ck[0].as_py())
    # 'ef' is parsed and ignored, as in the reference (vector_search_exchange.go:98-103,155-158)
    cv = col("query_vector")
    if cv is None:
        raise ExchangeError("InvalidArgument", "missing 'query_vector' column")
    t = cv.type
    if pa.types.is_fixed_size_list(t) or pa.types.is_list(t):
        q = np.asarray(cv[0].values.to_numpy(zero_copy_only=False), np.float32)  # row 0 only
    else:
        raise ExchangeError("InvalidArgument", f"unsupported query_vector type: {t}")
    ds = datasets.get(name)
    if ds is None:
        raise ExchangeError("NotFound", f"dataset not found: {name}")
    if q.size != ds.dim:
        raise ExchangeError("InvalidArgument", f"dimension mismatch: expected {ds.dim}, got {q.size}")
    try:
        ids, scores = ds.index.Search(q, k)
    except Exception as e:
        raise ExchangeError("Internal", f"search failed: {e}")
    return _to_ipc(_result_batch(ids, scores))


def handle_vector_search_action(datasets, body):
    """body: JSON VectorSearchRequest.  Returns one IPC-encoded result batch per query vector."""
    try:
        req = json.loads(body)
    except Exception as e:
        raise ExchangeError("InvalidArgument", f"invalid JSON request: {e}")
    k = int(req.get("k", 0))
    if k < 1:
        raise ExchangeError("InvalidArgument", "k must be at least 1")
    qs = []
    if req.get("vector"):
        qs.append(req["vector"])
    qs.extend(req.get("vectors") or [])
    if not qs:
        raise ExchangeError("InvalidArgument", "no query vector(s) provided")
    ds = datasets.get(req.get("dataset"))
    if ds is None:
        raise ExchangeError("NotFound", f"dataset not found: {req.get('dataset')}")
    for q in qs:
        if len(q) != ds.dim:
            raise ExchangeError("InvalidArgument", f"dimension mismatch: expected {ds.dim}, got {len(q)}")
    labels, dist = ds.index.SearchBatch(np.asarray(qs, np.float32), k)  # the sequential loop of the reference, batched
    return [_to_ipc(_result_batch(labels[i], dist[i])) for i in range(len(qs))]
