"""Arrow IPC framing in the C ABI (lb_flight_*): the response writer is read back with pyarrow, the request
reader is fed pyarrow-written streams.  CPU part: encoding, and every status the reference handler returns
before it searches (internal/store/vector_search_exchange.go:43-147).  GPU part: the whole exchange and IPC
ingestion against the pyarrow mirror (longbow_amd/arrow_io.py) and the oracle."""
import ctypes as C

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
F = np.float32


def _lib():
    from longbow_amd import _lib
    try:
        return _lib.load()
    except (RuntimeError, OSError) as e:
        pytest.skip(f"liblongbow_gpu.so not loadable here: {e}")


def _ipc(batch):
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, batch.schema) as w:
        w.write_batch(batch)
    return sink.getvalue().to_pybytes()


def _take(lib, p, n):
    data = C.string_at(p, n.value)
    lib.lb_flight_free_buffer(p)
    return data


def _exchange(lib, reg, req_bytes):
    out, n = C.c_void_p(), C.c_size_t()
    err = C.create_string_buffer(512)
    rc = lib.lb_flight_vector_search_exchange(reg, req_bytes, len(req_bytes), C.byref(out), C.byref(n), err, 512)
    if rc != 0:
        return rc, err.value.decode(), None
    return 0, "", pa.ipc.open_stream(_take(lib, out, n)).read_all()


def _request(dataset="ds", k=5, q=None, fixed=True, extra_first=False, with_k=True, with_ds=True, with_q=True, rows=1):
    q = np.arange(8, dtype=F) if q is None else q
    cols, names = [], []
    if extra_first:
        cols.append(pa.array([["x", "y"]] * rows, pa.list_(pa.string())))
        names.append("tags")
    if with_ds:
        cols.append(pa.array([dataset] * rows, pa.string()))
        names.append("dataset")
    if with_k:
        cols.append(pa.array([k] * rows, pa.int32()))
        names.append("k")
    cols.append(pa.array([64] * rows, pa.int32()))
    names.append("ef")
    if with_q:
        if fixed:
            cols.append(pa.FixedSizeListArray.from_arrays(pa.array(np.tile(q, rows), pa.float32()), len(q)))
        else:
            cols.append(pa.array([q.tolist()] * rows, pa.list_(pa.float32())))
        names.append("query_vector")
    return _ipc(pa.record_batch(cols, names=names))


@pytest.mark.parametrize("n", [0, 1, 7, 100])
def test_encode_results_is_a_valid_ipc_stream(n):
    lib = _lib()
    rng = np.random.default_rng(n)
    ids = rng.integers(0, 2**40, n).astype(np.int64)
    scores = rng.random(n, dtype=F)
    pad_ids = np.concatenate([ids, [-1, -1]]).astype(np.int64)          # FAISS-style padding is trimmed
    pad_sc = np.concatenate([scores, [3.4e38, 3.4e38]]).astype(F)
    out, ln = C.c_void_p(), C.c_size_t()
    assert lib.lb_flight_encode_results(pad_ids.ctypes.data, pad_sc.ctypes.data, pad_ids.size, C.byref(out), C.byref(ln)) == 0
    t = pa.ipc.open_stream(_take(lib, out, ln)).read_all()
    assert t.schema == pa.schema([pa.field("id", pa.uint64(), nullable=False), pa.field("score", pa.float32(), nullable=False)]) or \
        t.schema.equals(pa.schema([pa.field("id", pa.uint64()), pa.field("score", pa.float32())]), check_metadata=False)
    assert t.num_rows == n
    assert np.array_equal(t.column("id").to_numpy(), ids.astype(np.uint64))
    assert np.array_equal(t.column("score").to_numpy(), scores)


def test_exchange_statuses_before_the_search():
    lib = _lib()
    reg = lib.lb_flight_datasets_new()
    try:
        # codes: InvalidArgument 3, NotFound 5, Internal 13
        assert _exchange(lib, reg, _request(with_ds=False))[:2] == (3, "missing 'dataset' column")
        assert _exchange(lib, reg, _request(with_q=False))[:2] == (3, "missing 'query_vector' column")
        rc, msg, _ = _exchange(lib, reg, _request(dataset="nope"))
        assert (rc, msg) == (5, "dataset not found: nope")
        rc, msg, _ = _exchange(lib, reg, _request(dataset="nope", fixed=False, extra_first=True, with_k=False))
        assert (rc, msg) == (5, "dataset not found: nope")  # List<f32> vector, a nested column in front, default k
        empty = _request(rows=1)
        sch = pa.ipc.open_stream(empty).schema
        sink = pa.BufferOutputStream()
        with pa.ipc.new_stream(sink, sch):
            pass
        assert _exchange(lib, reg, sink.getvalue().to_pybytes())[:2] == (3, "empty search request")
        zero = pa.ipc.open_stream(empty).read_next_batch().slice(0, 0)
        assert _exchange(lib, reg, _ipc(zero))[:2] == (3, "empty search request parameters")
        bad_q = _ipc(pa.record_batch([pa.array(["ds"]), pa.array([[1, 2]], pa.list_(pa.int32()))], names=["dataset", "query_vector"]))
        assert _exchange(lib, reg, bad_q)[:2] == (3, "unsupported query_vector type")
        # garbage / truncated input never crashes
        good = _request()
        for cut in (0, 3, 8, 40, len(good) // 2, len(good) - 9):
            rc, msg, _ = _exchange(lib, reg, good[:cut])
            assert rc in (3, 13), (cut, rc, msg)
        rng = np.random.default_rng(0)
        for _ in range(200):
            b = bytearray(good)
            for i in rng.integers(8, len(b), 6):
                b[i] = int(rng.integers(0, 256))
            rc, msg, _ = _exchange(lib, reg, bytes(b))
            assert rc in (0, 3, 5, 13, 14)
    finally:
        lib.lb_flight_datasets_free(reg)


@pytest.mark.gpu
def test_exchange_and_ingestion_match_the_pyarrow_mirror(oracle):
    from tests.gpu_util import gpu_or_skip, new_index
    gpu_or_skip()
    lib = _lib()
    from longbow_amd import arrow_io
    rng = np.random.default_rng(5)
    n, d = 5000, 64
    X = rng.random((n, d), dtype=F)
    ids = (rng.permutation(n).astype(np.uint64) * 3 + 1)
    batches = []
    for a in range(0, n, 1250):  # four batches; the values buffers go to the library as they lie in the IPC body
        v = pa.FixedSizeListArray.from_arrays(pa.array(X[a:a + 1250].reshape(-1), pa.float32()), d)
        batches.append(pa.record_batch([pa.array(ids[a:a + 1250], pa.uint64()), v, pa.array([f"r{i}" for i in range(a, a + 1250)])],
                                       names=["id", "vector", "note"]))
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, batches[0].schema) as w:
        for b in batches:
            w.write_batch(b)
    stream = sink.getvalue().to_pybytes()
    idx = new_index(d, 0)
    added = C.c_int64()
    err = C.create_string_buffer(512)
    assert lib.lb_flight_index_add_ipc(idx._h, stream, len(stream), C.byref(added), err, 512) == 0, err.value
    assert added.value == n and idx.ntotal == n
    ds = arrow_io.GPUDataset("mirror", d)
    assert ds.add_ipc_stream(stream) == n
    reg = lib.lb_flight_datasets_new()
    assert lib.lb_flight_datasets_put(reg, b"emb", idx._h) == 0
    try:
        for fixed in (True, False):
            for k in (1, 10, 37):
                q = rng.random(d, dtype=F)
                req = _request("emb", k, q, fixed=fixed)
                rc, msg, t = _exchange(lib, reg, req)
                assert rc == 0, msg
                want = pa.ipc.open_stream(arrow_io.handle_vector_search_exchange({"emb": ds}, req)).read_all()
                assert t.column("id").to_pylist() == want.column("id").to_pylist()
                assert np.array_equal(t.column("score").to_numpy(), want.column("score").to_numpy())
                oi, od = oracle.search_batch(0, q[None, :], X, k, ids=ids.astype(np.int64))
                assert np.array_equal(t.column("id").to_numpy().astype(np.int64), oi[0])
                assert np.array_equal(t.column("score").to_numpy(), od[0])
        # k > N rows: min(k, N) results; wrong dimension; removed dataset
        small = new_index(d, 0)
        small.Add(None, X[:3])
        lib.lb_flight_datasets_put(reg, b"small", small._h)
        rc, msg, t = _exchange(lib, reg, _request("small", 10, X[1]))
        assert rc == 0 and t.num_rows == 3 and t.column("id").to_pylist()[0] == 1
        rc, msg, _ = _exchange(lib, reg, _request("emb", 5, np.zeros(8, F)))
        assert (rc, msg) == (3, f"dimension mismatch: expected {d}, got 8")
        lib.lb_flight_datasets_put(reg, b"small", None)
        assert _exchange(lib, reg, _request("small", 5, X[1]))[0] == 5
        small.Close()
        # ingestion errors
        bad = _ipc(pa.record_batch([pa.FixedSizeListArray.from_arrays(pa.array(np.zeros(16, F)), 8)], names=["vector"]))
        assert lib.lb_flight_index_add_ipc(idx._h, bad, len(bad), C.byref(added), err, 512) == 3
        assert b"dimension mismatch" in err.value
    finally:
        lib.lb_flight_datasets_free(reg)
        ds.close()
        idx.Close()
