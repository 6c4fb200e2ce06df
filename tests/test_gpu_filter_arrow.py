"""SURVEY 8(f) rows on the GPU: predicate masks (f-3), Arrow ingestion (f-1), DoExchange framing (f-2)."""
import json

import numpy as np
import pytest

from tests.gpu_util import assert_same, gpu_or_skip, new_index

pytestmark = pytest.mark.gpu
F = np.float32


def test_match_kernels_vs_oracle(oracle):
    gpu_or_skip()
    from longbow_amd import simd
    rng = np.random.default_rng(1)
    for n in (1, 15, 16, 17, 1000, 100003):
        a = rng.integers(-50, 50, n).astype(np.int64)
        f = rng.standard_normal(n).astype(F)
        f[::11] = np.nan
        for op in range(6):
            dst = np.full(n, 7, np.uint8)
            simd.MatchInt64(a, 3, op, dst)
            assert np.array_equal(dst, oracle.match_int64(a, 3, op)), (n, op)
            simd.MatchFloat32(f, 0.25, op, dst)
            assert np.array_equal(dst, oracle.match_float32(f, 0.25, op)), (n, op)
    m1 = rng.integers(0, 2, 5000).astype(np.uint8)
    m2 = rng.integers(0, 2, 5000).astype(np.uint8)
    exp = oracle.and_bytes(m1, m2)
    simd.AndBytes(m1, m2)
    assert np.array_equal(m1, exp)
    with pytest.raises(ValueError):
        simd.MatchInt64(np.zeros(3, np.int64), 0, "eq", np.zeros(2, np.uint8))   # simd: length mismatch
    with pytest.raises(ValueError):
        simd.AndBytes(np.zeros(3, np.uint8), np.zeros(2, np.uint8))


def test_filtered_search_with_device_side_predicates(oracle):
    """config-5 style: int64 metadata column, predicate `< 10` (10 % selectivity), AND a float column,
    nulls never match"""
    gpu_or_skip()
    rng = np.random.default_rng(2)
    n, d = 30000, 64
    X = rng.random((n, d), dtype=F)
    Q = rng.random((40, d), dtype=F)
    meta = rng.integers(0, 100, n).astype(np.int64)
    price = rng.random(n).astype(F)
    valid = rng.random(n) > 0.05                       # 5 % nulls in `meta`
    bitmap = np.packbits(valid, bitorder="little")     # Arrow validity bitmap
    idx = new_index(d, 0)
    idx.Add(None, X)
    idx.filter_column(meta, "<", 10, validity=bitmap)
    mask = oracle.match_int64(meta, 10, 4) & valid.astype(np.uint8)
    for qs in (Q[:3], Q):
        lab, dist = idx.SearchBatch(qs, 10)
        oi, od = oracle.search_batch(0, qs, X, 10, mask=mask, nthreads=4)
        assert_same(lab, dist, oi, od)
    idx.filter_column(price, "ge", 0.5, combine=True)  # AND chain (simd.AndBytes semantics)
    mask2 = oracle.and_bytes(mask, oracle.match_float32(price, 0.5, 3))
    lab, dist = idx.SearchBatch(Q, 10)
    oi, od = oracle.search_batch(0, Q, X, 10, mask=mask2, nthreads=4)
    assert_same(lab, dist, oi, od)
    with pytest.raises(Exception):
        idx.filter_column(meta[:-1], "<", 10)          # wrong length
    idx.Close()


def _make_batch(pa, X, ids=None, id_type=None, name="vector"):
    n, d = X.shape
    vec = pa.FixedSizeListArray.from_arrays(pa.array(X.reshape(-1), pa.float32()), d)
    cols, names = [vec], [name]
    if ids is not None:
        cols.insert(0, pa.array(ids, id_type))
        names.insert(0, "id")
    return pa.record_batch(cols, names=names)


def test_arrow_ingestion_and_exchange(oracle):
    gpu_or_skip()
    pa = pytest.importorskip("pyarrow")
    from longbow_amd import arrow_io
    rng = np.random.default_rng(3)
    d = 32
    parts = [rng.random((n, d), dtype=F) for n in (1000, 1, 2500)]
    X = np.concatenate(parts)
    ids = np.arange(len(X), dtype=np.uint64) * 3 + 2**32 + 5      # > 2^32: truncated to uint32 like VectorID
    ds = arrow_io.GPUDataset("docs", d)
    sink = pa.BufferOutputStream()
    pos = 0
    w = None
    for p in parts:
        b = _make_batch(pa, p, ids[pos:pos + len(p)], pa.uint64())
        if w is None:
            w = pa.ipc.new_stream(sink, b.schema)
        w.write_batch(b)
        pos += len(p)
    w.close()
    assert ds.add_ipc_stream(sink.getvalue().to_pybytes()) == len(X)   # ingest from Arrow IPC bytes
    exp_ids = (ids & np.uint64(0xFFFFFFFF)).astype(np.int64)

    # DoExchange request: row 0 only, FixedSizeList and List forms, default k = 10
    q = rng.random(d, dtype=F)
    for form in ("fsl", "list"):
        qcol = (pa.FixedSizeListArray.from_arrays(pa.array(np.concatenate([q, q * 2]), pa.float32()), d) if form == "fsl"
                else pa.array([q.tolist(), (q * 2).tolist()], pa.list_(pa.float32())))
        req = pa.record_batch([pa.array(["docs", "docs"]), pa.array([7, 3], pa.int32()), pa.array([64, 64], pa.int32()), qcol],
                              names=["dataset", "k", "ef", "query_vector"])
        s = pa.BufferOutputStream()
        with pa.ipc.new_stream(s, req.schema) as wr:
            wr.write_batch(req)
        out = arrow_io.handle_vector_search_exchange({"docs": ds}, s.getvalue().to_pybytes())
        res = pa.ipc.open_stream(out).read_next_batch()
        assert res.schema == arrow_io.RESPONSE_SCHEMA                 # {id uint64, score float32}
        oi, od = oracle.search_batch(0, q[None], X, 7, ids=exp_ids)
        assert np.array_equal(res.column(0).to_numpy().astype(np.int64), oi[0])
        assert np.array_equal(res.column(1).to_numpy(), od[0])
    # default k
    req = pa.record_batch([pa.array(["docs"]), pa.FixedSizeListArray.from_arrays(pa.array(q, pa.float32()), d)],
                          names=["dataset", "query_vector"])
    s = pa.BufferOutputStream()
    with pa.ipc.new_stream(s, req.schema) as wr:
        wr.write_batch(req)
    res = pa.ipc.open_stream(arrow_io.handle_vector_search_exchange({"docs": ds}, s.getvalue().to_pybytes())).read_next_batch()
    assert res.num_rows == 10

    # error statuses of the reference handler
    def send(batch):
        s2 = pa.BufferOutputStream()
        with pa.ipc.new_stream(s2, batch.schema) as wr:
            wr.write_batch(batch)
        return arrow_io.handle_vector_search_exchange({"docs": ds}, s2.getvalue().to_pybytes())
    fq = pa.FixedSizeListArray.from_arrays(pa.array(q, pa.float32()), d)
    with pytest.raises(arrow_io.ExchangeError, match="missing 'dataset' column"):
        send(pa.record_batch([fq], names=["query_vector"]))
    with pytest.raises(arrow_io.ExchangeError, match="missing 'query_vector' column"):
        send(pa.record_batch([pa.array(["docs"])], names=["dataset"]))
    with pytest.raises(arrow_io.ExchangeError, match="dataset not found"):
        send(pa.record_batch([pa.array(["nope"]), fq], names=["dataset", "query_vector"]))
    with pytest.raises(arrow_io.ExchangeError, match="dimension mismatch: expected 32, got 4"):
        send(pa.record_batch([pa.array(["docs"]), pa.FixedSizeListArray.from_arrays(pa.array(q[:4], pa.float32()), 4)],
                             names=["dataset", "query_vector"]))

    # DoAction("VectorSearch"): `vector` + `vectors`, one result batch per query, computed as one GPU batch
    Q = rng.random((20, d), dtype=F)
    outs = arrow_io.handle_vector_search_action({"docs": ds}, json.dumps(
        {"dataset": "docs", "k": 5, "vector": Q[0].tolist(), "vectors": Q[1:].tolist()}))
    assert len(outs) == 20
    oi, od = oracle.search_batch(0, Q, X, 5, ids=exp_ids, nthreads=4)
    for i, o in enumerate(outs):
        r = pa.ipc.open_stream(o).read_next_batch()
        assert np.array_equal(r.column(0).to_numpy().astype(np.int64), oi[i])
        assert np.array_equal(r.column(1).to_numpy(), od[i])
    with pytest.raises(arrow_io.ExchangeError, match="k must be at least 1"):
        arrow_io.handle_vector_search_action({"docs": ds}, json.dumps({"dataset": "docs", "k": 0, "vector": q.tolist()}))
    with pytest.raises(arrow_io.ExchangeError, match="no query vector"):
        arrow_io.handle_vector_search_action({"docs": ds}, json.dumps({"dataset": "docs", "k": 3}))
    ds.close()


def test_arrow_upcast_and_missing_id_column(oracle):
    gpu_or_skip()
    pa = pytest.importorskip("pyarrow")
    from longbow_amd import arrow_io
    rng = np.random.default_rng(4)
    X64 = rng.random((500, 16))
    vec = pa.FixedSizeListArray.from_arrays(pa.array(X64.reshape(-1), pa.float64()), 16)
    ds = arrow_io.GPUDataset("f64", 16)
    ds.add_record_batch(pa.record_batch([vec], names=["vector"]))       # float64 -> float32 up-cast, ids = positions
    X = X64.astype(F)
    q = rng.random(16, dtype=F)
    ids, dist = ds.index.Search(q, 5)
    oi, od = oracle.search_batch(0, q[None], X, 5)
    assert np.array_equal(ids, oi[0]) and np.array_equal(dist, od[0])
    with pytest.raises(arrow_io.ExchangeError):
        ds.add_record_batch(pa.record_batch([pa.array([1, 2, 3])], names=["embedding"]))  # no "vector" column
    ds.close()


def test_rrf_fusion_on_gpu(oracle, golden):
    """store.ReciprocalRankFusion (hybrid_search_test.go:75-150) + batched fusion vs the oracle"""
    gpu_or_skip()
    from longbow_amd import hybrid
    for c in golden:
        if c["op"] != "rrf":
            continue
        ids, sc = hybrid.ReciprocalRankFusion(c["dense"], c["sparse"], c["k"], c["limit"])
        assert list(ids) == c["expected_ids"], c["name"]
        assert np.array_equal(sc, np.array(c["expected_scores"], F)), c["name"]
    assert len(hybrid.ReciprocalRankFusion(None, None, 60, 10)[0]) == 0
    rng = np.random.default_rng(9)
    nq, kd, ks, lim = 50, 200, 150, 100          # dense side asks for k*2 (hybrid_search.go:62)
    dense = np.stack([rng.permutation(1000)[:kd] for _ in range(nq)]).astype(np.int64)
    sparse = np.stack([rng.permutation(1000)[:ks] for _ in range(nq)]).astype(np.int64)
    sparse[3, 100:] = -1                          # ragged sparse list (padding)
    oi, os_ = hybrid.fuse_batch(dense, sparse, 60, lim)
    for q in range(nq):
        ei, es = oracle.rrf(dense[q], sparse[q], 60, lim)
        assert np.array_equal(oi[q, :len(ei)], ei) and np.array_equal(os_[q, :len(es)], es), q


def test_hybrid_candidate_handoff(oracle):
    """ArrowHNSW.SearchHybrid's GPU hand-off: min(10k, Len) candidates, first k live (hnsw_gpu.go:84-123)"""
    gpu_or_skip()
    from longbow_amd import hybrid
    rng = np.random.default_rng(10)
    X = rng.random((500, 32), dtype=F)
    idx = new_index(32, 0)
    idx.Add(None, X)
    q = rng.random(32, dtype=F)
    dead = set(range(0, 500, 3))                 # tombstones
    ids, dist = hybrid.search_hybrid_candidates(idx, q, 10, is_live=lambda i: i not in dead)
    oi, od = oracle.search_batch(0, q[None], X, 100)
    exp = [i for i in oi[0] if i not in dead][:10]
    assert list(ids) == exp
    idx.Close()
