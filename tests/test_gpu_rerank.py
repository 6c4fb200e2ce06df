"""Candidate re-rank through the C ABI: the distance step of processChunkInternal
(internal/store/parallel_search.go:274-364) on rows gathered from the resident corpus / codes, checked
against the oracle's batch-flat restatement on the same gathered rows (simd.EuclideanDistanceBatchFlat,
internal/simd/simd.go:203-229; ADC: simd.go:345-355) and Score = 1/(1+d) (parallel_search.go:355-362)."""
import numpy as np
import pytest

from tests.gpu_util import gpu_or_skip, new_index

pytestmark = pytest.mark.gpu
F = np.float32
FLT_MAX = np.finfo(np.float32).max


def _score(d):
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        return (F(1.0) / (F(1.0) + d.astype(F))).astype(F)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("dim", [768, 100, 7])
def test_rerank_matches_batch_flat_on_gathered_rows(oracle, metric, dim):
    gpu_or_skip()
    rng = np.random.default_rng(dim * 10 + metric)
    n = 6000
    X = rng.random((n, dim), dtype=F)
    q = rng.random(dim, dtype=F)
    idx = new_index(dim, metric)
    idx.Add(None, X)
    for ncand, order in ((1, 1), (50, 1), (513, 0), (4000, 1)):
        rows = rng.integers(0, n, ncand).astype(np.int64)  # duplicates allowed, arbitrary order
        dist, score = idx.Rerank(q, rows, order=order)
        want = oracle.batch_flat(metric, q, X[rows], order)
        assert np.array_equal(dist, want), (ncand, order)
        assert np.array_equal(score, _score(want))
    # the index's own order (order=None -> -1)
    idx.set_order(0)
    rows = np.arange(0, n, 7, dtype=np.int64)
    dist = idx.Rerank(q, rows, order=None, want_score=False)
    assert np.array_equal(dist, oracle.batch_flat(metric, q, X[rows], 0))
    idx.Close()


def test_rerank_invalid_rows_and_empty(oracle):
    gpu_or_skip()
    rng = np.random.default_rng(5)
    X = rng.random((300, 64), dtype=F)
    q = rng.random(64, dtype=F)
    idx = new_index(64, 0)
    # empty index: every row is invalid
    d, s = idx.Rerank(q, [0, 5])
    assert np.all(d == FLT_MAX) and np.all(s == 0)
    idx.Add(None, X)
    rows = np.array([0, -1, 299, 300, 2**40, 17], np.int64)
    d, s = idx.Rerank(q, rows)
    ok = np.array([True, False, True, False, False, True])
    want = oracle.batch_flat(0, q, X[rows[ok]], 1)
    assert np.array_equal(d[ok], want) and np.all(d[~ok] == FLT_MAX)
    assert np.array_equal(s[ok], _score(want)) and np.all(s[~ok] == 0)
    d, s = idx.Rerank(q, np.empty(0, np.int64))
    assert d.size == 0 and s.size == 0
    with pytest.raises(ValueError):
        idx.Rerank(q[:-1], [0])
    idx.Close()


def test_rerank_is_what_search_reports(oracle):
    """re-ranking the labels a search returned reproduces the search's own distances"""
    gpu_or_skip()
    rng = np.random.default_rng(9)
    X = rng.random((20000, 128), dtype=F)
    Q = rng.random((3, 128), dtype=F)
    idx = new_index(128, 0, order=1)
    idx.Add(None, X)
    lab, dist = idx.SearchBatch(Q, 50)
    for b in range(3):
        d, s = idx.Rerank(Q[b], lab[b])
        assert np.array_equal(d, dist[b])
    idx.Close()


def test_pq_rerank(oracle):
    gpu_or_skip()
    from longbow_amd import pq
    rng = np.random.default_rng(21)
    M, dims, n = 96, 768, 9000
    cb = rng.random((M, 256, dims // M), dtype=F)
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    enc.add_codes(codes)
    q = rng.random(dims, dtype=F)
    rows = np.concatenate([rng.integers(0, n, 700), [-3, n, n - 1]]).astype(np.int64)
    d, s = enc.Rerank(q, rows)
    ok = (rows >= 0) & (rows < n)
    want = oracle.adc_batch(oracle.build_adc_table(cb, q), codes[rows[ok]])
    assert np.array_equal(d[ok], want) and np.array_equal(s[ok], _score(want))
    assert np.all(d[~ok] == FLT_MAX) and np.all(s[~ok] == 0)
    enc.Close()
