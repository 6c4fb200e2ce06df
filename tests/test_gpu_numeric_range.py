"""Numeric range of the candidate contractions (f32 MFMA, split-bf16, one fp16 product): corpora whose row norms span many
binades, a few rows far longer than the rest, sparse rows, a large common offset.  Every candidate mode must return the strict
mode's lists bit for bit, and the strict mode the oracle's (reference semantics: internal/simd/simd.go:131-163,365-479 through
BruteForceIndex.SearchVectors, internal/store/adaptive_index.go:159-230).  The long sweep is tools/probe/fuzz_scales.py."""
import numpy as np
import pytest

from tests.gpu_util import assert_same, gpu_or_skip, new_index

pytestmark = pytest.mark.gpu
F = np.float32


def _corpus(kind, n, d, rng):
    base = rng.standard_normal((n, d)).astype(F)
    if kind == "row_binades":
        X = base * np.exp2(rng.integers(-10, 11, (n, 1))).astype(F)
    elif kind == "dim_binades":
        X = base * np.exp2(rng.integers(-6, 7, (1, d))).astype(F)
    elif kind == "few_long_rows":
        X = base * F(2.0 ** -6)
        X[rng.integers(0, n, 3)] *= F(2.0 ** 16)
    elif kind == "sparse":
        X = base * (rng.random((n, d)) < 0.05).astype(F)
    elif kind == "offset":
        X = base * F(0.01) + F(100.0)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(X, dtype=F)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("kind", ["row_binades", "dim_binades", "few_long_rows", "sparse", "offset"])
def test_every_candidate_mode_equals_the_strict_mode_and_the_oracle(oracle, metric, kind):
    gpu_or_skip()
    rng = np.random.default_rng(1000 * metric + sum(kind.encode()))
    n, d, k = 270_000, 64, 10
    X = _corpus(kind, n, d, rng)
    rows = rng.integers(0, n, 130)
    Q = np.ascontiguousarray(X[rows] + rng.standard_normal((130, d)).astype(F) * F(0.02) * np.abs(X[rows]).mean(1, keepdims=True).astype(F))
    idx = new_index(d, metric)
    idx.Add(None, X)
    idx.set_candidate_mode(0)
    want = {nq: idx.SearchBatch(Q[:nq], k) for nq in (4, 64, 130)}
    oi, od = oracle.search_batch(metric, Q[:4], X, k, nthreads=8)
    assert_same(want[4][0], want[4][1], oi, od, f"strict mode vs oracle, {kind}")
    for mode in (3, 4, 2, 1):
        idx.set_candidate_mode(mode)
        for nq in (4, 64, 130):
            lab, dist = idx.SearchBatch(Q[:nq], k)
            assert_same(lab, dist, want[nq][0], want[nq][1], f"mode {mode} nq {nq} {kind} route {idx.last_route}")
    idx.Close()


def test_l2_fast_routes_survive_row_norms_over_twenty_binades(oracle):
    """the property the L2 proof's norm limit buys (kernels_select.hip rerank_finish): no query of these batches is left to the
    exact scan although the corpus' largest norm is a million times the typical one"""
    gpu_or_skip()
    rng = np.random.default_rng(11)
    n, d, k = 270_000, 64, 10
    X = _corpus("row_binades", n, d, rng)
    rows = rng.integers(0, n, 64)
    Q = np.ascontiguousarray(X[rows] * F(1.01))
    idx = new_index(d, 0)
    idx.Add(None, X)
    lab, dist = idx.SearchBatch(Q, k)
    oi, od = oracle.search_batch(0, Q[:6], X, k, nthreads=8)
    assert_same(lab[:6], dist[:6], oi, od, "row binades")
    assert idx.last_fallbacks <= 2, idx.last_fallbacks
    idx.Close()


@pytest.mark.parametrize("offset", [100.0, 1000.0])
def test_l2_with_a_large_common_offset_stays_on_the_matrix_cores(oracle, offset):
    """round-3 verdict, item 4: L2 on data whose rows share a large offset (|mean| >> spread).  The plain candidate key
    |x|^2 - 2 q.x cancels -- the strict mode leaves every such query to the exact scan -- but L2 does not move when both sides
    are shifted: the index's fp16 image holds x - c (c = the column means), the queries' image q - c, and the proof runs on the
    centred norms.  The default mode must answer with the oracle's lists and leave (almost) nothing to the scan."""
    gpu_or_skip()
    rng = np.random.default_rng(int(offset))
    n, d, k = 300_000, 64, 10
    X = np.ascontiguousarray(rng.standard_normal((n, d)).astype(F) * F(0.01) + F(offset))
    rows = rng.integers(0, n, 300)
    Q = np.ascontiguousarray(X[rows] + rng.standard_normal((300, d)).astype(F) * F(0.002))
    idx = new_index(d, 0)
    idx.Add(None, X)
    assert idx.f16_image_bytes > 0
    oi, od = oracle.search_batch(0, Q[:8], X, k, nthreads=8)
    for nq in (1, 8, 40, 300):
        lab, dist = idx.SearchBatch(Q[:nq], k)
        m = min(nq, 8)
        assert_same(lab[:m], dist[:m], oi[:m], od[:m], f"offset {offset} nq {nq} route {idx.last_route}")
        assert idx.last_fallbacks <= max(1, nq // 100), (offset, nq, idx.last_fallbacks)
    # appended rows are shifted by the same centre
    extra = np.ascontiguousarray(rng.standard_normal((5000, d)).astype(F) * F(0.01) + F(offset))
    idx.Add(None, extra)
    X2 = np.concatenate([X, extra])
    lab, dist = idx.SearchBatch(Q[:40], k)
    oi2, od2 = oracle.search_batch(0, Q[:8], X2, k, nthreads=8)
    assert_same(lab[:8], dist[:8], oi2, od2, f"offset {offset} after an append")
    assert idx.last_fallbacks == 0
    idx.Close()


def test_dot_product_with_a_few_very_long_rows_stays_on_the_matrix_cores(oracle):
    """round-3 verdict, item 4: a few rows hundreds of times longer than the rest.  Their candidate products are uncertain by
    gamma_a |q||x| -- with plain keys that uncertainty widens EVERY query's proof (the strict mode leaves all of these to the
    exact scan).  The persistent fp16 kernels use the lower-bound key -(q.x)~ / G - |x| instead: the long rows sort to the front
    of the lists and are scored exactly, and nothing is left to the scan."""
    gpu_or_skip()
    rng = np.random.default_rng(4242)
    n, d, k = 300_000, 64, 10
    X = rng.standard_normal((n, d)).astype(F)
    long_rows = rng.integers(0, n, 3)
    X[long_rows] *= F(256.0)
    X = np.ascontiguousarray(X)
    Q = np.ascontiguousarray(rng.standard_normal((300, d)).astype(F))
    idx = new_index(d, 2)
    idx.Add(None, X)
    assert idx.f16_image_bytes > 0
    oi, od = oracle.search_batch(2, Q[:8], X, k, nthreads=8)
    assert np.isin(long_rows, oi).any()          # (the long rows do matter to the answers)
    for nq in (1, 8, 40, 300):
        lab, dist = idx.SearchBatch(Q[:nq], k)
        m = min(nq, 8)
        assert_same(lab[:m], dist[:m], oi[:m], od[:m], f"long rows nq {nq} route {idx.last_route}")
        if idx.last_route[0] in (6, 7):          # (the persistent fp16 kernels: nq >= 5 here, or the cost model's choice below)
            assert idx.last_fallbacks <= max(1, nq // 100), (nq, idx.last_fallbacks)
    idx.set_candidate_mode(4)                     # LB_CAND_F16: every batch size on those kernels
    for nq in (1, 40, 300):
        lab, dist = idx.SearchBatch(Q[:nq], k)
        m = min(nq, 8)
        assert_same(lab[:m], dist[:m], oi[:m], od[:m], f"long rows, forced fp16, nq {nq}")
        assert idx.last_fallbacks <= max(1, nq // 100), (nq, idx.last_fallbacks)
    idx.Close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_rounding_errors_that_all_point_the_same_way(oracle, metric):
    """Every element of the corpus and of the queries sits just above a midpoint between two fp16 values, all positive: the image
    rounds every element UP by half an ulp, so the candidate keys' errors add coherently instead of cancelling -- the worst case
    for an error bound taken from the measured residual NORMS (search_batch_device: gamma(q) = ... rho_x ... rho_q).  The lists
    must still be the oracle's; a bound that were only statistical would lose neighbours here."""
    gpu_or_skip()
    rng = np.random.default_rng(400 + metric)
    n, d, k = 270_000, 64, 10

    def above_midpoints(shape):
        m = rng.integers(1024, 2048, shape).astype(np.float64)          # 11-bit significands of fp16 values in [1, 2)
        return ((m + 0.5 + 1.0 / 64.0) * 2.0 ** -10).astype(F)            # just above the midpoint to the next one

    X = np.ascontiguousarray(above_midpoints((n, d)))
    Q = np.ascontiguousarray(above_midpoints((64, d)))
    assert np.all(X.astype(np.float16).astype(F) > X)                      # (the image rounds every element up)
    idx = new_index(d, metric)
    idx.Add(None, X)
    oi, od = oracle.search_batch(metric, Q[:8], X, k, nthreads=8)
    for mode in (3, 4):
        idx.set_candidate_mode(mode)
        for nq in (4, 8, 64):
            lab, dist = idx.SearchBatch(Q[:nq], k)
            m8 = min(nq, 8)
            assert_same(lab[:m8], dist[:m8], oi[:m8], od[:m8], f"metric {metric} mode {mode} nq {nq} route {idx.last_route}")
    idx.Close()
