"""Searches of 1 .. 128 queries over the fp16 image whose candidate launch computes its own thresholds (kernels_gemm_tall16.hip,
TAUIN: the last nq workgroups turn the sample into tau, everybody picks it up in front of its first epilogue): every batch
size around the kernel's tile limits, all three metrics, unfiltered and over a row list, against the oracle (reference
semantics: BruteForceIndex.SearchVectors, internal/store/adaptive_index.go:159-230) and against the same queries searched one
by one.  300k x 256 is large enough for the launch to take that form in the library's default mode (tall16_tin_ok)."""
import numpy as np
import pytest

from tests.gpu_util import assert_same, gpu_or_skip, new_index

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_batches_whose_candidate_launch_computes_its_thresholds(oracle, metric):
    gpu_or_skip()
    rng = np.random.default_rng(77 + metric)
    n, d, k = 300_000, 256, 20
    X = rng.standard_normal((n, d)).astype(F)
    if metric == 0:
        X += F(3.0)  # (a common offset: the centred image)
    Q = np.ascontiguousarray(X[rng.integers(0, n, 128)] + rng.standard_normal((128, d)).astype(F) * F(0.3))
    idx = new_index(d, metric)
    idx.Add(None, X)
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    visible = np.flatnonzero(mask)
    for filtered in (False, True):
        idx.set_filter(mask if filtered else None)
        Xo = X[visible] if filtered else X
        oi, od = oracle.search_batch(metric, Q[:6], Xo, k, nthreads=8)
        if filtered:
            oi = np.where(oi >= 0, visible[np.clip(oi, 0, visible.size - 1)], -1)
        whole = idx.SearchBatch(Q, k)
        assert idx.last_fallbacks == 0, (metric, filtered, idx.last_fallbacks)
        assert idx.fused_giveups == 0
        assert_same(whole[0][:6], whole[1][:6], oi, od, f"metric {metric} filtered {filtered} nq 128 route {idx.last_route}")
        for nq in (1, 7, 8, 9, 33, 64, 65, 127):
            lab, dist = idx.SearchBatch(Q[:nq], k)
            assert_same(lab, dist, whole[0][:nq], whole[1][:nq], f"metric {metric} filtered {filtered} nq {nq} route {idx.last_route}")
            assert idx.last_fallbacks == 0, (metric, filtered, nq, idx.last_fallbacks)
    idx.Close()
