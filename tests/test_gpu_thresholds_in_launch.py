"""Searches of 1 .. 128 queries over the fp16 image whose candidate launch computes its own thresholds (kernels_gemm_tall16.hip,
TAUIN: the first nq workgroups turn the sample into tau, everybody picks it up in front of its first epilogue): every batch
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


def test_searches_side_by_side_get_their_thresholds_without_giving_up(oracle):
    """Four callers with combining off: their candidate launches (one persistent workgroup per CU each) share the GPU, so a
    launch may be only partly resident while it waits for its thresholds.  The duty workgroups are the launch's FIRST ones --
    dispatched before any other -- so nobody waits for a workgroup that cannot start: no give-ups, and every list is the one
    the query gets alone."""
    import threading
    gpu_or_skip()
    rng = np.random.default_rng(5)
    n, d, k = 300_000, 256, 10
    X = rng.standard_normal((n, d)).astype(F)
    Q = np.ascontiguousarray(X[rng.integers(0, n, 32)] + rng.standard_normal((32, d)).astype(F) * F(0.3))
    idx = new_index(d, 1)
    idx.Add(None, X)
    want = [idx.Search(Q[i], k) for i in range(32)]
    oi, od = oracle.search_batch(1, Q[:4], X, k, nthreads=8)
    for i in range(4):
        assert_same(want[i][0], want[i][1], oi[i], od[i], f"query {i} alone vs oracle")
    idx.set_search_combining(False)
    errors = []

    def caller(t):
        try:
            for rep in range(40):
                for i in range(t, 32, 4):
                    lab, dist = idx.Search(Q[i], k)
                    if not (np.array_equal(lab, want[i][0]) and np.array_equal(dist, want[i][1])):
                        errors.append(f"thread {t} query {i} rep {rep}: differs from the search on its own")
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {t}: {type(e).__name__}: {e}")

    ths = [threading.Thread(target=caller, args=(t,)) for t in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors[:5]
    assert idx.fused_giveups == 0, idx.fused_giveups
    idx.Close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_batches_that_end_65_to_128_queries_into_a_256_query_tile(oracle, metric):
    """such a batch runs its whole 256-query tiles on the 256-wide kernel and the rest on the 128-query one (a second pass over
    the image; kernels_gemm_tall16.hip, tall16_window): the lists must be the ones the queries get in any other grouping"""
    gpu_or_skip()
    rng = np.random.default_rng(300 + metric)
    n, d, k = 300_000, 128, 10
    X = rng.standard_normal((n, d)).astype(F)
    Q = np.ascontiguousarray(X[rng.integers(0, n, 384)] + rng.standard_normal((384, d)).astype(F) * F(0.3))
    idx = new_index(d, metric)
    idx.Add(None, X)
    whole = idx.SearchBatch(Q, k)                      # 256 + 128
    assert idx.last_fallbacks == 0
    oi, od = oracle.search_batch(metric, Q[250:262], X, k, nthreads=8)
    assert_same(whole[0][250:262], whole[1][250:262], oi, od, f"metric {metric} nq 384 route {idx.last_route}")
    for nq in (321, 352, 383):                         # 256 + 65 / 96 / 127
        lab, dist = idx.SearchBatch(Q[:nq], k)
        assert_same(lab, dist, whole[0][:nq], whole[1][:nq], f"metric {metric} nq {nq} route {idx.last_route}")
    a = idx.SearchBatch(Q[:256], k)
    b = idx.SearchBatch(Q[256:], k)
    assert_same(np.concatenate([a[0], b[0]]), np.concatenate([a[1], b[1]]), whole[0], whole[1], f"metric {metric} 256 | 128")
    idx.Close()


def test_a_wait_for_the_thresholds_that_gives_up_is_reported_and_the_batch_redone(oracle, monkeypatch):
    """diagnostic build, LB_F16_ABL=12: the duty workgroups never publish their thresholds, every wave's bounded wait (~1 ms)
    gives up, admits nothing and says so through the pinned word; the host redoes the batch on the exact path -- the lists
    are still the oracle's, and the give-up is counted."""
    from tests.gpu_util import diag_lib
    lib = diag_lib()
    rng = np.random.default_rng(12)
    n, d, k = 300_000, 256, 10
    X = rng.standard_normal((n, d)).astype(F)
    Q = np.ascontiguousarray(X[rng.integers(0, n, 8)] + rng.standard_normal((8, d)).astype(F) * F(0.3))
    idx = new_index(d, 1, lib=lib)
    idx.Add(None, X)
    oi, od = oracle.search_batch(1, Q, X, k, nthreads=8)
    lab, dist = idx.SearchBatch(Q[:6], k)
    assert_same(lab, dist, oi[:6], od[:6], "before")
    assert idx.fused_giveups == 0 and idx.last_fallbacks == 0
    monkeypatch.setenv("LB_F16_ABL", "12")
    lab, dist = idx.SearchBatch(Q[:6], k)
    monkeypatch.delenv("LB_F16_ABL")
    assert_same(lab, dist, oi[:6], od[:6], "wait gave up")
    assert idx.fused_giveups == 1 and idx.last_fallbacks == 6, (idx.fused_giveups, idx.last_fallbacks)
    lab, dist = idx.SearchBatch(Q, k)
    assert_same(lab, dist, oi, od, "after")
    assert idx.fused_giveups == 1 and idx.last_fallbacks == 0
    idx.Close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_image_less_corpora_of_16k_to_64k_rows_run_separate_sample_launches(oracle, metric):
    """Since round 4 corpora from 16,384 rows take the sampled threshold.  An index WITHOUT its fp16 image (switched off
    here; also norms outside fp16's range, or memory pressure) runs the split tiles there, and the fused form of their
    sample launch is kept for 131,072 rows and more: on these sizes its in-launch waits gave up (17k rows at 16 queries, 40k
    at 32) and the batch was redone exactly -- right answers, seven times the time.  Every batch size around the tile limits:
    the oracle's lists, no give-up, nothing left to the exact scan."""
    gpu_or_skip()
    rng = np.random.default_rng(1234 + metric)
    d, k = 64, 20
    for n in (17000, 40000, 60000):
        X = rng.standard_normal((n, d)).astype(F)
        Q = rng.standard_normal((64, d)).astype(F)
        idx = new_index(d, metric)
        idx.set_f16_image(0)
        idx.Add(None, X)
        assert idx.f16_image_bytes == 0
        for nq in (5, 8, 16, 32, 33, 64):
            lab, dist = idx.SearchBatch(Q[:nq], k)
            oi, od = oracle.search_batch(metric, Q[:nq], X, k, nthreads=8)
            assert_same(lab, dist, oi, od, f"metric={metric} n={n} nq={nq}")
            assert idx.last_fallbacks == 0, (n, nq)
        assert idx.fused_giveups == 0, n
        idx.Close()


@pytest.mark.parametrize("image", [1, 0])
def test_long_result_lists_on_mid_size_corpora(oracle, image):
    """k = 300 (1024 candidates, thresholds of rank ~48) on 50k and 200k rows, the cells of tools/probe/small_corpus_grid.py
    that fell back or gave up before the gates of round 4: the fused launch (waits gave up at 200k rows, 32 queries, every
    search), the wave-per-row sample in front of the split tiles (L2, 8 queries over 50k rows: all flagged), the fp16
    routes when one sampled span cannot cover the view.  Oracle lists, nothing left to the scan, no give-up."""
    gpu_or_skip()
    rng = np.random.default_rng(4321)
    d, k = 128, 300
    for metric, n in ((0, 50000), (1, 200000), (0, 200000)):
        X = rng.random((n, d), dtype=F)
        Q = rng.random((32, d), dtype=F)
        idx = new_index(d, metric)
        if not image:
            idx.set_f16_image(0)
        idx.Add(None, X)
        for nq in (8, 32):
            lab, dist = idx.SearchBatch(Q[:nq], k)
            oi, od = oracle.search_batch(metric, Q[:nq], X, k, nthreads=8)
            assert_same(lab, dist, oi, od, f"metric={metric} n={n} nq={nq} image={image}")
            assert idx.last_fallbacks == 0, (metric, n, nq)
        assert idx.fused_giveups == 0, (metric, n)
        idx.Close()


def test_k_300_on_a_large_corpus_takes_one_sampled_span(oracle):
    """k = 300 over 600k rows on the library default: 16,384-entry lists (from 1,024 candidates), one sampled span, the fp16
    image -- oracle lists for a batch and for a single query, nothing left to the exact scan."""
    gpu_or_skip()
    rng = np.random.default_rng(99)
    n, d, k = 600_000, 32, 300
    X = rng.random((n, d), dtype=F)
    Q = rng.random((40, d), dtype=F)
    idx = new_index(d, 1)
    idx.Add(None, X)
    assert idx.f16_image_bytes > 0
    for nq in (1, 40):
        lab, dist = idx.SearchBatch(Q[:nq], k)
        oi, od = oracle.search_batch(1, Q[:nq], X, k, nthreads=8)
        assert_same(lab, dist, oi, od, f"k=300 nq={nq}")
        assert idx.last_fallbacks == 0
    idx.Close()
