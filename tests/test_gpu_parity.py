"""Parity of the HIP k-NN path with the CPU oracle on seeded inputs: index sets, order and
distances BIT-EXACT (the north star asks for exact index sets and 1e-5 relative distances;
the re-rank / scan kernels reproduce the oracle's f32 operation order, so equality holds)."""
import threading

import numpy as np
import pytest

from tests.gpu_util import assert_same, diag_lib, gpu_or_skip, new_index

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("n,d,nq,k", [
    (1, 8, 1, 1), (37, 3, 2, 5), (1000, 7, 1, 10), (5000, 33, 4, 10), (5000, 64, 8, 100),
    (20000, 128, 17, 10), (20000, 128, 64, 10), (30000, 100, 130, 37), (70000, 96, 40, 100),
    (4097, 16, 33, 3), (65537, 8, 20, 16),
])
def test_search_matches_oracle(oracle, metric, order, n, d, nq, k):
    gpu_or_skip()
    rng = np.random.default_rng(n * 31 + d * 7 + nq + metric)
    X = rng.random((n, d), dtype=F) - F(0.25)
    Q = rng.random((nq, d), dtype=F) - F(0.25)
    idx = new_index(d, metric, order)
    idx.Add(None, X)
    lab, dist = idx.SearchBatch(Q, k)
    oi, od = oracle.search_batch(metric, Q, X, k, order=order, nthreads=8)
    assert_same(lab, dist, oi, od, f"metric={metric} order={order} n={n} d={d} nq={nq} k={k}")
    idx.Close()


def test_scan_and_batched_paths_agree(oracle):
    """the same queries through the exact scan path (nq small) and the MFMA path (nq large)"""
    gpu_or_skip()
    rng = np.random.default_rng(99)
    X = rng.random((50000, 256), dtype=F)
    Q = rng.random((96, 256), dtype=F)
    for metric in (0, 1, 2):
        idx = new_index(256, metric)
        idx.Add(None, X)
        lab_b, dist_b = idx.SearchBatch(Q, 50)
        assert idx.last_fallbacks == 0
        for s in range(0, 96, 8):
            lab_s, dist_s = idx.SearchBatch(Q[s:s + 8], 50)
            assert np.array_equal(lab_s, lab_b[s:s + 8]) and np.array_equal(dist_s, dist_b[s:s + 8])
        idx.Close()


def test_append_ids_and_growth(oracle):
    """Add APPENDS; ids are reported when given (SURVEY 8b ID semantics); mixed id/no-id adds"""
    gpu_or_skip()
    rng = np.random.default_rng(5)
    d = 48
    parts = [rng.random((n, d), dtype=F) for n in (10, 1500, 1, 4000, 700)]
    X = np.concatenate(parts)
    ids = np.arange(len(X), dtype=np.int64) * 5 + 11
    Q = rng.random((20, d), dtype=F)
    idx = new_index(d, 0)
    pos = 0
    for p in parts:
        idx.Add(ids[pos:pos + len(p)], p)
        pos += len(p)
    assert idx.ntotal == len(X)
    lab, dist = idx.SearchBatch(Q, 10)
    oi, od = oracle.search_batch(0, Q, X, 10, ids=ids, nthreads=4)
    assert_same(lab, dist, oi, od)
    idx.Close()
    # no ids first, ids later: earlier rows report their positions
    idx = new_index(d, 2)
    idx.Add(None, parts[1])
    idx.Add(np.arange(len(parts[3]), dtype=np.int64) + 10**12, parts[3])
    X2 = np.concatenate([parts[1], parts[3]])
    ids2 = np.concatenate([np.arange(len(parts[1])), np.arange(len(parts[3])) + 10**12]).astype(np.int64)
    lab, dist = idx.SearchBatch(Q[:3], 7)
    oi, od = oracle.search_batch(2, Q[:3], X2, 7, ids=ids2)
    assert_same(lab, dist, oi, od)
    idx.Close()


def test_k_larger_than_n_and_empty(oracle):
    gpu_or_skip()
    rng = np.random.default_rng(8)
    X = rng.random((5, 16), dtype=F)
    Q = rng.random((30, 16), dtype=F)
    idx = new_index(16, 1)
    lab, dist = idx.SearchBatch(Q, 4)            # empty index
    assert np.all(lab == -1) and np.all(dist == np.finfo(F).max)
    idx.Add(None, X)
    for nq in (1, 30):
        lab, dist = idx.SearchBatch(Q[:nq], 9)   # k > n: n hits then -1 / FLT_MAX padding
        oi, od = oracle.search_batch(1, Q[:nq], X, 9)
        assert_same(lab, dist, oi, od)
        assert np.all(lab[:, 5:] == -1)
    idx.Close()


def test_ties_lowest_position_wins(oracle):
    """duplicate rows and an all-zero query: canonical (distance, position) order"""
    gpu_or_skip()
    rng = np.random.default_rng(3)
    base = rng.random((50, 32), dtype=F)
    X = np.concatenate([base] * 40)              # every row appears 40 times
    Q = np.concatenate([base[:20] + F(0.001), np.zeros((12, 32), F)])
    for metric in (0, 1, 2):
        idx = new_index(32, metric)
        idx.Add(None, X)
        for nq in (4, 32):
            lab, dist = idx.SearchBatch(Q[:nq] if nq == 4 else Q, 25)
            oi, od = oracle.search_batch(metric, Q[:nq] if nq == 4 else Q, X, 25, nthreads=4)
            assert_same(lab, dist, oi, od, f"metric={metric} nq={nq}")
        idx.Close()


def test_zero_vectors_cosine(oracle):
    """zero query / zero rows -> distance exactly 1.0 (simd_test.go:183-194)"""
    gpu_or_skip()
    rng = np.random.default_rng(4)
    X = rng.random((3000, 24), dtype=F)
    X[::7] = 0
    Q = rng.random((40, 24), dtype=F)
    Q[3] = 0
    idx = new_index(24, 1)
    idx.Add(None, X)
    lab, dist = idx.SearchBatch(Q, 12)
    oi, od = oracle.search_batch(1, Q, X, 12, nthreads=4)
    assert_same(lab, dist, oi, od)
    assert np.all(dist[3] == 1.0)
    idx.Close()


def test_filter_mask(oracle):
    """metadata predicate mask (byte per row, 0 = excluded): SURVEY f-3"""
    gpu_or_skip()
    rng = np.random.default_rng(12)
    X = rng.random((20000, 64), dtype=F)
    Q = rng.random((48, 64), dtype=F)
    meta = rng.integers(0, 100, 20000)
    mask = (meta < 10).astype(np.uint8)          # 10 % selectivity, as config 5
    for metric in (0, 2):
        idx = new_index(64, metric)
        idx.Add(None, X)
        idx.set_filter(mask)
        for qs in (Q[:5], Q):
            lab, dist = idx.SearchBatch(qs, 20)
            oi, od = oracle.search_batch(metric, qs, X, 20, mask=mask, nthreads=4)
            assert_same(lab, dist, oi, od)
            assert np.all(mask[lab] == 1)
        idx.set_filter(None)
        lab, dist = idx.SearchBatch(Q[:5], 20)
        oi, od = oracle.search_batch(metric, Q[:5], X, 20)
        assert_same(lab, dist, oi, od)
        idx.Close()


def test_selective_filter_walks_compacted_row_list(oracle):
    """a filter hiding >= 5 % of the corpus switches every search path (exact scan, narrow and wide
    MFMA tiles, split-bf16 candidates) to the compacted visible-row list; results stay bit-exact,
    including ids, ragged tile tails, fewer-than-k and zero visible rows, and appends after the filter"""
    gpu_or_skip()
    rng = np.random.default_rng(77)
    n, d = 70001, 64
    X = rng.random((n, d), dtype=F)
    X[5000:5040] = X[100]                        # duplicate rows: ties must still break by row position
    Q = rng.random((400, d), dtype=F)
    Q[7] = X[100]
    ids = (np.arange(n, dtype=np.int64) * 3 + 11)
    meta = rng.integers(0, 1000, n)
    meta[[100, 5003, 5017]] = 0
    for metric in (0, 1, 2):
        idx = new_index(d, metric)
        idx.Add(ids, X)
        for sel in (980, 500, 100, 7, 1):        # 98 % (per-row mask test), 50 %, 10 %, 0.7 %, ~0.1 % visible
            mask = (meta < sel).astype(np.uint8)
            idx.set_filter(mask)
            for qs in (Q[:1], Q[:3], Q[:8], Q[:40], Q):
                lab, dist = idx.SearchBatch(qs, 100)
                oi, od = oracle.search_batch(metric, qs, X, 100, mask=mask, ids=ids, nthreads=8)
                assert_same(lab, dist, oi, od, f"metric {metric} sel {sel} nq {len(qs)}")
        # split-bf16 candidate generation walks the same list
        mask = (meta < 100).astype(np.uint8)
        idx.set_filter(mask)
        idx.set_candidate_mode(1)
        lab, dist = idx.SearchBatch(Q, 100)
        oi, od = oracle.search_batch(metric, Q, X, 100, mask=mask, ids=ids, nthreads=8)
        assert_same(lab, dist, oi, od, f"split metric {metric}")
        idx.set_candidate_mode(0)
        # rows appended after the filter are visible (and join the list)
        X2 = rng.random((300, d), dtype=F)
        ids2 = np.arange(300, dtype=np.int64) + 10_000_000
        idx.Add(ids2, X2)
        Xa, ida = np.concatenate([X, X2]), np.concatenate([ids, ids2])
        ma = np.concatenate([mask, np.ones(300, np.uint8)])
        for qs in (Q[:2], Q[:40]):
            lab, dist = idx.SearchBatch(qs, 100)
            oi, od = oracle.search_batch(metric, qs, Xa, 100, mask=ma, ids=ida, nthreads=8)
            assert_same(lab, dist, oi, od, f"after append metric {metric}")
        # nothing visible -> every slot padded
        idx.set_filter(np.zeros(n + 300, np.uint8))
        for qs in (Q[:2], Q[:40]):
            lab, dist = idx.SearchBatch(qs, 10)
            assert np.all(lab == -1) and np.all(dist == np.finfo(F).max)
        # a column predicate evaluated on the device lands on the same path
        col = np.concatenate([meta, np.full(300, 999)]).astype(np.int64)
        idx.filter_column(col, "<", 50)
        mc = (col < 50).astype(np.uint8)
        lab, dist = idx.SearchBatch(Q[:40], 100)
        oi, od = oracle.search_batch(metric, Q[:40], Xa, 100, mask=mc, ids=ida, nthreads=8)
        assert_same(lab, dist, oi, od, f"column predicate metric {metric}")
        idx.Close()


def test_sampled_threshold_first_pass(oracle):
    """corpora >= 16k rows (64k before round 4) take a strided row sample's m-th best entry as admission threshold and walk
    the rows once; the classic bootstrap schedule is the fallback.  Both must equal the oracle on random,
    sorted (worst->best and best->worst), and heavily duplicated corpora, on every search path."""
    lib = diag_lib()  # lb_debug_set_sample_tau exists only in the diagnostic build
    rng = np.random.default_rng(2024)
    n, d = 150_000, 32
    base = rng.random((n, d), dtype=F)
    q = rng.random(d, dtype=F)
    dist = oracle.batch_flat(0, q, base)
    corpora = {
        "random": base,
        "worst_first": base[np.argsort(-dist)],
        "best_first": base[np.argsort(dist)],
        "duplicates": base[rng.integers(0, 40, n)],   # 40 distinct vectors: thresholds tie massively
    }
    Q = np.concatenate([np.stack([q + F(1e-3) * rng.random(d, dtype=F) for _ in range(8)]),
                        rng.random((192, d), dtype=F)])
    try:
        for name, X in corpora.items():
            for metric in (0, 1, 2):
                idx = new_index(d, metric, lib=lib)
                idx.Add(None, X)
                want = {}
                for nq in (1, 3, 8, 40, 200):
                    want[nq] = oracle.search_batch(metric, Q[:nq], X, 20, nthreads=8)
                for sampled in (1, 0):
                    lib.lb_debug_set_sample_tau(sampled)
                    for nq in (1, 3, 8, 40, 200):
                        lab, dd = idx.SearchBatch(Q[:nq], 20)
                        assert_same(lab, dd, *want[nq], f"{name} metric {metric} nq {nq} sampled {sampled}")
                        if name == "random" and sampled:
                            assert idx.last_fallbacks == 0
                idx.Close()
    finally:
        lib.lb_debug_set_sample_tau(1)


def test_bootstrap_chunk_entirely_hidden(oracle):
    """a predicate that hides a contiguous prefix (e.g. `timestamp > T` on time-ordered rows) leaves the
    bootstrap chunk without a single visible row: the "no threshold yet" state must then admit every row
    of the next chunk (regression: the MFMA kernels decoded it as NaN and admitted nothing)"""
    gpu_or_skip()
    rng = np.random.default_rng(99)
    n, d = 60000, 64                              # below the sampled-threshold limit: classic schedule
    X = rng.random((n, d), dtype=F)
    Q = rng.random((300, d), dtype=F)
    mask = np.ones(n, np.uint8)
    mask[:2500] = 0                               # 95.8 % visible: the per-row mask test, not the compacted list
    for metric in (0, 1, 2):
        idx = new_index(d, metric)
        idx.Add(None, X)
        idx.set_filter(mask)
        for nq in (1, 6, 16, 300):
            lab, dist = idx.SearchBatch(Q[:nq], 50)
            oi, od = oracle.search_batch(metric, Q[:nq], X, 50, mask=mask, nthreads=8)
            assert_same(lab, dist, oi, od, f"metric {metric} nq {nq}")
        idx.Close()


def test_non_finite_rows_and_queries_rank_canonically(oracle):
    """inf / NaN components: distances that are NaN rank after +inf, ties by row position, on every batch
    size (an index holding non-finite rows always takes the exact scan path; a NaN query falls back to it)"""
    gpu_or_skip()
    rng = np.random.default_rng(404)
    n, d = 70_000, 16
    X = rng.random((n, d), dtype=F)
    X[5] = np.inf
    X[6] = np.nan
    X[7] = -np.inf
    X[40000, 3] = np.nan
    Q = rng.random((20, d), dtype=F)
    Q[3, 0] = np.nan
    Q[11] = np.inf
    clean = rng.random((n, d), dtype=F)
    for metric in (0, 1, 2):
        for corpus in (X, clean):
            idx = new_index(d, metric)
            idx.Add(None, corpus)
            for nq in (1, 4, 20):
                lab, dist = idx.SearchBatch(Q[:nq], 10)
                oi, od = oracle.search_batch(metric, Q[:nq], corpus, 10, nthreads=4)
                assert np.array_equal(lab, oi), f"metric {metric} nq {nq}: {np.argwhere(lab != oi)[:4]}"
                assert np.array_equal(dist, od, equal_nan=True), f"metric {metric} nq {nq}"
            idx.Close()


def test_adversarial_order_forces_list_overflow(oracle):
    """rows sorted from worst to best: every row is admitted, the candidate lists overflow, and the
    library must fall back to overflow-proof chunking and still be exact.  (The classic bootstrap schedule is forced
    through the diagnostic build's hook: since round 4 a corpus of this size takes the sampled threshold, whose strided
    sample does not care about the row order -- the second half of the test checks exactly that.)"""
    lib = diag_lib()
    rng = np.random.default_rng(21)
    d = 32
    q = rng.random(d, dtype=F)
    X = rng.random((60000, d), dtype=F)
    dist = oracle.batch_flat(0, q, X)
    X = X[np.argsort(-dist)]                     # descending distance to q
    Q = np.stack([q + F(1e-3) * rng.random(d, dtype=F) for _ in range(24)])
    idx = new_index(d, 0, lib=lib)
    idx.Add(None, X)
    lib.lb_debug_set_sample_tau(0)                # classic schedule: bootstrap chunk, then growing chunks
    try:
        for qs in (Q[:2], Q):
            lab, dd = idx.SearchBatch(qs, 10)
            oi, od = oracle.search_batch(0, qs, X, 10, nthreads=4)
            assert_same(lab, dd, oi, od)
        assert idx.last_fallbacks > 0             # the batched path did have to fall back
    finally:
        lib.lb_debug_set_sample_tau(1)
    idx.Close()
    idx = new_index(d, 0, lib=lib)                # (a fresh index: the one above remembers the widened lists of its hard batches)
    idx.Add(None, X)
    for qs in (Q[:2], Q):                         # sampled threshold: the order does not matter, nothing falls back
        lab, dd = idx.SearchBatch(qs, 10)
        oi, od = oracle.search_batch(0, qs, X, 10, nthreads=4)
        assert_same(lab, dd, oi, od)
        assert idx.last_fallbacks == 0
    idx.Close()


def test_near_duplicates_force_containment_fallback(oracle):
    """a corpus of near-identical rows: approximate keys cannot separate rank k from rank kc, the
    containment check must fail and the exact scan must take over"""
    gpu_or_skip()
    rng = np.random.default_rng(22)
    d = 128
    c = rng.random(d, dtype=F)
    X = (c[None, :] + F(1e-6) * rng.standard_normal((30000, d)).astype(F)).astype(F)
    Q = rng.random((20, d), dtype=F)
    for metric in (0, 1, 2):
        idx = new_index(d, metric)
        idx.Add(None, X)
        lab, dd = idx.SearchBatch(Q, 10)
        oi, od = oracle.search_batch(metric, Q, X, 10, nthreads=8)
        assert_same(lab, dd, oi, od, f"metric={metric}")
        idx.Close()


def test_concurrent_search_from_many_threads(oracle):
    """Search takes the read lock: many goroutines search one index at once (faiss_gpu.go:108)"""
    gpu_or_skip()
    rng = np.random.default_rng(30)
    X = rng.random((20000, 64), dtype=F)
    idx = new_index(64, 0)
    idx.Add(None, X)
    Qs = [rng.random((nq, 64), dtype=F) for nq in (1, 3, 40, 1, 64, 2, 33, 5)]
    exp = [oracle.search_batch(0, q, X, 10, nthreads=2) for q in Qs]
    out = [None] * len(Qs)
    errs = []

    def work(i):
        try:
            for _ in range(3):
                out[i] = idx.SearchBatch(Qs[i], 10)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(Qs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    for (lab, dd), (oi, od) in zip(out, exp):
        assert_same(lab, dd, oi, od)
    idx.Close()


def test_large_k(oracle):
    gpu_or_skip()
    rng = np.random.default_rng(41)
    X = rng.random((30000, 32), dtype=F)
    Q = rng.random((20, 32), dtype=F)
    idx = new_index(32, 0)
    idx.Add(None, X)
    for k in (1000, 2048):
        for qs in (Q[:2], Q):
            lab, dd = idx.SearchBatch(qs, k)
            oi, od = oracle.search_batch(0, qs, X, k, nthreads=4)
            assert_same(lab, dd, oi, od, f"k={k}")
    from longbow_amd import gpu
    with pytest.raises(gpu.LongbowGPUError):
        idx.SearchBatch(Q[:1], 5000)
    idx.Close()


def test_fill_uniform_matches_oracle(oracle):
    gpu_or_skip()
    torch = pytest.importorskip("torch")
    from longbow_amd import _lib
    lib = _lib.load()
    t = torch.empty(100003, device="cuda")
    assert lib.lb_gpu_fill_uniform_device(0, t.data_ptr(), t.numel(), 12345, 77, None) == 0
    assert np.array_equal(t.cpu().numpy(), oracle.fill_uniform(100003, 12345, 77))
    c = torch.empty(5000, dtype=torch.uint8, device="cuda")
    assert lib.lb_gpu_fill_codes_device(0, c.data_ptr(), c.numel(), 7, 3, None) == 0
    assert np.array_equal(c.cpu().numpy(), oracle.fill_codes(5000, 7, 3))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_split_bf16_candidates_give_identical_results(oracle, metric):
    """candidate mode 1 (3 x bf16 MFMA on the split image) must change nothing but speed: the exact
    re-rank + containment proof (or the scan fallback) make the results equal the oracle bit for bit"""
    gpu_or_skip()
    rng = np.random.default_rng(77 + metric)
    for (n, d, nq, k) in ((30000, 64, 40, 10), (50000, 256, 130, 50), (20000, 96, 33, 100)):
        X = (rng.random((n, d), dtype=F) - F(0.3)) * F(3.0)
        Q = (rng.random((nq, d), dtype=F) - F(0.3)) * F(3.0)
        idx = new_index(d, metric)
        idx.Add(None, X[: n // 2])
        idx.set_candidate_mode(1)          # mirror built for existing rows ...
        idx.Add(None, X[n // 2:])          # ... and kept in sync by later adds
        lab, dist = idx.SearchBatch(Q, k)
        oi, od = oracle.search_batch(metric, Q, X, k, nthreads=8)
        assert_same(lab, dist, oi, od, f"split metric={metric} n={n} d={d}")
        fb = idx.last_fallbacks
        idx.set_candidate_mode(2)          # the same contraction with the operands split in registers (no image)
        lab2, dist2 = idx.SearchBatch(Q, k)
        assert np.array_equal(lab2, lab) and np.array_equal(dist2, dist)
        big = np.concatenate([Q] * (400 // nq + 1))[:400]   # > 384 queries: the 128-query tile in this mode
        labb, distb = idx.SearchBatch(big, k)
        assert np.array_equal(labb[:nq], lab) and np.array_equal(distb[:nq], dist)
        assert np.array_equal(labb[nq:2 * nq], lab[: min(nq, 400 - nq)]) if 2 * nq <= 400 else True
        for mode in (0, 3, 4):             # strict f32 beyond 384 queries; AUTO (the default); one fp16 product
            idx.set_candidate_mode(mode)
            lab0, dist0 = idx.SearchBatch(Q, k)
            assert np.array_equal(lab0, lab) and np.array_equal(dist0, dist)
            lab4, dist4 = idx.SearchBatch(big, k)
            assert np.array_equal(lab4, labb) and np.array_equal(dist4, distb)
        assert fb <= nq // 4
        idx.Close()
    idx = new_index(48, metric)            # dim % 32 != 0 -> unsupported, index keeps working in f32 mode
    from longbow_amd import gpu
    with pytest.raises(gpu.LongbowGPUError):
        idx.set_candidate_mode(1)
    idx.Close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_tall_tile_every_route(oracle, metric):
    """the 256-row tile of the split contraction (kernels_gemm_tall.hip) on each of its routes: default path at 65-384
    queries (corpus split in registers), candidate mode 1 (corpus image, 128- and 256-query tiles) and mode 2, with a
    ragged row count, a ragged last query tile, a predicate mask and a selective filter (row list) -- all equal to the
    oracle bit for bit and to each other"""
    gpu_or_skip()
    rng = np.random.default_rng(500 + metric)
    n, d, k = 41111, 96, 20                       # n % 256 != 0: the last corpus tile is ragged
    X = (rng.random((n, d), dtype=F) - F(0.4)) * F(2.0)
    Q = (rng.random((512, d), dtype=F) - F(0.4)) * F(2.0)
    idx = new_index(d, metric)
    idx.Add(None, X)
    want = {}
    for nq in (70, 200, 384, 512):
        want[nq] = oracle.search_batch(metric, Q[:nq], X, k, nthreads=8)
    for mode in (3, 0, 1, 2, 4):                  # 3 = AUTO, what a fresh index runs; 4 = one fp16 product
        idx.set_candidate_mode(mode)
        for nq in (70, 200, 384, 512):            # 512 in mode 1: the 256-query tile
            lab, dist = idx.SearchBatch(Q[:nq], k)
            assert_same(lab, dist, want[nq][0], want[nq][1], f"tall metric={metric} mode={mode} nq={nq}")
    for frac in (0.97, 0.3):                      # per-row test / compacted row list
        mask = (rng.random(n) < frac).astype(np.uint8)
        idx.set_filter(mask)
        oi, od = oracle.search_batch(metric, Q[:130], X, k, mask=mask, nthreads=8)
        for mode in (3, 0, 1, 4):
            idx.set_candidate_mode(mode)
            for nq in (130, 512):
                lab, dist = idx.SearchBatch(Q[:nq], k)
                assert_same(lab[:130], dist[:130], oi, od, f"tall masked metric={metric} mode={mode} frac={frac} nq={nq}")
    idx.set_filter(None)
    idx.Close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_every_path_boundary(oracle, metric):
    """batch sizes on both sides of every switch of the path selection (exact scan | fused narrow launch | 64-query tile |
    tall tile / multi-pass | f32 tile), on a corpus large enough for the sampled threshold, plain and under a filter"""
    gpu_or_skip()
    rng = np.random.default_rng(4242 + metric)
    n, d, k = 70_001, 64, 12
    X = rng.standard_normal((n, d)).astype(F)
    Q = rng.standard_normal((385, d)).astype(F)
    idx = new_index(d, metric)
    idx.Add(None, X)
    full = oracle.search_batch(metric, Q, X, k, nthreads=8)
    for nq in (1, 4, 5, 8, 9, 16, 17, 32, 33, 64, 65, 128, 129, 192, 193, 384, 385):
        lab, dist = idx.SearchBatch(Q[:nq], k)
        assert_same(lab, dist, full[0][:nq], full[1][:nq], f"boundary metric={metric} nq={nq}")
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    idx.set_filter(mask)
    fm = oracle.search_batch(metric, Q[:129], X, k, mask=mask, nthreads=8)
    for nq in (5, 32, 33, 65, 129):
        lab, dist = idx.SearchBatch(Q[:nq], k)
        assert_same(lab, dist, fm[0][:nq], fm[1][:nq], f"boundary masked metric={metric} nq={nq}")
    idx.Close()


def test_fused_sample_launch_and_its_give_up_path(oracle):
    """5-32 queries on >= 16k rows: the sampled threshold rides inside the candidate launch (sample tiles, per-query
    threshold workgroups, corpus workgroups that pick the thresholds up).  Results equal the oracle; when a wait inside
    the launch gives up (forced here through the host hook) the whole batch is redone on the exact path and the next
    searches run fused again."""
    lib = diag_lib()  # the give-up path is forced through a hook that only the diagnostic build exports
    rng = np.random.default_rng(99)
    n, d, k = 150_000, 64, 25
    X = rng.standard_normal((n, d)).astype(F)
    Q = rng.standard_normal((32, d)).astype(F)
    for metric in (0, 1, 2):
        idx = new_index(d, metric, lib=lib)
        idx.set_f16_image(0)  # (with its fp16 image a corpus of this size takes the one-tile fp16 kernel: this test is about the fused split tile)
        idx.Add(None, X)
        want = {nq: oracle.search_batch(metric, Q[:nq], X, k, nthreads=8) for nq in (5, 9, 32)}
        for rep in range(3):
            for nq in (5, 9, 32):
                lab, dist = idx.SearchBatch(Q[:nq], k)
                assert_same(lab, dist, want[nq][0], want[nq][1], f"fused metric={metric} nq={nq} rep={rep}")
                assert idx.last_fallbacks <= nq // 4
        before = idx.fused_giveups
        lib.lb_debug_fused_fail_next(1)
        lab, dist = idx.SearchBatch(Q[:9], k)
        assert_same(lab, dist, want[9][0], want[9][1], f"fused give-up metric={metric}")
        assert idx.last_fallbacks == 9 and idx.fused_giveups == before + 1
        lab, dist = idx.SearchBatch(Q[:32], k)
        assert_same(lab, dist, want[32][0], want[32][1], f"fused after give-up metric={metric}")
        assert idx.last_fallbacks <= 8
        mask = (rng.random(n) < 0.6).astype(np.uint8)   # compacted row list under the fused launch
        idx.set_filter(mask)
        oi, od = oracle.search_batch(metric, Q[:16], X, k, mask=mask, nthreads=8)
        lab, dist = idx.SearchBatch(Q[:16], k)
        assert_same(lab, dist, oi, od, f"fused masked metric={metric}")
        idx.Close()


def test_growth_survives_a_refused_mapping(oracle):
    """the corpus grows in place through the virtual-memory API; when the driver refuses to extend the mapping
    (forced here) the rows move once into a hipMalloc buffer and the index keeps working, ids and all"""
    lib = diag_lib()
    rng = np.random.default_rng(31)
    d, k = 48, 15
    X = rng.random((9000, d), dtype=F)
    ids = np.arange(9000, dtype=np.int64) * 3 + 7
    Q = rng.random((6, d), dtype=F)
    idx = new_index(d, 0, lib=lib)
    idx.Add(ids[:1000], X[:1000])
    try:
        lib.lb_debug_vmm_fail_next(1)
        idx.reserve(700_000)                      # needs more than what is mapped: refused -> migrate
    finally:
        lib.lb_debug_vmm_fail_next(0)
    idx.Add(ids[1000:5000], X[1000:5000])         # geometric hipMalloc growth from here on
    idx.Add(ids[5000:], X[5000:])
    assert idx.ntotal == 9000
    for qs in (Q[:1], Q):
        lab, dist = idx.SearchBatch(qs, k)
        oi, od = oracle.search_batch(0, qs, X, k, ids=ids)
        assert_same(lab, dist, oi, od, "after the forced migration")
    idx.Close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_fp16_route_on_hostile_data_stays_exact_and_backs_off(oracle, metric):
    """one fp16 product per element is only as good as ~1e-3 of |q||x|: on a corpus of tight clusters the containment proof
    fails for most queries, which are then redone by the exact scan (results stay bit-equal to the oracle), and AUTO stops
    offering the route for a while; norms outside the fp16 range switch it off altogether"""
    gpu_or_skip()
    rng = np.random.default_rng(900 + metric)
    n, d, k = 300_000, 64, 30
    centers = rng.standard_normal((300, d)).astype(F)
    X = (centers[rng.integers(0, 300, n)] + F(2e-4) * rng.standard_normal((n, d)).astype(F)).astype(F)
    Q = (centers[rng.integers(0, 300, 200)] + F(1e-3) * rng.standard_normal((200, d)).astype(F)).astype(F)
    oi, od = oracle.search_batch(metric, Q, X, k, nthreads=8)
    idx = new_index(d, metric)
    idx.Add(None, X)
    idx.set_candidate_mode(4)
    lab, dist = idx.SearchBatch(Q, k)
    assert_same(lab, dist, oi, od, f"fp16 forced, clustered, metric={metric}")
    idx.set_candidate_mode(3)
    for rep in range(4):  # AUTO: the first call may take the fp16 route and fall back, later calls leave it alone
        lab, dist = idx.SearchBatch(Q, k)  # (clusters this tight defeat every approximate key: exact scans either way)
        assert_same(lab, dist, oi, od, f"auto, clustered, metric={metric} rep={rep}")
    idx.Close()
    # a corpus whose norms leave the fp16 range: the route is never offered, results exact
    Xb = (rng.standard_normal((100_000, d)) * 3e4).astype(F)
    Qb = rng.standard_normal((130, d)).astype(F)
    idx = new_index(d, metric)
    idx.Add(None, Xb)
    for mode in (4, 3):
        idx.set_candidate_mode(mode)
        lab, dist = idx.SearchBatch(Qb, k)
        oi, od = oracle.search_batch(metric, Qb, Xb, k, nthreads=8)
        assert_same(lab, dist, oi, od, f"out-of-range norms, mode={mode}, metric={metric}")
    idx.Close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_fp16_persistent_tile_shapes(oracle, metric):
    """the persistent fp16 kernel (kernels_gemm_tall16.hip) over the shapes its tile walk has to get right: one K-step per
    row (D = 32: the side input of the next tile is asked for in the only step there is), fewer corpus tiles than
    workgroups, a ragged last tile, 1 .. 6 query tiles per corpus tile (3, 5 and 6 leave workgroup slots of an XCD idle) and
    more query tiles than an XCD has slots (the one-workgroup-per-tile form takes over), batches that end 1 .. 64 queries into
    a 256-query tile (257, 320, 1050: whole tiles on the 256-wide kernel, the tail on the one-tile kernel; 384, 600: not split) -- equal to the oracle on the
    first queries and to the strict mode on all of them"""
    gpu_or_skip()
    rng = np.random.default_rng(900 + metric)
    k = 10
    saw_fp16 = 0
    for n, d in ((300, 32), (5000, 64), (70001, 96), (33000, 768), (5000, 100), (20000, 300), (3000, 50)):  # (the last three: zero-padded planes)
        X = (rng.random((n, d), dtype=F) - F(0.45)) * F(1.5)
        nq_max = 8300 if n == 5000 else 1500
        Q = (rng.random((nq_max, d), dtype=F) - F(0.45)) * F(1.5)
        idx = new_index(d, metric)
        idx.Add(None, X)
        oi, od = oracle.search_batch(metric, Q[:24], X, k, nthreads=8)
        for nq in (1, 3, 5, 17, 40, 64, 65, 100, 128, 129, 200, 256, 257, 320, 384, 600, 1050, 1500) + ((8300,) if n == 5000 else ()):  # <= 64: the 64-query tile (with the copy)
            idx.set_candidate_mode(0)
            want = idx.SearchBatch(Q[:nq], k)
            idx.set_candidate_mode(4)
            for image in (1, 0):  # from the corpus's fp16 image (four-stage ring) / f32 rows rounded in registers
                idx.set_f16_image(image)
                assert (idx.f16_image_bytes > 0) == bool(image), (image, idx.f16_image_bytes)
                lab, dist = idx.SearchBatch(Q[:nq], k)
                saw_fp16 += idx.last_route[0] in (6, 7)  # (a batch the fp16 bound cannot prove is re-run on the split route)
                assert_same(lab, dist, want[0], want[1], f"fp16 persistent metric={metric} n={n} d={d} nq={nq} image={image}")
                m = min(24, nq)
                assert_same(lab[:m], dist[:m], oi[:m], od[:m], f"fp16 persistent vs oracle metric={metric} n={n} d={d} nq={nq} image={image}")
            idx.set_f16_image(1)
        idx.Close()
    assert saw_fp16 >= 16, saw_fp16


@pytest.mark.parametrize("d", [16, 32, 64])
def test_fp16_one_and_two_k_steps_per_tile_many_tiles(oracle, d):
    """D <= 64 over the fp16 copy: a tile is one or two K-steps, so the side input (norms) of a tile is asked for fewer requests
    before its use than the six-stage ring keeps in flight -- the kernel waits for it by count before the epilogue, and asks
    for the next one only behind a barrier of the tile (round 3: found by test_random_shapes seed 12 as a run-to-run
    difference).  Dozens of tiles per workgroup, every query-tile width, repeated: identical to the strict mode each time,
    and to the oracle on the first queries."""
    gpu_or_skip()
    rng = np.random.default_rng(4300 + d)
    n, k = 1_500_000, 10
    X = rng.random((n, d), dtype=F)
    Q = rng.random((300, d), dtype=F)
    Q[0] = X[n // 2]
    for metric in (1, 0):
        idx = new_index(d, metric)
        idx.Add(None, X)
        oi, od = oracle.search_batch(metric, Q[:8], X, k, nthreads=8)
        for nq in (6, 64, 100, 128, 300):
            idx.set_candidate_mode(0)
            want = idx.SearchBatch(Q[:nq], k)
            assert_same(want[0][:8][:nq], want[1][:8][:nq], oi[:nq], od[:nq], f"strict d={d} metric={metric} nq={nq}")
            idx.set_candidate_mode(4)
            assert idx.f16_image_bytes > 0
            for rep in range(4):
                lab, dist = idx.SearchBatch(Q[:nq], k)
                assert idx.last_route[0] in (6, 7), idx.last_route
                assert_same(lab, dist, want[0], want[1], f"fp16 d={d} metric={metric} nq={nq} repeat {rep}")
        idx.Close()


def test_fp16_image_follows_the_corpus(oracle):
    """the fp16 copy of the corpus is brought up to date inside Add (appends, growth past the reserved capacity) and dropped
    when the route is no longer on offer; searches equal the oracle at every stage"""
    gpu_or_skip()
    rng = np.random.default_rng(77)
    d, k, nq = 64, 10, 300
    X = (rng.random((9000, d), dtype=F) - F(0.5))
    Q = (rng.random((nq, d), dtype=F) - F(0.5))
    idx = new_index(d, 1)
    idx.set_candidate_mode(4)
    assert idx.f16_image_bytes == 0
    done = 0
    for chunk in (1000, 300, 4000, 3700):      # (the third Add grows the index: the planes move apart, the copy is rebuilt)
        idx.Add(None, X[done:done + chunk])
        done += chunk
        assert idx.f16_image_bytes >= done * d * 2
        lab, dist = idx.SearchBatch(Q, k)
        oi, od = oracle.search_batch(1, Q, X[:done], k, nthreads=8)
        assert_same(lab, dist, oi, od, f"fp16 image after {done} rows")
    idx.set_candidate_mode(0)                  # strict mode: no fp16 route, no copy
    assert idx.f16_image_bytes == 0
    idx.set_candidate_mode(3)                  # AUTO below 16,384 rows: not on offer either
    assert idx.f16_image_bytes == 0
    idx.Close()


def test_fp16_copy_waits_for_free_memory(oracle):
    """memory policy of the fp16 copy (index.hip: sync_f16_image): it is taken only while that leaves max(2 GiB, 1/16 of the
    device) free -- with the device nearly full the index comes up WITHOUT the copy (the fp16 route rounds the f32 rows in
    registers, same results), and the next Add after memory has been released builds it"""
    gpu_or_skip()
    import torch
    rng = np.random.default_rng(78)
    d, k, nq = 64, 10, 200
    X = (rng.random((40000, d), dtype=F) - F(0.5))
    Q = (rng.random((nq, d), dtype=F) - F(0.5))
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    hog = torch.empty(free - (3 << 30), dtype=torch.uint8, device="cuda")  # leaves ~3 GiB: less than the policy keeps free
    try:
        idx = new_index(d, 1)
        idx.set_candidate_mode(4)
        idx.Add(None, X[:30000])
        assert idx.f16_image_bytes == 0, "the copy must not be taken while the device is nearly full"
        lab, dist = idx.SearchBatch(Q, k)
        assert idx.last_route[0] == 6, idx.last_route
        oi, od = oracle.search_batch(1, Q, X[:30000], k, nthreads=8)
        assert_same(lab, dist, oi, od, "fp16 route without the copy (device nearly full)")
    finally:
        del hog
        torch.cuda.empty_cache()
    idx.Add(None, X[30000:])
    assert idx.f16_image_bytes >= 40000 * d * 2, "memory is free again: the Add builds the copy"
    lab, dist = idx.SearchBatch(Q, k)
    oi, od = oracle.search_batch(1, Q, X, k, nthreads=8)
    assert_same(lab, dist, oi, od, "fp16 route over the copy built by the later Add")
    idx.Close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_fp16_route_ties_and_duplicates(oracle, metric):
    """thousands of exact duplicates among the nearest rows: their candidate keys tie AT the sampled threshold (the fp16
    kernels admit key <= threshold: every tie), the lists overflow or the proof fails, and the fallbacks take over -- the answer
    is still the oracle's (equal distances: lowest position first), from the fp16 copy and without it"""
    gpu_or_skip()
    rng = np.random.default_rng(4100 + metric)
    n, d, k = 70000, 64, 20
    X = (rng.random((n, d), dtype=F) - F(0.5))
    v = (rng.random(d, dtype=F) - F(0.5)) * F(1.5)
    X[1000:9000] = v                       # 8000 copies of one vector ...
    X[30000:30100] = F(0)                  # ... and some zero rows
    Q = np.tile(v, (200, 1)).astype(F) + (rng.random((200, d), dtype=F) - F(0.5)) * F(0.01)
    idx = new_index(d, metric)
    idx.Add(None, X)
    oi, od = oracle.search_batch(metric, Q[:40], X, k, nthreads=8)
    idx.set_candidate_mode(4)
    for image in (1, 0):
        idx.set_f16_image(image)
        for nq in (3, 40, 200):
            lab, dist = idx.SearchBatch(Q[:nq], k)
            m = min(nq, 40)
            assert_same(lab[:m], dist[:m], oi[:m], od[:m], f"duplicates metric={metric} image={image} nq={nq}")
    idx.Close()


def test_random_shapes_every_mode_agrees_with_the_strict_one():
    """40 random (corpus size, dimension, batch, k, metric, data shape) draws: the fp16 routes (with the index's fp16 copy and
    without it), the split-bf16 routes and AUTO return the strict mode's lists bit for bit (the strict mode itself is pinned to
    the oracle by the tests above)"""
    gpu_or_skip()
    rng = np.random.default_rng(20261004)
    for case in range(48):
        big = case >= 40  # eight larger corpora: many tiles per persistent workgroup, several query tiles, AUTO's own choices
        d = int(rng.choice([64, 128, 100]) if big else rng.choice([32, 64, 96, 128, 160, 256, 512, 50, 100, 300]))
        n = int(rng.integers(270000, 420000) if big else rng.integers(1, 40000))
        nq = int(rng.integers(1, 1100) if big else rng.integers(1, 700))
        k = int(rng.integers(1, 60))
        metric = int(rng.integers(0, 3))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            X = rng.random((n, d), dtype=F) - F(0.5)
        elif kind == 1:
            X = rng.standard_normal((n, d)).astype(F) * F(rng.choice([0.05, 1.0, 30.0]))
        elif kind == 2:  # a few tight clusters
            c = rng.standard_normal((8, d)).astype(F)
            X = c[rng.integers(0, 8, n)] + rng.standard_normal((n, d)).astype(F) * F(0.01)
        else:            # blocks of exact duplicates and some zero rows
            base = rng.standard_normal((max(1, n // 50), d)).astype(F)
            X = base[rng.integers(0, len(base), n)].copy()
            X[rng.integers(0, n, max(1, n // 100))] = F(0)
        Q = X[rng.integers(0, n, nq)] + rng.standard_normal((nq, d)).astype(F) * F(0.02)
        idx = new_index(d, metric)
        idx.Add(None, X)
        idx.set_candidate_mode(0)
        want = idx.SearchBatch(Q, k)
        for mode, image in ((4, 1), (4, 0), (2, 1), (3, 1)):
            if mode == 2 and d % 32 != 0:
                continue  # (the split-bf16 tiles need whole 128-B lines per K-step)
            idx.set_candidate_mode(mode)
            idx.set_f16_image(image)
            lab, dist = idx.SearchBatch(Q, k)
            assert_same(lab, dist, want[0], want[1],
                        f"case {case}: n={n} d={d} nq={nq} k={k} metric={metric} data={kind} mode={mode} image={image}")
        idx.Close()


@pytest.mark.parametrize("metric", [0, 1])
def test_tight_clusters_widen_the_candidate_list_instead_of_scanning(oracle, metric):
    """clusters of ~350 rows whose distances to a query differ by less than the candidate keys resolve: the proof fails for
    the whole batch at 256 candidates; the batch is redone with four times as many (the whole cluster inside the list, a wide
    gap behind it) instead of one exact scan per query, the widened list is remembered for the next searches, and the
    answers are the oracle's"""
    gpu_or_skip()
    rng = np.random.default_rng(5200 + metric)
    n, d, k, nq = 70000, 96, 20, 300
    centres = rng.standard_normal((200, d)).astype(F)
    X = centres[rng.integers(0, 200, n)] + rng.standard_normal((n, d)).astype(F) * F(0.0005)
    X /= np.linalg.norm(X, axis=1, keepdims=True).astype(F)
    Q = X[rng.integers(0, n, nq)] + rng.standard_normal((nq, d)).astype(F) * F(0.0002)
    idx = new_index(d, metric)
    idx.Add(None, X)
    oi, od = oracle.search_batch(metric, Q[:24], X, k, nthreads=8)
    for mode in (3, 0, 4):
        idx.set_candidate_mode(mode)
        for rep in range(2):  # (the second search starts with the remembered list)
            lab, dist = idx.SearchBatch(Q, k)
            assert_same(lab[:24], dist[:24], oi, od, f"tight clusters metric={metric} mode={mode} rep={rep}")
            assert idx.last_fallbacks <= 8, (mode, rep, idx.last_fallbacks)
    idx.Close()
