"""PQ / ADC path on the GPU vs the oracle (internal/pq/adc_table.go, internal/simd/simd.go:345-355)."""
import numpy as np
import pytest

from tests.gpu_util import gpu_or_skip

pytestmark = pytest.mark.gpu
F = np.float32


def _setup(oracle, dims, M, n, seed):
    from longbow_amd import pq
    rng = np.random.default_rng(seed)
    cb = rng.random((M, 256, dims // M), dtype=F)
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    enc.add_codes(codes)
    return rng, cb, codes, enc


@pytest.mark.parametrize("dims,M", [(32, 4), (768, 96), (60, 5), (128, 16)])
def test_build_adc_table_and_batch(oracle, dims, M):
    gpu_or_skip()
    rng, cb, codes, enc = _setup(oracle, dims, M, 5000, dims + M)
    assert (enc.M, enc.Dims, enc.ntotal) == (M, dims, 5000)
    q = rng.random(dims, dtype=F)
    table = enc.BuildADCTable(q)
    assert table.size == M * 256
    assert np.array_equal(table, oracle.build_adc_table(cb, q))
    res = np.empty(5000, F)
    enc.ADCDistanceBatch(table, res)
    assert np.array_equal(res, oracle.adc_batch(table, codes))
    part = np.empty(100, F)
    enc.ADCDistanceBatch(table, part, row0=1234)
    assert np.array_equal(part, res[1234:1334])
    with pytest.raises(ValueError):
        enc.ADCDistanceBatch(table[:-1], res)               # invalid table size
    with pytest.raises(ValueError):
        enc.ADCDistanceBatch(table, np.empty(5001, F))      # flatCodes buffer too small
    with pytest.raises(ValueError):
        enc.BuildADCTable(q[:-1])                           # query dimension mismatch
    enc.Close()


def test_adc_property_matches_decoded_l2(oracle):
    """pq/adc_test.go:11-66: ADC sum == L2^2(query, decode(code)) within 1e-4"""
    gpu_or_skip()
    rng, cb, codes, enc = _setup(oracle, 32, 4, 1000, 77)
    q = rng.random(32, dtype=F)
    table = enc.BuildADCTable(q)
    res = np.empty(1000, F)
    enc.ADCDistanceBatch(table, res)
    for i in (0, 1, 500, 999):
        dec = oracle.pq_decode(cb, codes[i])
        manual = ((q - dec) ** 2).sum(dtype=np.float64)
        assert abs(float(res[i]) ** 2 - manual) < 1e-4
    enc.Close()


@pytest.mark.parametrize("n,nq,k", [(300, 1, 10), (70000, 3, 100), (5000, 2, 1)])
def test_adc_search(oracle, n, nq, k):
    gpu_or_skip()
    rng, cb, codes, enc = _setup(oracle, 768, 96, n, n + k)
    Q = rng.random((nq, 768), dtype=F)
    lab, dist = enc.Search(Q, k)
    for b in range(nq):
        d = oracle.adc_batch(oracle.build_adc_table(cb, Q[b]), codes)
        oi, od, cnt = oracle.topk_canonical(d, k)
        assert np.array_equal(lab[b], oi) and np.array_equal(dist[b], od)
    enc.Close()


@pytest.mark.parametrize("n,dims,M,k", [(2_600_000, 64, 16, 100), (300_000, 40, 5, 7), (150_000, 32, 16, 1500)])
def test_adc_search_sampled_threshold(oracle, n, dims, M, k):
    """corpora >= 64k codes take the sampled admission threshold (two-level m-th minimum once the sample
    exceeds one list: 2.6M rows -> 10157 sampled rows); heavy duplication makes the threshold tie and
    forces the bootstrap fallback; k = 1500 is beyond what sampling supports at this size"""
    gpu_or_skip()
    rng, cb, codes, enc = _setup(oracle, dims, M, n, n + k)
    Q = rng.random((2, dims), dtype=F)
    lab, dist = enc.Search(Q, k)
    for b in range(2):
        d = oracle.adc_batch(oracle.build_adc_table(cb, Q[b]), codes)
        oi, od, cnt = oracle.topk_canonical(d, k)
        assert np.array_equal(lab[b], oi) and np.array_equal(dist[b], od)
    enc.Close()
    # 50 distinct code rows only: every distance ties thousands of times
    from longbow_amd import pq
    codes2 = codes[rng.integers(0, 50, min(n, 200_000))]
    enc2 = pq.PQEncoder(pq.serialize_codebooks(cb))
    enc2.add_codes(codes2)
    lab, dist = enc2.Search(Q[:1], min(k, 100))
    d = oracle.adc_batch(oracle.build_adc_table(cb, Q[0]), codes2)
    oi, od, cnt = oracle.topk_canonical(d, min(k, 100))
    assert np.array_equal(lab[0], oi) and np.array_equal(dist[0], od)
    enc2.Close()


def test_blob_validation():
    """DeserializePQEncoder error cases (persistence.go:38-56) and the K == 256 restriction"""
    gpu_or_skip()
    import struct
    from longbow_amd import _lib, pq
    with pytest.raises(ValueError):
        pq.PQEncoder(b"\x00" * 8)
    with pytest.raises(ValueError):
        pq.PQEncoder(struct.pack("<III", 33, 4, 256) + bytes(16))
    with pytest.raises(ValueError):
        pq.PQEncoder(struct.pack("<III", 32, 4, 256) + bytes(100))
    with pytest.raises(_lib.LongbowGPUError):
        pq.PQEncoder(struct.pack("<III", 32, 4, 16) + bytes(4 * 16 * 8 * 4))  # K != 256


@pytest.mark.parametrize("dims,M", [(768, 96), (64, 16), (96, 8), (60, 12), (35, 5), (32, 32)])
def test_encode_decode_match_the_oracle(oracle, dims, M):
    """pq.Encode (encoder.go:76-136 -> simd.FindNearestCentroid, simd.go:305-326: sqrt'd 4-accumulator
    distances, FIRST strict minimum) and pq.Decode (encoder.go:139-158), SubDim 8 / 4 / 12 / 5 / 7 / 1"""
    gpu_or_skip()
    from longbow_amd import pq
    rng = np.random.default_rng(dims * 3 + M)
    sub = dims // M
    cb = rng.random((M, 256, sub), dtype=F)
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    n = 700
    V = rng.random((n, dims), dtype=F)
    codes = enc.Encode(V)
    assert codes.shape == (n, M) and codes.dtype == np.uint8
    want = np.stack([oracle.pq_encode(cb, V[i]) for i in range(n)])
    assert np.array_equal(codes, want)
    assert np.array_equal(enc.Encode(V[3]), want[3])
    dec = enc.Decode(codes[:50])
    assert np.array_equal(dec, np.stack([oracle.pq_decode(cb, want[i]) for i in range(50)]))
    assert np.array_equal(enc.Decode(codes[7]), dec[7])
    with pytest.raises(ValueError):
        enc.Encode(V[0][:-1])     # vector dimension mismatch
    with pytest.raises(ValueError):
        enc.Decode(codes[0][:-1])  # code length mismatch
    enc.Close()


def test_encode_ties_first_centroid_wins(oracle):
    """duplicate and near-duplicate centroids on a coarse grid: exact ties and sums that differ in the last
    bits but round to the same float32 sqrt -- the reference keeps the FIRST minimum of the sqrt'd values"""
    gpu_or_skip()
    from longbow_amd import pq
    rng = np.random.default_rng(1234)
    M, sub = 6, 8
    base = (rng.integers(0, 9, (M, 64, sub)) / 8.0).astype(F)
    cb = np.concatenate([base, base, base + F(2.0 ** -20), base[:, ::-1]], axis=1)  # 256 centroids, many ties
    assert cb.shape == (M, 256, sub)
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    V = (rng.integers(0, 17, (3000, M * sub)) / 16.0).astype(F)
    codes = enc.Encode(V)
    want = np.stack([oracle.pq_encode(cb, v) for v in V])
    assert np.array_equal(codes, want)
    assert (codes < 64).mean() > 0.5  # the first copy wins the exact ties
    enc.Close()


def test_search_on_encoded_vectors_and_prefilter_equivalence(oracle):
    """codes produced by Encode on the device (add_vectors_device), searched with and without the byte-table
    prefilter: identical results, equal to the oracle's ADC over the oracle's own codes"""
    gpu_or_skip()
    torch = pytest.importorskip("torch")
    import ctypes as C
    from longbow_amd import _lib, pq
    lib = _lib.load()
    rng = np.random.default_rng(77)
    M, dims, n, k = 96, 768, 120_000, 100
    cb = rng.random((M, 256, dims // M), dtype=F)
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    V = torch.empty((n, dims), device="cuda")
    assert lib.lb_gpu_fill_uniform_device(0, V.data_ptr(), V.numel(), 99, 0, None) == 0
    enc.add_vectors_device(n, V.data_ptr())
    assert enc.ntotal == n
    Vh = V.cpu().numpy()
    sub = rng.integers(0, n, 200)
    codes_sub = np.stack([oracle.pq_encode(cb, Vh[i]) for i in sub])
    codes_all = enc.Encode(Vh)  # host-pointer entry, in pieces
    assert np.array_equal(codes_all[sub], codes_sub)
    assert np.array_equal(enc.get_codes(), codes_all) and np.array_equal(enc.get_codes(5000, 10), codes_all[5000:5010])
    Q = rng.random((3, dims), dtype=F)
    enc.set_prefilter(True)
    lab1, dist1 = enc.Search(Q, k)
    enc.set_prefilter(False)  # exact f32-table pass over every row
    lab0, dist0 = enc.Search(Q, k)
    enc.set_prefilter(True)
    assert np.array_equal(lab0, lab1) and np.array_equal(dist0, dist1)
    for b in range(3):
        d = oracle.adc_batch(oracle.build_adc_table(cb, Q[b]), codes_all)
        oi, od, cnt = oracle.topk_canonical(d, k)
        assert np.array_equal(lab1[b], oi) and np.array_equal(dist1[b], od)
    enc.Close()


def test_prefilter_degenerate_tables(oracle):
    """constant sub-tables (scale 0), heavy duplication (candidate buffer overflow -> exact path) and a
    non-finite query (prefilter refused on the device) all return the exact kernel's results"""
    gpu_or_skip()
    from longbow_amd import pq
    rng = np.random.default_rng(5)
    M, dims, n, k = 16, 64, 200_000, 10
    cb = np.zeros((M, 256, dims // M), F)          # every centroid identical: all ADC distances tie
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    enc.add_codes(codes)
    q = rng.random(dims, dtype=F)
    lab, dist = enc.Search(q, k)
    assert np.array_equal(lab[0], np.arange(k)) and len(np.unique(dist)) == 1
    enc.Close()
    cb = rng.random((M, 256, dims // M), dtype=F)
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    enc.add_codes(codes)
    qbad = q.copy()
    qbad[3] = np.inf
    lab, dist = enc.Search(np.stack([q, qbad]), k)
    d = oracle.adc_batch(oracle.build_adc_table(cb, q), codes)
    oi, od, cnt = oracle.topk_canonical(d, k)
    assert np.array_equal(lab[0], oi) and np.array_equal(dist[0], od)
    d = oracle.adc_batch(oracle.build_adc_table(cb, qbad), codes)
    oi, od, cnt = oracle.topk_canonical(d, k)
    assert np.array_equal(lab[1], oi)
    enc.Close()


def test_two_queries_per_code_pass_equal_single_passes(oracle):
    """batches run the byte-table prefilter for TWO queries per pass over the codes; every query's result must equal its
    own single-query search (and the oracle), for even and odd batch sizes"""
    gpu_or_skip()
    from longbow_amd import pq
    rng = np.random.default_rng(123)
    M, dims, n, k = 16, 128, 300_000, 20
    cb = rng.random((M, 256, dims // M), dtype=F)
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    enc.add_codes(codes)
    Q = rng.random((5, dims), dtype=F)
    single = [enc.Search(Q[i:i + 1], k) for i in range(5)]
    for nq in (2, 3, 4, 5):
        lab, dist = enc.Search(Q[:nq], k)
        for i in range(nq):
            assert np.array_equal(lab[i], single[i][0][0]) and np.array_equal(dist[i], single[i][1][0]), (nq, i)
    for i in (0, 4):
        d = oracle.adc_batch(oracle.build_adc_table(cb, Q[i]), codes)
        oi, od, _ = oracle.topk_canonical(d, k)
        assert np.array_equal(single[i][0][0], oi) and np.array_equal(single[i][1][0], od)
    # a query whose table the prefilter cannot serve (NaN component) rides in a pair with a healthy one
    Qb = Q[:2].copy()
    Qb[1, 3] = np.nan
    lab, dist = enc.Search(Qb, k)
    assert np.array_equal(lab[0], single[0][0][0]) and np.array_equal(dist[0], single[0][1][0])
    lab1, dist1 = enc.Search(Qb[1:2], k)
    assert np.array_equal(lab[1], lab1[0]) and np.array_equal(dist[1], dist1[0], equal_nan=True)
    enc.Close()


def test_concurrent_adc_searches_are_combined_and_identical(oracle):
    """single-query ADC searches from several host threads: calls that overlap are answered by one batch (pairs of queries
    share a pass over the codes) -- every caller gets exactly its own single search's lists; counters move; off = untouched"""
    gpu_or_skip()
    import threading
    from longbow_amd import pq
    rng = np.random.default_rng(321)
    M, dims, n, k = 16, 128, 2_000_000, 20
    cb = rng.random((M, 256, dims // M), dtype=F)
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    enc.add_codes(codes)
    Q = rng.random((24, dims), dtype=F)
    enc.set_search_combining(False)
    want = [enc.Search(Q[i], k) for i in range(len(Q))]
    d = oracle.adc_batch(oracle.build_adc_table(cb, Q[0]), codes)
    oi, od, _ = oracle.topk_canonical(d, k)
    assert np.array_equal(want[0][0][0], oi) and np.array_equal(want[0][1][0], od)
    assert enc.combining_stats == (0, 0)
    enc.set_search_combining(True)
    errors = []

    def caller(t):
        try:
            for rep in range(12):
                for i in range(t, len(Q), 6):
                    lab, dist = enc.Search(Q[i], k)
                    if not (np.array_equal(lab, want[i][0]) and np.array_equal(dist, want[i][1])):
                        errors.append(f"thread {t} query {i} rep {rep}")
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {t}: {type(e).__name__}: {e}")

    ths = [threading.Thread(target=caller, args=(t,)) for t in range(6)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors[:5]
    batches, requests = enc.combining_stats
    assert batches > 0 and requests >= 2 * batches, (batches, requests)
    enc.Close()
