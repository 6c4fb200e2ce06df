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
