"""BASELINE.json's full-size configuration (1M x 768 f32, batch 1024, k = 100) on one MI355X:
spot parity with the oracle on a query subsample plus size-independent properties."""
import numpy as np
import pytest

from tests.gpu_util import gpu_or_skip

pytestmark = pytest.mark.gpu
F = np.float32

N, D, B, K = 1_000_000, 768, 1024, 100


@pytest.fixture(scope="module")
def corpus():
    gpu_or_skip()
    torch = pytest.importorskip("torch")
    from longbow_amd import _lib
    lib = _lib.load()
    X = torch.empty((N, D), device="cuda")
    Q = torch.empty((B, D), device="cuda")
    assert lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None) == 0
    assert lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None) == 0
    return torch, X, Q


@pytest.mark.parametrize("metric", [1, 0, 2])
def test_full_size_batch(oracle, corpus, metric):
    torch, X, Q = corpus
    from longbow_amd import gpu
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=0, Dimension=D, Metric=metric))
    idx.reserve(N)
    idx.add_device(N, X.data_ptr())
    dist = torch.empty((B, K), device="cuda")
    lab = torch.empty((B, K), dtype=torch.int64, device="cuda")
    # the library default (LB_CAND_AUTO: one fp16 product per element over the index's fp16 copy, 256 x 256 tiles) ...
    idx.search_device(B, Q.data_ptr(), K, dist.data_ptr(), lab.data_ptr())
    fallbacks = idx.last_fallbacks
    dist_h, lab_h = dist.cpu().numpy(), lab.cpu().numpy()
    # ... and the strict mode (f32 MFMA candidates, the headline of bench.py): identical lists for ALL 1024 queries
    idx.set_candidate_mode(0)
    idx.search_device(B, Q.data_ptr(), K, dist.data_ptr(), lab.data_ptr())
    assert np.array_equal(lab.cpu().numpy(), lab_h) and np.array_equal(dist.cpu().numpy(), dist_h)
    fallbacks = max(fallbacks, idx.last_fallbacks)
    idx.set_candidate_mode(3)
    # properties: ascending, labels unique and in range
    assert np.all(np.diff(dist_h, axis=1) >= 0)
    assert lab_h.min() >= 0 and lab_h.max() < N
    assert all(len(np.unique(r)) == K for r in lab_h[::64])
    # 4 queries per call, ALL 1024 queries: with the fp16 copy the one-tile candidate pass over it (0.34 ms per call) ...
    d4 = torch.empty((4, K), device="cuda")
    l4 = torch.empty((4, K), dtype=torch.int64, device="cuda")
    assert idx.f16_image_bytes == N * D * 2
    for s in range(0, B, 4):
        idx.search_device(4, Q[s:s + 4].contiguous().data_ptr(), K, d4.data_ptr(), l4.data_ptr())
        assert np.array_equal(l4.cpu().numpy(), lab_h[s:s + 4]) and np.array_equal(d4.cpu().numpy(), dist_h[s:s + 4]), s
    assert idx.last_route[0] == 7, idx.last_route
    # ... and without it the EXACT SCAN path (below the batched path's minimum, 0.55 ms per call): bit for bit the batched
    # results, for every fourth group of 4 and then, without the copy, the whole batch again on the f32-staging form
    idx.set_f16_image(0)
    for s in range(0, B, 16):
        idx.search_device(4, Q[s:s + 4].contiguous().data_ptr(), K, d4.data_ptr(), l4.data_ptr())
        assert np.array_equal(l4.cpu().numpy(), lab_h[s:s + 4]) and np.array_equal(d4.cpu().numpy(), dist_h[s:s + 4]), s
    assert idx.last_route[0] == 0, idx.last_route
    idx.search_device(B, Q.data_ptr(), K, dist.data_ptr(), lab.data_ptr())
    assert np.array_equal(lab.cpu().numpy(), lab_h) and np.array_equal(dist.cpu().numpy(), dist_h)
    fallbacks = max(fallbacks, idx.last_fallbacks)
    idx.set_f16_image(1)
    # oracle on 64 queries spread over the batch (scalar CPU restatement: ~1 s per query per core, 16 threads)
    Xh = X.cpu().numpy()
    qs = np.arange(0, B, 16)
    oi, od = oracle.search_batch(metric, Q.cpu().numpy()[qs], Xh, K, nthreads=16)
    assert np.array_equal(lab_h[qs], oi)
    assert np.array_equal(dist_h[qs], od)
    assert fallbacks <= B // 20, f"{fallbacks} of {B} queries needed the exact-scan fallback"
    idx.Close()


# ---------------------------------------------------------------------------------------------------
# BASELINE config 3, one GPU's view: 10M x 768 f32 dot (30.7 GB resident; the 8-GPU run shards these rows).
# The first sampled span covers ~2.5M rows and the rest follows in growing chunks -- a path the 1M-row
# tests never reach.
# ---------------------------------------------------------------------------------------------------
def test_config3_10m_dot_single_gpu_view(oracle):
    gpu_or_skip()
    torch = pytest.importorskip("torch")
    from longbow_amd import _lib, gpu
    from tests.gpu_util import oracle_topk_rows_parallel
    lib = _lib.load()
    rows, D, K3 = 10_000_000, 768, 100
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 40 * 2**30:
        pytest.skip("needs ~31 GB of HBM")
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=0, Dimension=D, Metric=2))
    CH = 1_000_000
    buf = torch.empty((CH, D), device="cuda")
    Xh = np.empty((rows, D), F)  # host copy for the oracle (30.7 GB)
    for r0 in range(0, rows, CH):  # grows in place, ten appends
        assert lib.lb_gpu_fill_uniform_device(0, buf.data_ptr(), CH * D, 12345, r0 * D, None) == 0
        idx.add_device(CH, buf.data_ptr())
        Xh[r0:r0 + CH] = buf.cpu().numpy()
    del buf
    assert idx.ntotal == rows
    Q = torch.empty((256, D), device="cuda")
    assert lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None) == 0
    res = {}
    for Bq in (1, 256):
        od = torch.empty((Bq, K3), device="cuda")
        ol = torch.empty((Bq, K3), dtype=torch.int64, device="cuda")
        idx.search_device(Bq, Q.data_ptr(), K3, od.data_ptr(), ol.data_ptr())
        res[Bq] = (ol.cpu().numpy(), od.cpu().numpy())
        assert np.all(np.diff(res[Bq][1], axis=1) >= 0)
        assert res[Bq][0].min() >= 0 and res[Bq][0].max() < rows
    assert np.array_equal(res[256][0][:1], res[1][0]) and np.array_equal(res[256][1][:1], res[1][1])
    Qh = Q.cpu().numpy()
    for qi in (0, 37, 74, 111, 148, 185, 222, 255):  # 8 queries against the oracle over all 10M rows
        lab, dist = oracle_topk_rows_parallel(oracle, 2, Qh[qi], Xh, K3, nthreads=16)
        assert np.array_equal(res[256][0][qi], lab) and np.array_equal(res[256][1][qi], dist), qi
    # every query of the batch against the exact scan path (4 per call)
    d4 = torch.empty((4, K3), device="cuda")
    l4 = torch.empty((4, K3), dtype=torch.int64, device="cuda")
    for s in range(0, 256, 4):
        idx.search_device(4, Q[s:s + 4].contiguous().data_ptr(), K3, d4.data_ptr(), l4.data_ptr())
        assert np.array_equal(l4.cpu().numpy(), res[256][0][s:s + 4]) and np.array_equal(d4.cpu().numpy(), res[256][1][s:s + 4]), s
    idx.Close()


# ---------------------------------------------------------------------------------------------------
# BASELINE config 4: 100M x 768 -> PQ (m = 96, 8-bit) codes, ADC k-NN, k = 100 on one GPU.  The codes are
# produced by the GPU encoder from real (synthetic) vectors, 1M rows at a time.
# ---------------------------------------------------------------------------------------------------
def test_config4_100m_pq_adc(oracle):
    gpu_or_skip()
    torch = pytest.importorskip("torch")
    import ctypes as C
    from longbow_amd import _lib, pq
    from tests.gpu_util import oracle_adc_parallel
    lib = _lib.load()
    n, dims, M, K4 = 100_000_000, 768, 96, 100
    cb = oracle.fill_uniform(M * 256 * (dims // M), 7).reshape(M, 256, dims // M)  # codebooks from seed 7
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    enc.reserve(n)
    CH = 2_000_000
    buf = torch.empty((CH, dims), device="cuda")
    for r0 in range(0, n, CH):
        assert lib.lb_gpu_fill_uniform_device(0, buf.data_ptr(), CH * dims, 12345, r0 * dims, None) == 0
        enc.add_vectors_device(CH, buf.data_ptr())
    assert enc.ntotal == n
    # the encoder against the oracle on rows spread over the corpus (vectors regenerated on the host)
    rng = np.random.default_rng(4)
    for r in rng.integers(0, n, 24):
        v = oracle.fill_uniform(dims, 12345, int(r) * dims)
        one = torch.from_numpy(v[None, :]).cuda()
        dc = torch.empty((1, M), dtype=torch.uint8, device="cuda")
        enc.encode_device(1, one.data_ptr(), dc.data_ptr())
        assert np.array_equal(dc.cpu().numpy()[0], oracle.pq_encode(cb, v)), int(r)
    del buf
    NQ4 = 8
    Q = torch.empty((NQ4, dims), device="cuda")
    assert lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None) == 0
    od = torch.empty((NQ4, K4), device="cuda")
    ol = torch.empty((NQ4, K4), dtype=torch.int64, device="cuda")
    enc.search_device(NQ4, Q.data_ptr(), K4, od.data_ptr(), ol.data_ptr())  # (two queries per pass over the codes)
    lab, dist = ol.cpu().numpy().copy(), od.cpu().numpy().copy()
    enc.search_device(1, Q[5:6].contiguous().data_ptr(), K4, od.data_ptr(), ol.data_ptr())  # a single-query pass agrees
    assert np.array_equal(ol.cpu().numpy()[0], lab[5]) and np.array_equal(od.cpu().numpy()[0], dist[5])
    assert np.all(np.diff(dist, axis=1) >= 0) and lab.min() >= 0 and lab.max() < n
    assert all(len(np.unique(r)) == K4 for r in lab)
    enc.set_prefilter(False)  # the exact full pass (no byte-table prefilter) returns the same lists
    enc.search_device(2, Q.data_ptr(), K4, od.data_ptr(), ol.data_ptr())
    enc.set_prefilter(True)
    assert np.array_equal(ol.cpu().numpy()[:2], lab[:2]) and np.array_equal(od.cpu().numpy()[:2], dist[:2])
    # re-rank of the reported rows reproduces the reported distances (gathered codes, same arithmetic)
    Qh = Q.cpu().numpy()
    d2, s2 = enc.Rerank(Qh[0], lab[0])
    assert np.array_equal(d2, dist[0])
    # oracle over ALL 100M codes for every one of the 8 queries (threaded ADC restatement on the downloaded codes)
    codes = enc.get_codes()
    assert codes.shape == (n, M)
    for qi in range(NQ4):
        table = oracle.build_adc_table(cb, Qh[qi])
        d = oracle_adc_parallel(oracle, table, codes, nthreads=16)
        oi, odist, cnt = oracle.topk_canonical(d, K4)
        assert np.array_equal(lab[qi], oi) and np.array_equal(dist[qi], odist)
    enc.Close()


# ---------------------------------------------------------------------------------------------------
# BASELINE config 5 on one GPU's share: 1.25M x 1536 f32 dot, int64 metadata `< 10` (10 % of the rows),
# batch 256, dense side asks for 2k = 200 (internal/store/hybrid_search.go:62), RRF k = 60
# (internal/store/rrf.go:10-51) with a synthetic sparse ranking.
# ---------------------------------------------------------------------------------------------------
def test_config5_filtered_hybrid_share(oracle):
    gpu_or_skip()
    torch = pytest.importorskip("torch")
    from longbow_amd import _lib, gpu, hybrid
    from tests.gpu_util import oracle_topk_rows_parallel
    lib = _lib.load()
    rows, D, Bq, K5 = 1_250_000, 1536, 256, 100
    X = torch.empty((rows, D), device="cuda")
    Q = torch.empty((Bq, D), device="cuda")
    assert lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None) == 0
    assert lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None) == 0
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=0, Dimension=D, Metric=2))
    idx.add_device(rows, X.data_ptr())
    meta = np.random.default_rng(5).integers(0, 100, rows).astype(np.int64)
    idx.filter_column(meta, "<", 10)
    visible = np.flatnonzero(meta < 10)
    dd = torch.empty((Bq, 2 * K5), device="cuda")
    dl = torch.empty((Bq, 2 * K5), dtype=torch.int64, device="cuda")
    idx.search_device(Bq, Q.data_ptr(), 2 * K5, dd.data_ptr(), dl.data_ptr())
    lab, dist = dl.cpu().numpy(), dd.cpu().numpy()
    assert np.all(meta[lab] < 10), "a hidden row was returned"
    assert np.all(np.diff(dist, axis=1) >= 0)
    Xh = X.cpu().numpy()
    Qh = Q.cpu().numpy()
    sub = np.arange(0, Bq, 32)  # 8 queries against the oracle over the visible rows
    for qi in sub:
        ol, od = oracle_topk_rows_parallel(oracle, 2, Qh[qi], Xh, 2 * K5, nthreads=16, visible=visible)
        assert np.array_equal(lab[qi], ol) and np.array_equal(dist[qi], od), int(qi)
    # fusion on the device == store.ReciprocalRankFusion restated, for every query of the batch
    sparse = np.stack([np.random.default_rng(100 + b).permutation(visible)[:2 * K5] for b in range(Bq)]).astype(np.int64)
    sp = torch.from_numpy(sparse).cuda()
    oi = torch.empty((Bq, K5), dtype=torch.int64, device="cuda")
    osc = torch.empty((Bq, K5), device="cuda")
    _lib.check(lib.lb_gpu_rrf_fuse_device(0, Bq, 2 * K5, dl.data_ptr(), 2 * K5, sp.data_ptr(), 60, K5, oi.data_ptr(),
                                          osc.data_ptr(), None))
    fi, fs = oi.cpu().numpy(), osc.cpu().numpy()
    for b in range(0, Bq, 8):
        ri, rs = oracle.rrf(lab[b], sparse[b], 60, K5)
        assert np.array_equal(fi[b], ri) and np.array_equal(fs[b], rs)
    idx.Close()


def test_config5_share_smaller_batches_stay_on_the_image():
    """The config-5 share (1.25M x 1536 dot, 10 % visible) at 32 / 64 / 128 queries: every batch is answered from the fp16
    image by the one-tile kernel under the row list (at exactly 64 and 128 queries -- whole query tiles of the split-bf16
    kernel -- the cost model of round 3 sent it to the f32 rows: 0.30 / 0.40 ms instead of 0.25 / 0.28), nothing falls back to
    the scan, and each query's list equals the one it gets inside the 256-query batch."""
    gpu_or_skip()
    torch = pytest.importorskip("torch")
    from longbow_amd import _lib, gpu
    lib = _lib.load()
    rows, D, K2 = 1_250_000, 1536, 200
    X = torch.empty((rows, D), device="cuda")
    Q = torch.empty((256, D), device="cuda")
    assert lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None) == 0
    assert lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None) == 0
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=0, Dimension=D, Metric=2))
    idx.add_device(rows, X.data_ptr())
    del X
    meta = np.random.default_rng(5).integers(0, 100, rows).astype(np.int64)
    idx.filter_column(meta, "<", 10)
    dd = torch.empty((256, K2), device="cuda")
    dl = torch.empty((256, K2), dtype=torch.int64, device="cuda")
    idx.search_device(256, Q.data_ptr(), K2, dd.data_ptr(), dl.data_ptr())
    assert idx.last_fallbacks == 0
    lab, dist = dl.cpu().numpy(), dd.cpu().numpy()
    assert np.all(meta[lab] < 10), "a hidden row was returned"
    for B in (32, 64, 128):
        bd = torch.empty((B, K2), device="cuda")
        bl = torch.empty((B, K2), dtype=torch.int64, device="cuda")
        idx.search_device(B, Q.data_ptr(), K2, bd.data_ptr(), bl.data_ptr())
        assert idx.last_fallbacks == 0
        assert idx.last_route[0] == 7, (B, idx.last_route)  # ROUTE_NARROW16: the one-tile kernel over the fp16 image
        assert np.array_equal(bl.cpu().numpy(), lab[:B]) and np.array_equal(bd.cpu().numpy(), dist[:B]), B
    idx.Close()
