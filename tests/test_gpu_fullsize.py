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
    idx.search_device(B, Q.data_ptr(), K, dist.data_ptr(), lab.data_ptr())
    fallbacks = idx.last_fallbacks
    dist_h, lab_h = dist.cpu().numpy(), lab.cpu().numpy()
    # properties: ascending, labels unique and in range
    assert np.all(np.diff(dist_h, axis=1) >= 0)
    assert lab_h.min() >= 0 and lab_h.max() < N
    assert all(len(np.unique(r)) == K for r in lab_h[::64])
    # the batched (MFMA candidate + exact re-rank) path == the exact scan path, bit for bit
    sub = np.arange(0, B, 128)
    for s in sub:
        d1 = torch.empty((1, K), device="cuda")
        l1 = torch.empty((1, K), dtype=torch.int64, device="cuda")
        idx.search_device(1, Q[s:s + 1].contiguous().data_ptr(), K, d1.data_ptr(), l1.data_ptr())
        assert np.array_equal(l1.cpu().numpy()[0], lab_h[s]) and np.array_equal(d1.cpu().numpy()[0], dist_h[s])
    # oracle on a subsample (scalar CPU restatement: ~1 s per query per core)
    Xh = X.cpu().numpy()
    qs = np.arange(0, B, 64)
    oi, od = oracle.search_batch(metric, Q.cpu().numpy()[qs], Xh, K, nthreads=16)
    assert np.array_equal(lab_h[qs], oi)
    assert np.array_equal(dist_h[qs], od)
    assert fallbacks <= B // 20, f"{fallbacks} of {B} queries needed the exact-scan fallback"
    idx.Close()
