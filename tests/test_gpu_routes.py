"""Path selection is a cost choice between exact routes (index.hip: choose_route): on a sub-grid of dimensions, corpus
sizes and batch sizes the route the library picks must be within 10 % (+ 40 us: launch noise on the small points) of the
best route it can be forced onto.  Forcing a route is a diagnostic-build switch (LB_FORCE_ROUTE), so this test runs on
liblongbow_gpu_diag.so; tools/route_grid.py prints the full grid the constants were fitted to."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from tests.gpu_util import diag_lib, new_index

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ms(idx, Q, B, K, od, ol, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


@pytest.mark.parametrize("D,n", [(128, 1_000_000), (768, 1_000_000), (1536, 500_000), (384, 2_000_000)])
def test_route_choice_is_near_the_best_forced_route(D, n):
    lib = diag_lib()
    from longbow_amd import _lib
    raw = C.CDLL(_lib.DIAG_SO_PATH)
    raw.lb_debug_last_route.restype = C.c_int
    K = 100
    X = torch.empty((n, D), device="cuda")
    Q = torch.empty((1024, D), device="cuda")
    assert lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None) == 0
    assert lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None) == 0
    idx = new_index(D, 1, lib=lib)
    idx.add_device(n, X.data_ptr())
    del X
    bad = []
    try:
        for B in (8, 32, 48, 96, 128, 256, 384, 512, 1024):
            od = torch.empty((B, K), device="cuda")
            ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
            os.environ["LB_FORCE_ROUTE"] = "0"
            idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
            picked = raw.lb_debug_last_route() // 10
            want = (ol.cpu().numpy().copy(), od.cpu().numpy().copy())
            forced = {}
            for r in (1, 2, 3, 5, 6, 7):  # narrow32, narrow64, tall, tall2, tall16, narrow16 (the f32 tile is never competitive: tools/route_grid.py)
                os.environ["LB_FORCE_ROUTE"] = str(r)
                idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
                if raw.lb_debug_last_route() // 10 != r:
                    continue
                # every route is exact: same lists bit for bit
                assert np.array_equal(ol.cpu().numpy(), want[0]) and np.array_equal(od.cpu().numpy(), want[1]), (D, n, B, r)
                forced[r] = _ms(idx, Q, B, K, od, ol)
            os.environ["LB_FORCE_ROUTE"] = "0"
            assert picked in forced, (D, n, B, picked, forced)
            best = min(forced.values())
            if forced[picked] > 1.10 * best + 0.04:
                bad.append((D, n, B, picked, {k: round(v, 3) for k, v in forced.items()}))
    finally:
        os.environ["LB_FORCE_ROUTE"] = "0"
        idx.Close()
    assert not bad, bad
