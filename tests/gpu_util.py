import numpy as np
import pytest

F = np.float32


def gpu_or_skip():
    from longbow_amd import _lib
    lib = _lib.load()
    if lib.lb_gpu_device_count() <= 0:
        pytest.skip("no MI355X visible")
    return lib


def new_index(dim, metric=0, order=0, device=0, lib=None):
    """lib: another build of the library -- tests that force a fallback path pass diag_lib()"""
    from longbow_amd import gpu
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=device, Dimension=dim, Metric=metric), lib=lib)
    idx.set_order(order)
    return idx


def diag_lib():
    """liblongbow_gpu_diag.so (-DLB_DIAG): the only build that exports the lb_debug_* test hooks"""
    from longbow_amd import _lib
    lib = _lib.load_diag()
    if lib.lb_gpu_device_count() <= 0:
        pytest.skip("no MI355X visible")
    return lib


def assert_same(gpu_labels, gpu_dist, oi, od, ctx=""):
    assert np.array_equal(gpu_labels, oi), f"index sets/order differ {ctx}: {np.argwhere(gpu_labels != oi)[:5]}"
    assert np.array_equal(gpu_dist, od), f"distances differ {ctx}: max abs {np.abs(gpu_dist - od).max()}"


def oracle_topk_rows_parallel(oracle, metric, q, X, k, nthreads=16, visible=None, order=0):
    """canonical top-k of ONE query over a large corpus with the oracle's distances computed row-parallel
    (oracle.search_batch threads over queries only).  visible: ascending row indices the predicate leaves
    (None = all rows).  Returns (labels[k], dist[k]) with -1 / FLT_MAX padding."""
    from concurrent.futures import ThreadPoolExecutor
    rows = np.arange(X.shape[0]) if visible is None else np.asarray(visible)
    n = rows.size
    if n == 0:
        return np.full(k, -1, np.int64), np.full(k, np.finfo(F).max, F)
    step = max(1, (n + nthreads - 1) // nthreads)
    chunks = [(i, min(n, i + step)) for i in range(0, n, step)]

    def work(c):
        a, b = c
        sub = X[a:b] if visible is None else X[rows[a:b]]
        return oracle.batch_flat(metric, q, sub, order)

    with ThreadPoolExecutor(max_workers=nthreads) as ex:
        d = np.concatenate(list(ex.map(work, chunks)))
    oi, od, cnt = oracle.topk_canonical(d, k)
    lab = np.where(oi >= 0, rows[np.clip(oi, 0, n - 1)], -1).astype(np.int64)
    return lab, od


def oracle_adc_parallel(oracle, table, codes, nthreads=16):
    from concurrent.futures import ThreadPoolExecutor
    n = codes.shape[0]
    step = max(1, (n + nthreads - 1) // nthreads)
    with ThreadPoolExecutor(max_workers=nthreads) as ex:
        parts = list(ex.map(lambda a: oracle.adc_batch(table, codes[a:a + step]), range(0, n, step)))
    return np.concatenate(parts)
