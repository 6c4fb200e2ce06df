import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from longbow_amd import gpu
from oracle import oracle_c as oc
oc.build()
F = np.float32
rng = np.random.default_rng(3)
def check(n, d, k, nq, metric=0, tag=""):
    X = rng.random((n, d), dtype=F); Q = rng.random((nq, d), dtype=F)
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, metric)); idx.Add(None, X)
    lab, dist = idx.SearchBatch(Q, k)
    oi, od = oc.search_batch(metric, Q, X, k, nthreads=8)
    ok = np.array_equal(lab, oi) and np.array_equal(dist, od)
    print(f"{tag or ''} n={n} d={d} k={k} nq={nq} metric={metric}: {'ok' if ok else 'MISMATCH'} fallbacks={idx.last_fallbacks}", flush=True)
    idx.Close()
    assert ok
check(100_000, 2, 10, 1); check(100_000, 2, 10, 40, 1); check(100_000, 3, 5, 300, 2)
check(80_000, 16, 10, 5000, 0, "nq > internal batch")
check(70_000, 32, 2048, 3, 1, "k=2048")
check(200_000, 8, 1, 1, 2, "k=1")
check(66_000, 4, 100, 9, 0)
# non-finite inputs: NaN distances rank last, ties by row
X = rng.random((70_000, 16), dtype=F); X[5] = np.inf; X[6] = np.nan; X[7] = -np.inf
Q = rng.random((12, 16), dtype=F); Q[3, 0] = np.nan
for metric in (0, 1, 2):
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, 16, metric)); idx.Add(None, X)
    for nq in (1, 12):
        lab, dist = idx.SearchBatch(Q[:nq], 10)
        oi, od = oc.search_batch(metric, Q[:nq], X, 10, nthreads=4)
        same = np.array_equal(lab, oi) and np.array_equal(dist, od, equal_nan=True)
        print(f"non-finite rows, metric {metric} nq {nq}: canonical order (NaN last) agrees with the oracle: {same}", flush=True)
        assert same
    idx.Close()
print("robustness ok")
