"""Parity of long result lists (k = 600 .. LB_MAX_K) with the oracle on two small corpora, batch and single query.
usage: python tools/probe/big_k_parity.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from longbow_amd import gpu
from oracle import oracle_c as oc
rng = np.random.default_rng(5)
for n, d in ((30000, 32), (120000, 64)):
    X = rng.random((n, d), dtype=np.float32); Q = rng.random((40, d), dtype=np.float32)
    for metric in (0, 1):
        idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=0, Dimension=d, Metric=metric)); idx.Add(None, X)
        for k in (600, 1024, 2048):  # (LB_MAX_K)
            for nq in (1, 40):
                lab, dist = idx.SearchBatch(Q[:nq], k)
                oi, od = oc.search_batch(metric, Q[:nq], X, k, nthreads=8)
                ok = np.array_equal(lab, oi) and np.array_equal(dist, od)
                print(f"n {n} d {d} metric {metric} k {k} nq {nq}: {'ok' if ok else 'MISMATCH'} fallbacks {idx.last_fallbacks}", flush=True)
        idx.Close()
