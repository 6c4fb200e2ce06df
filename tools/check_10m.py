#!/usr/bin/env python3
"""One-off validation at BASELINE config-3 scale on a single GPU: 10M x 768 f32 dot (30.7 GB in HBM).
The first sampled span covers ~2.5M rows, the rest follows in growing chunks -- a path the 1M-row tests never
reach.  Checks: every batch regime returns identical results for the same queries; two queries against the
CPU oracle on the same counter-generated data.  usage: python tools/check_10m.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from longbow_amd import _lib, gpu
from oracle import oracle_c as oc
oc.build()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D, K = 768, 100
lib = _lib.require_gpu(0)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 2)); idx.reserve(rows)
CH = 1_000_000
buf = torch.empty((CH, D), device="cuda")
for r0 in range(0, rows, CH):
    n = min(CH, rows - r0)
    lib.lb_gpu_fill_uniform_device(0, buf.data_ptr(), n * D, 12345, r0 * D, None)
    idx.add_device(n, buf.data_ptr())
del buf
Q = torch.empty((256, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
res = {}
for B in (1, 4, 16, 64, 256):
    od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
        ts.append(time.perf_counter() - t0)
    res[B] = (ol.cpu().numpy(), od.cpu().numpy())
    print(f"B={B:4d}: {min(ts)*1e3:9.3f} ms  ({4.0*rows*D/min(ts)/1e12:.2f} TB/s corpus-read-equivalent)  fallbacks {idx.last_fallbacks}", flush=True)
for B in (4, 16, 64, 256):
    for Bs in (1, 4, 16, 64):
        if Bs < B:
            assert np.array_equal(res[B][0][:Bs], res[Bs][0]) and np.array_equal(res[B][1][:Bs], res[Bs][1]), (B, Bs)
print("all batch regimes agree on the shared queries")
Xh = oc.fill_uniform(rows * D, 12345).reshape(rows, D)
Qh = Q[:2].cpu().numpy()
t0 = time.time()
oi, od = oc.search_batch(2, Qh, Xh, K, nthreads=16)
print(f"oracle: 2 queries in {time.time()-t0:.1f} s")
assert np.array_equal(res[4][0][:2], oi) and np.array_equal(res[4][1][:2], od)
print("GPU == oracle (ids and distances, bitwise) at", rows, "rows")
