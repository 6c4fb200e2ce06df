#!/usr/bin/env python3
"""AUTO on embedding-like data (1M x 768 unit vectors around 1000 cluster centres, queries = perturbed rows): route, time,
fallbacks, and agreement with the strict mode.  usage: python tools/probe/clustered_probe.py [noise]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from longbow_amd import _lib, gpu
lib = _lib.require_gpu(0)
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
n, D, K = 1_000_000, 768, 100
g = torch.Generator(device="cuda"); g.manual_seed(1)
C = torch.randn((1000, D), device="cuda", generator=g)
X = torch.empty((n, D), device="cuda")
for s in range(0, n, 100_000):
    idc = torch.randint(0, 1000, (100_000,), device="cuda", generator=g)
    x = C[idc] + noise * torch.randn((100_000, D), device="cuda", generator=g)
    X[s:s + 100_000] = x / x.norm(dim=1, keepdim=True)
qi = torch.randint(0, n, (1024,), device="cuda", generator=g)
Q = X[qi] + 0.05 * torch.randn((1024, D), device="cuda", generator=g) / D ** 0.5
Q = (Q / Q.norm(dim=1, keepdim=True)).contiguous()
for metric in (1, 0):
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, metric)); idx.reserve(n); idx.add_device(n, X.data_ptr())
    for B in (1, 32, 256, 1024):
        q = Q[:B].contiguous()
        od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
        idx.set_candidate_mode(0); idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
        want = (ol.clone(), od.clone())
        idx.set_candidate_mode(3)
        fb = []
        ts = []
        for i in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter(); idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr()); ts.append(time.perf_counter() - t0)
            fb.append(idx.last_fallbacks)
        same = bool(torch.equal(ol, want[0]) and torch.equal(od, want[1]))
        print(f"metric {metric} noise {noise} B={B:5d}: {sorted(ts)[4]*1e3:.3f} ms  route {idx.last_route[2]}  fallbacks per search {fb}  identical to strict: {same}", flush=True)
    idx.Close()
