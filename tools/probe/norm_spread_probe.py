#!/usr/bin/env python3
"""AUTO on unnormalised L2 data at the benchmark's size (1M x 768): embedding-like rows around 1000 centres, every row scaled by
its own factor 2^U(-spread, +spread), queries = perturbed rows.  Route, time, fallbacks, agreement with the strict mode, and the
oracle on two queries.  usage: python tools/probe/norm_spread_probe.py [spread=2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from longbow_amd import _lib, gpu
from oracle import oracle_c as oc
oc.build()
lib = _lib.require_gpu(0)
spread = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n, D, K = 1_000_000, 768, 100
g = torch.Generator(device="cuda"); g.manual_seed(1)
C = torch.randn((1000, D), device="cuda", generator=g)
X = torch.empty((n, D), device="cuda")
for s in range(0, n, 100_000):
    idc = torch.randint(0, 1000, (100_000,), device="cuda", generator=g)
    x = C[idc] + 0.5 * torch.randn((100_000, D), device="cuda", generator=g)
    x = x / x.norm(dim=1, keepdim=True)
    X[s:s + 100_000] = x * torch.exp2((torch.rand((100_000, 1), device="cuda", generator=g) * 2 - 1) * spread)
qi = torch.randint(0, n, (1024,), device="cuda", generator=g)
Q = (X[qi] * (1 + 0.05 * torch.randn((1024, D), device="cuda", generator=g))).contiguous()
nr = X.norm(dim=1)
print(f"row norms: min {nr.min().item():.3f} median {nr.median().item():.3f} max {nr.max().item():.3f}", flush=True)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 0)); idx.reserve(n); idx.add_device(n, X.data_ptr())
Xh = None
for B in (1, 32, 256, 1024):
    q = Q[:B].contiguous()
    od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
    idx.set_candidate_mode(0); idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
    want = (ol.clone(), od.clone()); fb0 = idx.last_fallbacks
    idx.set_candidate_mode(3)
    fb, ts = [], []
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr()); ts.append(time.perf_counter() - t0)
        fb.append(idx.last_fallbacks)
    same = bool(torch.equal(ol, want[0]) and torch.equal(od, want[1]))
    print(f"L2 spread 2^+-{spread} B={B:5d}: {sorted(ts)[4]*1e3:.3f} ms  route {idx.last_route[2]}  fallbacks per search {fb} (strict mode: {fb0})  identical to strict: {same}", flush=True)
    if B == 32:
        Xh = X.cpu().numpy()
        oi, odist = oc.search_batch(0, q[:2].cpu().numpy(), Xh, K, nthreads=16)
        print("   oracle on 2 queries:", bool(np.array_equal(ol[:2].cpu().numpy(), oi) and np.array_equal(od[:2].cpu().numpy(), odist)), flush=True)
idx.Close()
