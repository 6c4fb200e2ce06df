#!/usr/bin/env python3
"""what AUTO does for dimensions that are not multiples of 32 (1M rows): time and route at 1 / 32 / 256 / 1024 queries"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from longbow_amd import _lib, gpu
lib = _lib.require_gpu(0)
n, K = 1_000_000, 100
for D in (100, 300, 96, 320):
    X = torch.empty((n, D), device="cuda"); Q = torch.empty((1024, D), device="cuda")
    lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
    lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.reserve(n); idx.add_device(n, X.data_ptr())
    for B in (1, 32, 256, 1024):
        q = Q[:B].contiguous(); od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
        ts = []
        for i in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter(); idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr()); ts.append(time.perf_counter() - t0)
        print(f"D={D:4d} B={B:5d}: {sorted(ts)[3]*1e3:.3f} ms  route {idx.last_route[2]}  fallbacks {idx.last_fallbacks}", flush=True)
    idx.Close(); del X
