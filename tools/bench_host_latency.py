#!/usr/bin/env python3
"""Latency of the HOST-pointer entry point (what the Go shim calls): lb_gpu_index_search with
numpy buffers, 1M x 768 f32, k = 100.  PCIe-inclusive."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from longbow_amd import _lib, gpu
rows, D, K = 1_000_000, 768, 100
lib = _lib.require_gpu(0)
X = torch.empty((rows, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
Q = np.random.default_rng(0).random((1024, D), dtype=np.float32)
for B in (1, 8, 32, 1024):
    ts = []
    for i in range(12):
        t0 = time.perf_counter(); idx.SearchBatch(Q[:B], K); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[2:])
    print(f"host API B={B:5d}: p50 {ts[len(ts)//2]*1e3:.3f} ms", flush=True)
