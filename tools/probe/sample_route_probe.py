#!/usr/bin/env python3
"""A/B of the sample launch of a large-batch search (diagnostic build: LB_TALL_SAMPLE_NARROW_MAXQ in the environment: batches above
it sample through the fp16 kernel itself, the others through the 64-query narrow tile):
1M x 768 cosine, AUTO, 512 / 1024 queries; wall per search and the library's per-class device times.
usage: LB_TALL_SAMPLE_NARROW_MAXQ=0 python tools/probe/sample_route_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from longbow_amd import _lib, gpu
lib = _lib.load_diag()
n, D, K = 1_000_000, 768, 100
X = torch.empty((n, D), device="cuda"); Q = torch.empty((1024, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1), lib=lib); idx.reserve(n); idx.add_device(n, X.data_ptr())
for B in (256, 320, 384, 512, 768, 1024):
    q = Q[:B].contiguous()
    od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
    ts = []
    for i in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter(); idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr()); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[5:])
    print(f"LB_TALL_SAMPLE_NARROW_MAXQ={os.environ.get('LB_TALL_SAMPLE_NARROW_MAXQ','(default)')} B={B}: median {ts[len(ts)//2]*1e3:.4f} ms  min {ts[0]*1e3:.4f}  route {idx.last_route[2]} fallbacks {idx.last_fallbacks} checksum {int(ol.sum().item())}", flush=True)
idx.Close()
