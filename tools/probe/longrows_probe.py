"""Dot product on a corpus with a few very long rows (their products are uncertain by gamma_a |q||x|: a plain key widens every
query's proof by the longest row's share): fallbacks, times and parity against the strict mode per batch size.
usage: python tools/probe/longrows_probe.py [rows] [dim] [log2 of the long rows' factor]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from longbow_amd import _lib, gpu
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
lg = int(sys.argv[3]) if len(sys.argv) > 3 else 8
K = 100
lib = _lib.require_gpu(0)
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((1024, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
X.sub_(0.5); Q.sub_(0.5)                       # signed data: the long rows do not simply win every query
for r in (7, rows // 2, rows - 3): X[r] *= float(2 ** lg)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 2)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
print(f"dot, 3 rows x 2^{lg}: fp16 image {idx.f16_image_bytes / 1e9:.2f} GB", flush=True)
for B in (1, 32, 256, 1024):
    od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
    q = Q[:B].contiguous()
    idx.set_candidate_mode(0)
    idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
    wl, wd = ol.cpu().numpy().copy(), od.cpu().numpy().copy()
    fb0 = idx.last_fallbacks
    idx.set_candidate_mode(3)
    ts = []
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
        ts.append(time.perf_counter() - t0)
    same = np.array_equal(ol.cpu().numpy(), wl) and np.array_equal(od.cpu().numpy(), wd)
    print(f"  B={B:5d}  AUTO {sorted(ts[1:])[2]*1e3:8.3f} ms  route {idx.last_route[2]}  fallbacks {idx.last_fallbacks} (strict: {fb0})  identical to strict: {same}", flush=True)
idx.Close()
