#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU's share: hybrid dense+sparse search with a metadata predicate.
10M x 1536 over 8 GPUs = 1.25M x 1536 per GPU, batch 256, k = 100, predicate `meta < 10` (10 % of the rows),
dense side asks for 2k (internal/store/hybrid_search.go:62), sparse ranking = synthetic id lists (BM25 stays
on the CPU in the reference), fusion = ReciprocalRankFusion(k = 60) on the device.
Prints ms per batch for: predicate -> row mask + visible-row list, dense search, fusion.
usage: python tools/bench_hybrid.py [rows] [dim]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from longbow_amd import _lib, gpu
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
B, K = 256, 100
lib = _lib.require_gpu(0)
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((B, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 2)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
meta = np.random.default_rng(5).integers(0, 100, rows).astype(np.int64)
visible = np.flatnonzero(meta < 10)
sparse = torch.from_numpy(np.random.default_rng(6).choice(visible, (B, 2 * K))).cuda()  # a BM25 stand-in: visible ids
dd = torch.empty((B, 2 * K), device="cuda"); dl = torch.empty((B, 2 * K), dtype=torch.int64, device="cuda")
oi = torch.empty((B, K), dtype=torch.int64, device="cuda"); osc = torch.empty((B, K), device="cuda")


def timed(fn, n=8):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts[2:])[len(ts[2:]) // 2] * 1e3


t_filter = timed(lambda: idx.filter_column(meta, "<", 10))
t_dense = timed(lambda: idx.search_device(B, Q.data_ptr(), 2 * K, dd.data_ptr(), dl.data_ptr()))
t_fuse = timed(lambda: _lib.check(lib.lb_gpu_rrf_fuse_device(0, B, 2 * K, dl.data_ptr(), 2 * K, sparse.data_ptr(), 60, K,
                                                            oi.data_ptr(), osc.data_ptr(), None)))
assert bool((torch.from_numpy(meta).cuda()[dl.clamp(min=0)] < 10).all()), "a hidden row was returned"
print(f"rows {rows} dim {D} batch {B} k {K} (dense side 2k = {2 * K}), 10 % of the rows visible")
print(f"predicate -> mask + visible-row list (incl. 10 MB column upload): {t_filter:.3f} ms")
print(f"dense filtered search, batch {B}:                                  {t_dense:.3f} ms")
print(f"reciprocal-rank fusion on the device:                              {t_fuse:.3f} ms")
print(f"per batch (filter reused across batches): {t_dense + t_fuse:.3f} ms = {B / (t_dense + t_fuse) * 1e3:.0f} queries/s per GPU")
