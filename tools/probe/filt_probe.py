import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from longbow_amd import _lib, gpu
lib = _lib.require_gpu(0)
rows, D, B, K = 1_250_000, 1536, 256, 200
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((B, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 777, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 778, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 2)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
meta = np.random.default_rng(5).integers(0, 100, rows).astype(np.int64)
idx.filter_column(meta, "<", 10)
dd = torch.empty((B, K), device="cuda"); dl = torch.empty((B, K), dtype=torch.int64, device="cuda")
for mode in (3, 0, 2, 4):
    idx.set_candidate_mode(mode)
    for _ in range(3): idx.search_device(B, Q.data_ptr(), K, dd.data_ptr(), dl.data_ptr())
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); idx.search_device(B, Q.data_ptr(), K, dd.data_ptr(), dl.data_ptr()); ts.append(time.perf_counter() - t0)
    idx.set_profiling(True); idx.search_device(B, Q.data_ptr(), K, dd.data_ptr(), dl.data_ptr()); tm = idx.last_timing(); idx.set_profiling(False)
    print(mode, f"{sorted(ts)[3]*1e3:.3f} ms", idx.last_route, {c: (round(tm[c][0]*1e3), tm[c][1]) for c in tm}, "fallbacks", idx.last_fallbacks, flush=True)
