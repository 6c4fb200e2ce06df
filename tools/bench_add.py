import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from longbow_amd import gpu
n, d = 1_000_000, 768
X = np.random.default_rng(0).random((n, d), dtype=np.float32)
for trial in range(2):
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, 1))
    t0 = time.perf_counter()
    for i in range(0, n, 100_000):
        idx.Add(None, X[i:i+100_000])
    t = time.perf_counter() - t0
    print(f"Add 10 x 100k x {d} f32 from pageable host memory: {t*1e3:.1f} ms = {X.nbytes/t/1e9:.1f} GB/s (incl. geometric growth copies)")
    idx.Close()
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, 1)); idx.reserve(n)
    t0 = time.perf_counter(); idx.Add(None, X); t = time.perf_counter() - t0
    print(f"Add 1M x {d} in one call after reserve: {t*1e3:.1f} ms = {X.nbytes/t/1e9:.1f} GB/s")
    idx.Close()
