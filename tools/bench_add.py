#!/usr/bin/env python3
"""Ingest throughput of lb_gpu_index_add from pageable host memory (the Arrow values buffer): the caller's
buffer pinned for the call (hipHostRegister, default for >= 64 MB batches) vs the double-buffered pinned slabs.
usage: LB_GPU_SO=longbow_amd/liblongbow_gpu_diag.so python tools/bench_add.py   (the A/B switch is a diagnostic-build hook)"""
import ctypes as C, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from longbow_amd import _lib, gpu
lib = _lib.require_gpu(0)
lib.lb_debug_set_add_register_min.argtypes = [C.c_longlong]
n, d = 1_000_000, 768
X = np.random.default_rng(0).random((n, d), dtype=np.float32)
for mode, name in ((64 << 20, "register >= 64 MB (default)"), (0, "pinned slabs only"), (1, "register everything")):
    lib.lb_debug_set_add_register_min(mode)
    for trial in range(2):
        idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, 1))
        t0 = time.perf_counter()
        for i in range(0, n, 100_000):
            idx.Add(None, X[i:i+100_000])
        t = time.perf_counter() - t0
        print(f"[{name}] Add 10 x 100k x {d} f32 (307 MB each): {t*1e3:.1f} ms = {X.nbytes/t/1e9:.1f} GB/s")
        idx.Close()
        idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, 1)); idx.reserve(n)
        t0 = time.perf_counter(); idx.Add(None, X); t = time.perf_counter() - t0
        print(f"[{name}] Add 1M x {d} in one call after reserve: {t*1e3:.1f} ms = {X.nbytes/t/1e9:.1f} GB/s", flush=True)
        idx.Close()
lib.lb_debug_set_add_register_min(64 << 20)
