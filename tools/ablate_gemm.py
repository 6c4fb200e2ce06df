#!/usr/bin/env python3
"""Timing-only ablations of gemm_filter_kernel, interleaved in ONE process (profiling aid).
usage: python tools/ablate_gemm.py [rows]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from longbow_amd import _lib, gpu
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
SPLIT = len(sys.argv) > 2 and sys.argv[2] == "split"
D, B, K = 768, 1024, 100
lib = _lib.require_gpu(0)
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((B, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
if SPLIT:
    idx.set_candidate_mode(1)
od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
raw = C.CDLL(_lib.SO_PATH); raw.lb_debug_set_gemm_ablation.argtypes = [C.c_int]
idx.set_profiling(True)
print('occupancy API: blocks/CU =', raw.lb_debug_gemm_occupancy(), flush=True)
raw.lb_debug_set_gemm_glds.argtypes = [C.c_int]
names = {-1: "GLDS staging (default)", 0: "register staging", 1: "no barrier", 2: "no global loads/LDS writes", 3: "no fragment reads", 4: "MFMA only", 5: "baseline + clock stamps", 6: "no epilogue", 7: "epilogue pass 1 only"}
if SPLIT:
    names = {0: "split baseline", 1: "no barrier", 2: "no global loads", 5: "baseline + clock stamps", 6: "no epilogue"}
res = {k: [] for k in names}
for rnd in range(4):
    for v in names:
        raw.lb_debug_set_gemm_glds(0 if v == -1 else -1)
        raw.lb_debug_set_gemm_ablation(max(v, 0))
        try:
            idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
        except Exception as e:
            print("variant", v, "error", e)
        if rnd > 0:
            res[v].append(idx.last_timing()["gemm"][0])
raw.lb_debug_set_gemm_ablation(0)
raw.lb_debug_set_gemm_glds(0)
probe = (C.c_ulonglong * 8)()
raw.lb_debug_read_clock_probe(probe, 1)
if probe[1]:
    print(f"in-kernel shader clock under load: {probe[0] / probe[1] * 100:.0f} MHz over {probe[2]} workgroups "
          f"(mean cycles: prologue {probe[3] / probe[2]:.0f}, main loop {probe[0] / probe[2]:.0f}, epilogue {probe[4] / probe[2]:.0f})")
fl = (3.0 if SPLIT else 1.0) * 2.0 * B * rows * D
for v, n in names.items():
    ms = sorted(res[v])[len(res[v]) // 2]
    print(f"{n:32s} gemm {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
