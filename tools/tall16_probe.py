#!/usr/bin/env python3
"""Where a wave of gemm_filter_tall16_kernel spends its main loop (diagnostic build, LB_F16_ABL=5): cycles per K-step in
s_waitcnt vmcnt, in s_barrier, and the shader clock inside the kernel (the requests go out between the MFMAs).
usage: LB_GPU_SO=longbow_amd/liblongbow_gpu_diag.so LB_F16_ABL=5 python tools/tall16_probe.py [B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from longbow_amd import _lib, gpu
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rows, D, K = 1_000_000, 768, 100
lib = _lib.require_gpu(0)
raw = C.CDLL(_lib.SO_PATH)
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((B, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.add_device(rows, X.data_ptr())
idx.set_candidate_mode(int(os.environ.get("CAND_MODE", "4")))
od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
for _ in range(20): idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
probe = (C.c_ulonglong * 8)()
raw.lb_debug_read_tall16_probe(probe, 1)
for _ in range(10): idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
raw.lb_debug_read_tall16_probe(probe, 1)
cyc, real, waves, wait, bar = [probe[i] for i in range(5)]
nk = D // 32
print(f"waves {waves}  loop cycles/wave {cyc/waves:.0f} = {cyc/waves/nk:.0f} per K-step (16 MFMAs = 512 cycles of one wave; two waves share a SIMD: 1024)")
print(f"  per K-step: vmcnt wait {wait/waves/nk:.0f}  barrier {bar/waves/nk:.0f}  rest (LDS reads, MFMAs, request issue) {((cyc-wait-bar)/waves/nk):.0f}")
print(f"  per tile: prologue {probe[5]/waves:.0f} cycles, loop {cyc/waves:.0f}, epilogue {probe[6]/waves:.0f}")
print(f"  shader clock inside the loop: {cyc/real*100:.0f} MHz")
