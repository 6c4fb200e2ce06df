#!/usr/bin/env python3
"""Timing probe of the fused sample + candidate launch (diagnostic build): when the thresholds are published relative
to the launch's first workgroup, and how long the corpus workgroups wait for them.
usage: LB_GPU_SO=longbow_amd/liblongbow_gpu_diag.so python tools/fused_probe.py [B ...]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from longbow_amd import _lib, gpu
rows, D, K = 1_000_000, 768, 100
lib = _lib.require_gpu(0)
raw = C.CDLL(os.environ.get("LB_GPU_SO", os.path.join(ROOT, "longbow_amd", "liblongbow_gpu_diag.so")))
probe = (C.c_ulonglong * 8)()
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((64, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
for B in [int(x) for x in sys.argv[1:]] or [8, 32]:
    od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
    q = Q[:B].contiguous()
    for i in range(5): idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
    torch.cuda.synchronize()
    raw.lb_debug_read_fused_probe(probe, 1)
    idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr()); torch.cuda.synchronize()
    raw.lb_debug_read_fused_probe(probe, 1)
    t0 = probe[0]
    print(f"B={B}: sample keys out at {(probe[5]-t0)/100:.1f} us, keys loaded by {(probe[6]-t0)/100:.1f}, rounds done by {(probe[7]-t0)/100:.1f}, thresholds published at {(probe[1]-t0)/100:.1f} us; "
          f"{probe[3]} corpus workgroups, {probe[4]} found no threshold yet, mean fetch/wait {probe[2]/max(probe[3],1)/100:.2f} us")
