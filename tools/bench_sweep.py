#!/usr/bin/env python3
"""Batch-size sweep on one MI355X: queries/s and effective corpus-read rate vs batch size
(1M x 768 f32 cosine, k = 100), with the library's own per-class device timing.
usage: python tools/bench_sweep.py [rows]   (SWEEP=1,2,4,... selects the batch sizes, CAND_MODE=1|2 the opt-in candidate modes)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from longbow_amd import _lib, gpu
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D, K = 768, 100
lib = _lib.require_gpu(0)
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((1024, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
if os.environ.get('CAND_MODE'): idx.set_candidate_mode(int(os.environ['CAND_MODE']))
if os.environ.get('F16_IMAGE'): idx.set_f16_image(int(os.environ['F16_IMAGE']))  # 0: the fp16 route rounds f32 rows in registers
print(f'fp16 image: {idx.f16_image_bytes / 1e9:.2f} GB', flush=True)  # 1 = split image, 2 = split in registers
for B in [int(x) for x in os.environ.get('SWEEP', '1,2,4,8,9,16,24,32,48,64,96,97,128,256,512,1024').split(',')]:
    od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
    q = Q[:B].contiguous()
    ts = []
    for i in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
        ts.append(time.perf_counter() - t0)
    t = sorted(ts[2:])[len(ts[2:]) // 2]
    idx.set_profiling(True)
    idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
    tm = idx.last_timing()
    idx.set_profiling(False)
    cls = " ".join(f"{c}={tm[c][0]*1e3:.0f}us/{tm[c][1]}" for c in ("gemm", "scan", "select", "rerank", "total"))
    print(f"B={B:5d}  {t*1e3:8.3f} ms/batch  {B/t:10.0f} q/s  corpus-read-equivalent {4.0*rows*D/t/1e12:5.2f} TB/s  fallbacks {idx.last_fallbacks} giveups {idx.fused_giveups}  [{cls}]", flush=True)
