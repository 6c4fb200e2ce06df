#!/usr/bin/env python3
"""Corpora and filtered views of 16k .. 64k rows (the sizes that take the sampled threshold and the fp16 image since round 4):
a grid of metrics x dimensions x k x batch sizes x {image, no image} x {unfiltered, 25 % visible}; prints time, route,
fallbacks and in-launch give-ups, and flags anything that fell back, gave up or took more than 0.6 ms.
usage: python tools/probe/small_corpus_grid.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from longbow_amd import _lib, gpu
lib = _lib.require_gpu(0)
rng = np.random.default_rng(3)
bad = 0
for metric in (0, 1, 2):
    for D in (100, 768):
        for rows in (20000, 50000, 200000):
            X = torch.empty((rows, D), device="cuda"); Q = torch.empty((1024, D), device="cuda")
            lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345 + rows, 0, None)
            lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
            meta = rng.integers(0, 100, rows).astype(np.int64)
            for image in (1, 0):
                idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, metric))
                if not image: idx.set_f16_image(0)
                idx.add_device(rows, X.data_ptr())
                for filt in (0, 1):
                    if filt and rows < 100000: continue  # (a 25 % view of 200k rows = 50k visible)
                    if filt: idx.filter_column(meta, "<", 25)
                    for K in (10, 300):
                        line = []
                        for B in (1, 8, 16, 32, 64, 256, 1024):
                            od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
                            ts = []
                            for i in range(5):
                                torch.cuda.synchronize(); t0 = time.perf_counter()
                                idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
                                ts.append(time.perf_counter() - t0)
                            t = sorted(ts[1:])[len(ts[1:]) // 2] * 1e3
                            fb, gu = idx.last_fallbacks, idx.fused_giveups
                            flag = "" if (fb == 0 and gu == 0 and t < 0.6 + 0.0012 * B * (rows / 200000) * (D / 768)) else " <<<"
                            if flag: bad += 1
                            line.append(f"{B}:{t:.3f}{'/fb%d' % fb if fb else ''}{'/gu%d' % gu if gu else ''}{flag}")
                        print(f"metric {metric} D {D:4d} rows {rows:6d} image {image} filtered {filt} k {K:3d}  " + "  ".join(line), flush=True)
                idx.Close()
print("flagged cells:", bad)
