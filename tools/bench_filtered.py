#!/usr/bin/env python3
"""Filtered search (BASELINE config-5 shape on one GPU's shard): 1.25M x 1536 f32 dot, predicate
keeping SEL % of the rows, batch 256 / 1, k = 100.  Prints ms/batch with the compacted visible-row
list (default) — run again with LB_ROWMAP_MAX_PCT=0 to time the per-row mask test instead.
usage: python tools/bench_filtered.py [rows] [dim]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from longbow_amd import _lib, gpu
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
K = int(os.environ.get('K', '100'))
lib = _lib.require_gpu(0)
import ctypes as C
raw = C.CDLL(_lib.SO_PATH)
have_probe = hasattr(raw, "lb_debug_read_finish_probe")  # diagnostic build: members the finish launch re-ranks per query
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((1024, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 2)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
meta = np.random.default_rng(5).integers(0, 100, rows).astype(np.int64)
if os.environ.get("META") == "sorted":  # the visible rows are one contiguous block (what the scatter itself costs: compare)
    meta = (np.arange(rows, dtype=np.int64) * 100) // rows
print(f"rows {rows} dim {D} LB_ROWMAP_MAX_PCT={os.environ.get('LB_ROWMAP_MAX_PCT', '(default 95)')}", flush=True)
for sel in [int(x) for x in os.environ.get("SELS", "100,50,10,1").split(",")]:
    t0 = time.perf_counter()
    if sel == 100:
        idx.set_filter(None)
    else:
        idx.filter_column(meta, "<", sel)
    tf = time.perf_counter() - t0
    for B in [int(x) for x in os.environ.get("BS", "1,32,256,1024").split(",")]:
        od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
        q = Q[:B].contiguous()
        ts = []
        for i in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
            ts.append(time.perf_counter() - t0)
        t = sorted(ts[2:])[len(ts[2:]) // 2]
        idx.set_profiling(True)
        probe = (C.c_ulonglong * 8)()
        if have_probe: raw.lb_debug_read_finish_probe(probe, 1)
        idx.search_device(B, q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
        if have_probe: raw.lb_debug_read_finish_probe(probe, 1)
        tm = idx.last_timing()
        idx.set_profiling(False)
        cls = " ".join(f"{c}={tm[c][0]*1e3:.0f}us/{tm[c][1]}" for c in ("gemm", "scan", "select", "rerank"))
        vis = rows * sel / 100.0
        if have_probe and probe[1]:
            cls += f" members/query {probe[0] / probe[1]:.0f} (max {probe[3]}) of {probe[2] / probe[1]:.0f} list entries"
        print(f"sel {sel:3d} %  B={B:5d}  {t*1e3:8.3f} ms/batch  {B/t:10.0f} q/s  visible-f32-bytes/8TBs {4.0*vis*D/t/8e12:5.3f}  "
              f"route {idx.last_route[2]}  fallbacks {idx.last_fallbacks}  [{cls}]  (filter set in {tf*1e3:.1f} ms incl. column upload)", flush=True)
