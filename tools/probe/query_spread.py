#!/usr/bin/env python3
"""Does a single-query search's time depend on the query?  64 different queries, each timed as bench.py's p50 leg does (one
call, synchronised on both sides), 7 calls each; prints the spread of the per-query medians and the kernel classes of the
fastest and slowest.  usage: python tools/probe/query_spread.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from longbow_amd import _lib, gpu
rows, D, K = 1_000_000, 768, 100
lib = _lib.require_gpu(0)
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((1024, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
d1 = torch.empty((1, K), device="cuda"); l1 = torch.empty((1, K), dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
def once(q):
    torch.cuda.synchronize(); t = time.perf_counter()
    idx.search_device(1, q.data_ptr(), K, d1.data_ptr(), l1.data_ptr(), stream)
    return 1e3 * (time.perf_counter() - t)
for i in range(20): once(Q[0:1])
med = []
for i in range(64):
    q = Q[i:i + 1].contiguous()
    ts = sorted(once(q) for _ in range(7))
    med.append((ts[3], i))
med.sort()
print("per-query medians (ms): min %.4f  p25 %.4f  p50 %.4f  p75 %.4f  max %.4f" % (med[0][0], med[16][0], med[32][0], med[48][0], med[-1][0]))
idx.set_profiling(True)
for t, i in (med[0], med[1], med[-2], med[-1]):
    q = Q[i:i + 1].contiguous(); once(q); tm = idx.last_timing()
    print(f"query {i:3d}: {t:.4f} ms  " + " ".join(f"{c}={tm[c][0]*1e3:.0f}us" for c in ("gemm", "select", "rerank", "total")))
idx.set_profiling(False)
# the p50 leg's own pattern: a different query every call
lat = sorted(once(Q[i % 1024:i % 1024 + 1].contiguous()) for i in range(224))
print("a different query every call: p50 %.4f p99 %.4f ms" % (lat[112], lat[221]))
lat = sorted(once(Q[0:1]) for i in range(224))
print("the same query every call:    p50 %.4f p99 %.4f ms" % (lat[112], lat[221]))
