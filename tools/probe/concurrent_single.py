"""Aggregate queries/s of T host threads that each call the host-pointer single-query search (the reference's gpu.Index.Search)
on one index, with and without request combining.  usage: python tools/probe/concurrent_single.py [threads...]"""
import os, sys, threading, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from longbow_amd import _lib, gpu
threads = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16]
rows, D, K = 1_000_000, 768, 100
lib = _lib.require_gpu(0)
X = torch.empty((rows, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
Q = np.random.default_rng(1).random((64, D), dtype=np.float32)
def worker(t, secs, out):
    q = np.ascontiguousarray(Q[t % 64]); od = np.empty(K, np.float32); ol = np.empty(K, np.int64)
    for _ in range(3): lib.lb_gpu_index_search(idx._h, 1, q.ctypes.data, K, od.ctypes.data, ol.ctypes.data)
    barrier.wait()
    t0 = time.perf_counter(); n = 0; lat = []
    while time.perf_counter() - t0 < secs:
        a = time.perf_counter(); lib.lb_gpu_index_search(idx._h, 1, q.ctypes.data, K, od.ctypes.data, ol.ctypes.data); lat.append(time.perf_counter() - a); n += 1
    out[t] = (n, time.perf_counter() - t0, sorted(lat))
for comb in (0, 1):
    idx.set_search_combining(comb)
    for T in threads:
        barrier = threading.Barrier(T); out = {}
        ths = [threading.Thread(target=worker, args=(t, 1.5, out)) for t in range(T)]
        [t.start() for t in ths]; [t.join() for t in ths]
        qps = sum(n / dt for n, dt, _ in out.values())
        lat = sorted(x for _, _, l in out.values() for x in l)
        print(f"combining={comb} threads={T:3d}: {qps:9.0f} queries/s  p50 {lat[len(lat)//2]*1e3:.3f} ms  p99 {lat[int(len(lat)*0.99)]*1e3:.3f} ms  stats {idx.combining_stats}", flush=True)
