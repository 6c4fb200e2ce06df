"""T host threads of single-query ADC searches (host pointers) over 100M x 96 codes, with and without request combining.
usage: python tools/probe/concurrent_pq.py [threads...]"""
import os, sys, threading, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from longbow_amd import _lib, pq
from oracle import oracle_c as oc
threads = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
n, dims, M, K = 100_000_000, 768, 96, 100
lib = _lib.require_gpu(0)
cb = oc.fill_uniform(M * 256 * (dims // M), 7).reshape(M, 256, dims // M)
enc = pq.PQEncoder(pq.serialize_codebooks(cb)); enc.reserve(n)
CH = 2_000_000
buf = torch.empty((CH, dims), device="cuda")
for r0 in range(0, n, CH):
    _lib.check(lib.lb_gpu_fill_uniform_device(0, buf.data_ptr(), CH * dims, 12345, r0 * dims, None)); enc.add_vectors_device(CH, buf.data_ptr())
del buf
Q = np.random.default_rng(1).random((64, dims), dtype=np.float32)
print("encoded", flush=True)
def worker(t, secs, out):
    q = np.ascontiguousarray(Q[t % 64]); od = np.empty(K, np.float32); ol = np.empty(K, np.int64)
    for _ in range(2): lib.lb_gpu_pq_search(enc._h, 1, q.ctypes.data, K, od.ctypes.data, ol.ctypes.data)
    barrier.wait()
    t0 = time.perf_counter(); lat = []
    while time.perf_counter() - t0 < secs:
        a = time.perf_counter(); lib.lb_gpu_pq_search(enc._h, 1, q.ctypes.data, K, od.ctypes.data, ol.ctypes.data); lat.append(time.perf_counter() - a)
    out[t] = (lat, time.perf_counter() - t0)
for comb in (0, 1):
    enc.set_search_combining(comb)
    for T in threads:
        barrier = threading.Barrier(T); out = {}
        ths = [threading.Thread(target=worker, args=(t, 1.5, out)) for t in range(T)]
        [t.start() for t in ths]; [t.join() for t in ths]
        lat = sorted(x for l, _ in out.values() for x in l)
        print(f"combining={comb} threads={T:3d}: {sum(len(l) / dt for l, dt in out.values()):8.1f} queries/s  p50 {lat[len(lat)//2]*1e3:.3f} ms  p99 {lat[int(len(lat)*0.99)]*1e3:.3f} ms  stats {enc.combining_stats}", flush=True)
