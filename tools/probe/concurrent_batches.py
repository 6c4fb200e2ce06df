"""Aggregate queries/s when T host threads each run batched searches on the same index (each call takes its own workspace and
stream from the library's pool): one thread's select / re-rank overlaps another's candidate kernel.
usage: python tools/probe/concurrent_batches.py [batch] [threads...]"""
import os, sys, threading, time
sys.path.insert(0, os.getcwd())
import torch
from longbow_amd import _lib, gpu
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
threads = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4]
rows, D, K = 1_000_000, 768, 100
lib = _lib.require_gpu(0)
X = torch.empty((rows, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
if os.environ.get('CAND_MODE'): idx.set_candidate_mode(int(os.environ['CAND_MODE']))
def worker(t, secs, out):
    Q = torch.empty((B, D), device="cuda"); lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42 + t, 0, None)
    od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3): idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
    barrier.wait()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr()); n += 1
    out[t] = (n, time.perf_counter() - t0)
for T in threads:
    barrier = threading.Barrier(T); out = {}
    ths = [threading.Thread(target=worker, args=(t, 2.0, out)) for t in range(T)]
    [t.start() for t in ths]; [t.join() for t in ths]
    qps = sum(n * B / dt for n, dt in out.values())
    print(f"B={B} threads={T}: {qps:10.0f} queries/s aggregate, {1e3 * sum(dt / n for n, dt in out.values()) / T:.3f} ms per search per thread", flush=True)
