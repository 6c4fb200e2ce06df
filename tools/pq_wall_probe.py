#!/usr/bin/env python3
"""Where the wall clock of a C4 search goes beyond the device time: a trajectory of back-to-back searches (clock / power
state), the same under the library's HIP-event profiling (kernel, device), and the fixed host costs around a call.
usage: python tools/pq_wall_probe.py [N_millions=100] [calls=60]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from longbow_amd import _lib, pq
N = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 100_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
M, D, K = 96, 768, 100
lib = _lib.require_gpu(0)
cb = torch.empty((M, 256, D // M), device="cuda")
lib.lb_gpu_fill_uniform_device(0, cb.data_ptr(), cb.numel(), 7, 0, None)
enc = pq.PQEncoder(pq.serialize_codebooks(cb.cpu().numpy()))
codes = torch.empty((N, M), dtype=torch.uint8, device="cuda")
lib.lb_gpu_fill_codes_device(0, codes.data_ptr(), codes.numel(), 99, 0, None)
enc.add_codes_device(N, codes.data_ptr())
del codes
Q = torch.empty((4, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
od = torch.empty((4, K), device="cuda"); ol = torch.empty((4, K), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
t = []
for _ in range(200):
    t0 = time.perf_counter(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
print(f"idle torch.cuda.synchronize: {sorted(t)[100]*1e6:.1f} us")
def run(nq):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc.search_device(nq, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
    return (time.perf_counter() - t0) * 1e3
ts = [run(1) for _ in range(calls)]
print("back-to-back wall ms:", " ".join(f"{x:.3f}" for x in ts))
s = sorted(ts[calls // 2:]); print(f"median of the second half: {s[len(s)//2]:.4f} ms = {N*M/s[len(s)//2]/8e9*100:.1f} % of 8 TB/s")
enc.set_profiling(True)
rows = []
for _ in range(12):
    w = run(1); a, b = enc.last_timing(); rows.append((w, a, b))
enc.set_profiling(False)
print("profiled (wall, kernel, device):", " ".join(f"({w:.3f},{a:.3f},{b:.3f})" for w, a, b in rows))
ts = [run(1) for _ in range(12)]
print("after profiling, wall ms:", " ".join(f"{x:.3f}" for x in ts))
time.sleep(2.0)
print("rested wall ms:", " ".join(f"{run(1):.3f}" for _ in range(6)))
ts = [run(2) for _ in range(12)]
print("B=2 wall ms per query:", " ".join(f"{x/2:.3f}" for x in ts))
