import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
from longbow_amd import _lib, pq
from oracle import oracle_c as oc
lib = _lib.require_gpu(0)
dev = torch.device("cuda", 0)
n, dims, M, K = 100_000_000, 768, 96, 100
cb = oc.fill_uniform(M * 256 * (dims // M), 7).reshape(M, 256, dims // M)
enc = pq.PQEncoder(pq.serialize_codebooks(cb))
enc.reserve(n)
CH = 2_000_000
buf = torch.empty((CH, dims), device=dev)
for r0 in range(0, n, CH):
    lib.lb_gpu_fill_uniform_device(0, buf.data_ptr(), CH * dims, 12345, r0 * dims, None)
    enc.add_vectors_device(CH, buf.data_ptr())
del buf
Q = torch.empty((4, dims), device=dev)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
od = torch.empty((4, K), device=dev); ol = torch.empty((4, K), dtype=torch.int64, device=dev)
def run(nq, reps):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        enc.search_device(nq, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
        torch.cuda.synchronize(); ts.append(round(1e3 * (time.perf_counter() - t0), 3))
    return ts
print("prefilter B=1", run(1, 14), flush=True)
print("prefilter B=4", run(4, 6), flush=True)
enc.set_prefilter(False)
print("exact B=1", run(1, 8), flush=True)
enc.set_prefilter(True)
print("prefilter B=1", run(1, 14), flush=True)
time.sleep(3)
print("prefilter B=1 after 3 s idle", run(1, 14), flush=True)
