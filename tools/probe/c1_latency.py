"""Config-1 shape on the GPU: 10k x 128 L2, k = 10, one query per call through the device-pointer entry (p50 / p99 by wall clock).
Five launches at this size (init, bootstrap chunk, select, rest, select + emit): 31 us of kernels in a 45 us search.
usage: [N=rows] python tools/probe/c1_latency.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from longbow_amd import _lib, gpu
lib = _lib.require_gpu(0)
rng = np.random.default_rng(0)
n, d, k = int(os.environ.get('N', '10000')), 128, 10
X = rng.random((n, d), dtype=np.float32)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, 0)); idx.Add(None, X)
Q = torch.tensor(rng.random((1, d), dtype=np.float32), device="cuda")
od = torch.empty((1, k), device="cuda"); ol = torch.empty((1, k), dtype=torch.int64, device="cuda")
ts = []
for i in range(300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx.search_device(1, Q.data_ptr(), k, od.data_ptr(), ol.data_ptr())
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[50:])
print(f"{n} x 128 L2 k=10 single query, device pointers: p50 %.1f us  p99 %.1f us" % (ts[len(ts)//2]*1e6, ts[int(len(ts)*0.99)]*1e6))
