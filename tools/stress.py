import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from longbow_amd import gpu
from oracle import oracle_c as oc
oc.build()
rng = np.random.default_rng(1)
n, d = 200_000, 64
X = rng.random((n, d), dtype=np.float32); Q = rng.random((600, d), dtype=np.float32)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, 0)); idx.Add(None, X)
want = {}
for nq in (1, 3, 7, 30, 130, 600):
    want[nq] = oc.search_batch(0, Q[:nq], X, 10, nthreads=8)
free0 = torch.cuda.mem_get_info()[0]
errs = []
def worker(tid):
    r = np.random.default_rng(tid)
    for it in range(300):
        nq = int(r.choice([1, 3, 7, 30, 130, 600]))
        lab, dist = idx.SearchBatch(Q[:nq], 10)
        if not (np.array_equal(lab, want[nq][0]) and np.array_equal(dist, want[nq][1])):
            errs.append((tid, it, nq)); return
ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
t0 = time.time(); [t.start() for t in ths]; [t.join() for t in ths]
print("concurrent: 6 threads x 300 searches in %.1f s, mismatches: %s" % (time.time() - t0, errs[:3]))
# filter churn
meta = rng.integers(0, 100, n)
for it in range(60):
    sel = int(rng.integers(1, 100))
    mask = (meta < sel).astype(np.uint8)
    idx.set_filter(mask)
    lab, dist = idx.SearchBatch(Q[:9], 10)
    oi, od = oc.search_batch(0, Q[:9], X, 10, mask=mask, nthreads=8)
    assert np.array_equal(lab, oi) and np.array_equal(dist, od), (it, sel)
idx.set_filter(None)
print("filter churn ok; device memory drift: %.1f MB" % ((free0 - torch.cuda.mem_get_info()[0]) / 1e6))
idx.Close()
