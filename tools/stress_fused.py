import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from longbow_amd import gpu
from oracle import oracle_c as oc
oc.build()
rng = np.random.default_rng(1)
n, d = 300_000, 64
X = rng.random((n, d), dtype=np.float32); Q = rng.random((600, d), dtype=np.float32)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, 1)); idx.Add(None, X)
want = {}
for nq in (1, 5, 8, 16, 30, 32, 130):
    want[nq] = oc.search_batch(1, Q[:nq], X, 10, nthreads=8)
errs = []; fb = [0]
def worker(tid):
    r = np.random.default_rng(tid)
    for it in range(400):
        nq = int(r.choice([1, 5, 8, 16, 30, 32, 130]))
        lab, dist = idx.SearchBatch(Q[:nq], 10)
        if not (np.array_equal(lab, want[nq][0]) and np.array_equal(dist, want[nq][1])):
            errs.append((tid, it, nq)); return
ths = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
t0 = time.time(); [t.start() for t in ths]; [t.join() for t in ths]
print("concurrent: 8 threads x 400 searches in %.2f s, mismatches: %s" % (time.time() - t0, errs[:3]))
t0 = time.time()
for it in range(400):
    lab, dist = idx.SearchBatch(Q[:16], 10)
print("sequential 400 x 16 queries: %.3f ms each, last fallbacks %d" % ((time.time() - t0) / 400 * 1e3, idx.last_fallbacks))
idx.Close()
