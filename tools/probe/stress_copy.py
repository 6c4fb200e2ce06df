#!/usr/bin/env python3
"""concurrent searches of mixed batch sizes from 6 host threads over an index that holds the fp16 copy (persistent kernels,
one workgroup per CU): every answer must equal the single-threaded one; then appends while searches run"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from longbow_amd import gpu
rng = np.random.default_rng(1)
n, d = 300_000, 96
X = rng.random((n + 40_000, d), dtype=np.float32); Q = rng.random((700, d), dtype=np.float32)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, 1)); idx.Add(None, X[:n])
assert idx.f16_image_bytes > 0
sizes = (1, 3, 7, 30, 100, 130, 300, 700)
want = {nq: idx.SearchBatch(Q[:nq], 10) for nq in sizes}
errs = []
def worker(tid):
    r = np.random.default_rng(tid)
    for it in range(400):
        nq = int(r.choice(sizes))
        lab, dist = idx.SearchBatch(Q[:nq], 10)
        if not (np.array_equal(lab, want[nq][0]) and np.array_equal(dist, want[nq][1])):
            errs.append((tid, it, nq)); return
ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
t0 = time.time(); [t.start() for t in ths]; [t.join() for t in ths]
print("concurrent: 6 threads x 400 searches in %.1f s, mismatches: %s" % (time.time() - t0, errs[:3]), flush=True)
# appends (the copy is brought up to date under the exclusive lock) while searches run: answers must be those of SOME prefix
stop = False
bad = []
def searcher():
    while not stop:
        lab, dist = idx.SearchBatch(Q[:50], 10)
        if lab.min() < 0 or not np.all(np.diff(dist, axis=1) >= 0): bad.append(1)
th = [threading.Thread(target=searcher) for _ in range(3)]
[t.start() for t in th]
for s in range(n, n + 40_000, 4000):
    idx.Add(None, X[s:s + 4000])
stop = True; [t.join() for t in th]
idx.set_candidate_mode(0); ref = idx.SearchBatch(Q[:300], 10); idx.set_candidate_mode(3); got = idx.SearchBatch(Q[:300], 10)
print("appends under searches: malformed answers %d; after: copy %d bytes, AUTO == strict: %s" %
      (len(bad), idx.f16_image_bytes, bool(np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]))), flush=True)
