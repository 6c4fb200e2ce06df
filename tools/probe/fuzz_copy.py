"""One-off randomised sweep over the routes that read the index's fp16 copy: large corpora (many tiles per persistent workgroup),
every dimension class (one / two / many K-steps per tile, odd dimensions), every batch regime; fp16 and AUTO against the strict
mode, each search repeated.  usage: python tools/probe/fuzz_copy.py [seed] [seconds]   (FUZZ_WIDE=1: dimensions 768 .. 4096, k up to 2048, the dense index's maximum)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.gpu_util import F, new_index
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
rng = np.random.default_rng(seed)
t0 = time.time(); case = 0; bad = 0
while time.time() - t0 < budget:
    wide = os.environ.get("FUZZ_WIDE") == "1"  # high dimensions and large k instead of many tiles
    d = int(rng.choice([768, 1000, 1024, 1536, 2048, 4096]) if wide else rng.choice([4, 8, 16, 24, 32, 40, 64, 72, 96, 100, 128, 200, 256, 300, 384]))
    n = int(rng.integers(263_000, 330_000) if wide else rng.integers(280_000, 2_500_000 if d <= 64 else 700_000))
    nq = int(rng.choice([1, 4, 6, 64, 130, 300]) if wide else
             rng.choice([1, 2, 4, 5, 6, 17, 33, 64, 65, 100, 128, 129, 200, 256, 257, 300, 320, 500, 513, 576, 700, 1024, 1100]))
    k = int(rng.choice([1, 100, 1000, 2048]) if wide else rng.choice([1, 5, 10, 37, 100]))
    metric = int(rng.integers(0, 3))
    kind = int(rng.integers(0, 3))
    if kind == 0: X = rng.random((n, d), dtype=F)
    elif kind == 1: X = rng.standard_normal((n, d)).astype(F) * F(rng.choice([0.05, 1.0, 30.0]))
    else:
        c = rng.standard_normal((64, d)).astype(F)
        X = c[rng.integers(0, 64, n)] + rng.standard_normal((n, d)).astype(F) * F(0.05)
    Q = X[rng.integers(0, n, nq)] + rng.standard_normal((nq, d)).astype(F) * F(0.02)
    idx = new_index(d, metric); idx.Add(None, X)
    idx.set_candidate_mode(0); want = idx.SearchBatch(Q, k)
    routes = []
    for mode in (4, 3):
        idx.set_candidate_mode(mode)
        for rep in range(3):
            lab, dist = idx.SearchBatch(Q, k)
            if rep == 0: routes.append(idx.last_route[0])
            if not (np.array_equal(lab, want[0]) and np.array_equal(dist, want[1])):
                bad += 1
                print(f"MISMATCH case {case}: n={n} d={d} nq={nq} k={k} metric={metric} data={kind} mode={mode} rep={rep} route={idx.last_route} "
                      f"rows differing {np.unique(np.argwhere(lab != want[0])[:, 0])[:8]}", flush=True)
    idx.Close()
    print(f"case {case}: n={n} d={d} nq={nq} k={k} metric={metric} data={kind} routes={routes} copy={'yes' if True else ''} t={time.time()-t0:.0f}s", flush=True)
    case += 1
print(f"{case} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
