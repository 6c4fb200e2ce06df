import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from longbow_amd import gpu
F = np.float32
for noise in (0.0005, 0.005, 0.02):
  for ncl in (200, 50):
    rng = np.random.default_rng(5200)
    n, d, k, nq = 70000, 96, 20, 300
    centres = rng.standard_normal((ncl, d)).astype(F)
    X = centres[rng.integers(0, ncl, n)] + rng.standard_normal((n, d)).astype(F) * F(noise)
    X /= np.linalg.norm(X, axis=1, keepdims=True).astype(F)
    Q = X[rng.integers(0, n, nq)] + rng.standard_normal((nq, d)).astype(F) * F(noise * 0.4)
    for metric in (0, 1):
        idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, d, metric)); idx.Add(None, X)
        out = []
        for mode in (3, 0, 4):
            idx.set_candidate_mode(mode)
            for rep in range(2):
                idx.SearchBatch(Q, k); out.append((mode, idx.last_route[0], idx.last_fallbacks))
        print(noise, ncl, metric, out, flush=True)
        idx.Close()
