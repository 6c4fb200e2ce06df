"""Randomised sweep aimed at the NUMERIC RANGE of the fp16 / split-bf16 candidate routes: rows and queries whose norms span many
binades (2^-14 .. 2^+14 per row, or per dimension), sparse rows, rows that are tiny next to one huge row, constant offsets that
cancel in L2, signed and non-negative data.  AUTO and every forced candidate mode against the strict mode (itself oracle-checked
in tests/), and the strict mode against the oracle on the first queries.  usage: python tools/probe/fuzz_scales.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.gpu_util import F, new_index
from oracle import oracle_c as oc
oc.build()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
rng = np.random.default_rng(seed)
t0 = time.time(); case = 0; bad = 0
while time.time() - t0 < budget:
    d = int(rng.choice([32, 64, 96, 128, 256, 384, 768]))
    n = int(rng.integers(262_144, 420_000))
    nq = int(rng.choice([1, 4, 17, 64, 130, 256, 300, 520]))
    k = int(rng.choice([1, 10, 100]))
    metric = int(rng.integers(0, 3))
    kind = int(rng.integers(0, 8))
    base = rng.standard_normal((n, d)).astype(F) if rng.integers(0, 2) else rng.random((n, d), dtype=F)
    lo, hi = (-14, 14) if rng.integers(0, 2) else (-6, 6)
    if kind == 0:    # every row its own binade
        X = base * np.exp2(rng.integers(lo, hi + 1, (n, 1))).astype(F)
    elif kind == 1:  # every dimension its own binade
        X = base * np.exp2(rng.integers(lo, hi + 1, (1, d))).astype(F)
    elif kind == 2:  # one huge row among tiny ones (the corpus' maximum norm is far from the typical norm)
        X = base * F(2.0 ** lo)
        X[rng.integers(0, n, 3)] *= F(2.0 ** (hi - lo))
    elif kind == 3:  # sparse rows (most products are exact zeros)
        X = base * (rng.random((n, d)) < 0.05).astype(F)
    elif kind == 4:  # a large common offset: L2 distances are differences of nearly equal numbers
        X = base * F(0.01) + F(rng.choice([10.0, 1000.0]))
    elif kind == 7:  # the same offset, queries INSIDE the cloud (below): what the centred L2 image is for.  (Kind 4's queries sit
        # 0.02 |mean| per dimension away from a cloud 0.01 wide: every row is then equally far within the rounding of the f32
        # distance itself, (D + 8) 2^-24 d^2 -- no key can certify an order that only the exact sums define)
        X = base * F(0.01) + F(rng.choice([10.0, 1000.0]))
    elif kind == 5:  # all rows uniformly huge or uniformly tiny
        X = base * F(2.0 ** rng.choice([lo, hi]))
    else:            # a few exact duplicates of the queries' rows at mixed scales
        X = base * np.exp2(rng.integers(-3, 4, (n, 1))).astype(F)
        X[rng.integers(0, n, 200)] = X[rng.integers(0, n, 200)]
    X = np.ascontiguousarray(X, dtype=F)
    qrows = rng.integers(0, n, nq)
    Q = X[qrows] + (rng.standard_normal((nq, d)).astype(F) * F(0.02) * np.abs(X[qrows]).mean(1, keepdims=True).astype(F))
    if kind == 7: Q = X[qrows] + rng.standard_normal((nq, d)).astype(F) * F(0.002)
    elif rng.integers(0, 3) == 0: Q = Q * np.exp2(rng.integers(lo, hi + 1, (nq, 1))).astype(F)
    Q = np.ascontiguousarray(Q, dtype=F)
    if not (np.isfinite(X).all() and np.isfinite(Q).all()): continue
    idx = new_index(d, metric); idx.Add(None, X)
    idx.set_candidate_mode(0); want = idx.SearchBatch(Q, k)
    cq = min(nq, 3)
    oi, od = oc.search_batch(metric, Q[:cq], X, k, nthreads=16)
    msg = []
    if not (np.array_equal(want[0][:cq], oi) and np.array_equal(want[1][:cq], od, equal_nan=True)):
        bad += 1; msg.append("strict-vs-oracle")
    routes = []
    for mode in (4, 3, 2, 1):
        try:
            idx.set_candidate_mode(mode)
        except Exception:
            continue
        lab, dist = idx.SearchBatch(Q, k)
        routes.append((mode, idx.last_route[0], int(idx.last_fallbacks)))
        if not (np.array_equal(lab, want[0]) and np.array_equal(dist, want[1], equal_nan=True)):
            bad += 1; msg.append(f"mode{mode} rows {np.unique(np.argwhere(lab != want[0])[:, 0])[:6]}")
    idx.Close()
    print(f"case {case}: n={n} d={d} nq={nq} k={k} metric={metric} kind={kind} range=2^[{lo},{hi}] routes(mode,route,fallbacks)={routes} "
          f"{'MISMATCH ' + '; '.join(msg) if msg else 'ok'} t={time.time()-t0:.0f}s", flush=True)
    case += 1
print(f"{case} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
