"""Materialise the inputs of tests/golden/reference_kats.json cases (generator
formulas restated from the reference's tests; see oracle/gen_golden.py)."""
import numpy as np

F = np.float32


def make_test_vector(dim, seed):
    i = np.arange(1, dim + 1, dtype=F)
    return (F(seed) * i) * F(0.1)


def pair_inputs(case):
    if "gen" in case:
        g = case["gen"]
        assert g["kind"] == "makeTestVector"
        return make_test_vector(g["dim"], g["seeds"][0]), make_test_vector(g["dim"], g["seeds"][1])
    return np.array(case["a"], F), np.array(case["b"], F)


def dataset(gen):
    k = gen["kind"]
    if k == "ds4":
        n = gen["n"]
        return (np.arange(n * 4, dtype=np.int64).astype(F) * F(0.01)).reshape(n, 4)
    if k == "ramp":
        n, d = gen["n"], gen["dim"]
        return (np.arange(n)[:, None] + np.arange(d)[None, :]).astype(F)
    if k == "flat_scaled":
        n, d = gen["n"], gen["dim"]
        return (np.arange(n * d, dtype=np.int64).astype(F) * F(gen["scale"])).reshape(n, d)
    if k == "mod10":
        d, nv = gen["dim"], gen["nvec"]
        q = (np.arange(d) % 10).astype(F) / F(10.0)
        V = np.stack([((np.arange(d) + j) % 10).astype(F) / F(10.0) for j in range(nv)])
        return q, V
    raise KeyError(k)


def approx_equal(a, b, rel_tol):
    """approxEqual (internal/simd/simd_test.go:374-391)"""
    diff = abs(float(a) - float(b))
    m = max(abs(float(a)), abs(float(b)), 1.0) if max(float(a), float(b)) >= 0 else max(-max(float(a), float(b)), 1.0)
    return diff < rel_tol * m if rel_tol > 0 else diff == 0
