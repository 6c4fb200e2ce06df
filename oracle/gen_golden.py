#!/usr/bin/env python3
"""Generate tests/golden/reference_kats.json: the reference's own known-answer tests
for the k-NN hot path, restated as data (inputs + expected outputs).

Run: python oracle/gen_golden.py      (rewrites tests/golden/reference_kats.json)

Each case cites the reference test it comes from.  Inputs are either the test's
literals or its deterministic generator formula evaluated here in f32; expected
values are either the literal the reference test asserts, or the value of the
in-test reference formula (referenceEuclidean / referenceCosine,
internal/simd/simd_test.go:13-33) evaluated by the numpy restatement
(oracle/oracle_np.py).  Nothing is read from /root/reference at run time and no
reference source text is stored: only numbers.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as onp  # noqa: E402

F = np.float32
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "tests", "golden", "reference_kats.json")


def make_test_vector(dim, seed):
    """makeTestVector (internal/simd/simd_test.go:366-372): v[i] = seed*float32(i+1)*0.1"""
    i = np.arange(1, dim + 1, dtype=F)
    return (F(seed) * i) * F(0.1)


def fl(x):
    return [float(v) for v in np.asarray(x, F).reshape(-1)]


def main():
    cases = []

    # (1) simd_test.go:105-121,196-213,238-260 -- makeTestVector pairs over dims
    dims_l2cos = [1, 3, 7, 8, 15, 16, 31, 32, 64, 128, 256, 384, 512, 768, 1024, 1536]
    for d in dims_l2cos:
        a, b = make_test_vector(d, 1.0), make_test_vector(d, 2.0)
        cases.append({"name": f"euclidean_various_dim_{d}", "src": "internal/simd/simd_test.go:105-121",
                      "op": "pair", "metric": "euclidean", "gen": {"kind": "makeTestVector", "dim": d, "seeds": [1.0, 2.0]},
                      "expected": float(onp.euclidean(a, b)[0]), "rel_tol": 1e-3})
        cases.append({"name": f"cosine_various_dim_{d}", "src": "internal/simd/simd_test.go:196-213",
                      "op": "pair", "metric": "cosine", "gen": {"kind": "makeTestVector", "dim": d, "seeds": [1.0, 2.0]},
                      "expected": float(onp.cosine(a, b)[0]), "rel_tol": 1e-3})
    for d in [1, 7, 8, 15, 16, 31, 32, 64, 128, 256, 512, 1024]:
        a = make_test_vector(d, 1.0)
        cases.append({"name": f"dot_various_dim_{d}", "src": "internal/simd/simd_test.go:238-260",
                      "op": "pair", "metric": "dot_raw", "gen": {"kind": "makeTestVector", "dim": d, "seeds": [1.0, 1.0]},
                      "expected": float(onp.dot(a, a)[0]), "rel_tol": 1e-3})

    # (2)/(3) literals
    lit = [
        ("euclidean_basic", "simd_test.go:69-80", "euclidean", [1, 2, 3, 4], [5, 6, 7, 8], 8.0, 1e-5, False),
        ("euclidean_identical", "simd_test.go:82-91", "euclidean", [1, 2, 3, 4, 5, 6, 7, 8], [1, 2, 3, 4, 5, 6, 7, 8], 0.0, 0.0, True),
        ("euclidean_zeros128", "simd_test.go:93-103", "euclidean", [0] * 128, [0] * 128, 0.0, 0.0, True),
        ("cosine_identical", "simd_test.go:146-155", "cosine", [1, 2, 3, 4, 5, 6, 7, 8], [1, 2, 3, 4, 5, 6, 7, 8], 0.0, 1e-5, False),
        ("cosine_orthogonal", "simd_test.go:157-168", "cosine", [1, 0, 0, 0], [0, 1, 0, 0], 1.0, 1e-5, False),
        ("cosine_opposite", "simd_test.go:170-181", "cosine", [1, 2, 3, 4], [-1, -2, -3, -4], 2.0, 1e-5, False),
        ("cosine_zero_vector", "simd_test.go:183-194", "cosine", [0, 0, 0, 0], [1, 2, 3, 4], 1.0, 0.0, True),
        ("dot_basic", "simd_test.go:224-236", "dot_raw", [1, 2, 3, 4], [5, 6, 7, 8], 70.0, 1e-5, False),
        ("docs_l2_sqrt27", "docs/distance_metrics.md:21", "euclidean", [1, 2, 3], [4, 5, 6], float(np.sqrt(27.0)), 1e-6, False),
        ("docs_cosine_orthogonal", "docs/distance_metrics.md:38", "cosine", [1, 0], [0, 1], 1.0, 0.0, True),
        ("docs_dot_neg11", "docs/distance_metrics.md:55", "dot_neg", [1, 2], [3, 4], -11.0, 0.0, True),
    ]
    for name, src, metric, a, b, exp, tol, exact in lit:
        cases.append({"name": name, "src": src if src.startswith("docs") else "internal/simd/" + src,
                      "op": "pair", "metric": metric, "a": a, "b": b, "expected": exp,
                      "rel_tol": tol, "exact": exact})
    # accumulator independence (parallel_reduction_test.go:168-190)
    a = [0.0] * 16
    for i in (0, 5, 10, 15):
        a[i] = 1.0
    cases.append({"name": "dot_accumulator_independence", "src": "internal/simd/parallel_reduction_test.go:168-190",
                  "op": "pair", "metric": "dot_raw", "a": a, "b": [1.0] * 16, "expected": 4.0,
                  "rel_tol": 1e-6, "order": "unroll4"})

    # (4) 8-dim batch literals (parallel_reduction_test.go:13-68)
    q8 = [1, 2, 3, 4, 5, 6, 7, 8]
    vc = [[8, 7, 6, 5, 4, 3, 2, 1], [1] * 8, [0, 0, 0, 0, 0, 0, 0, 1], [1, 2, 3, 4, 5, 6, 7, 8]]
    vd = [[8, 7, 6, 5, 4, 3, 2, 1], [1] * 8, [0, 0, 0, 0, 0, 0, 0, 1], [2] * 8]
    cases.append({"name": "cosine_batch_8dim", "src": "internal/simd/parallel_reduction_test.go:13-40",
                  "op": "batch", "metric": "cosine", "query": q8, "vectors": vc,
                  "expected": fl(onp.cosine(q8, vc)), "abs_tol": 1e-4})
    cases.append({"name": "dot_batch_8dim", "src": "internal/simd/parallel_reduction_test.go:42-68",
                  "op": "batch", "metric": "dot_raw", "query": q8, "vectors": vd,
                  "expected": fl(onp.dot(q8, vd)), "abs_tol": 1e-4})
    cases.append({"name": "cosine_batch_dispatch", "src": "internal/simd/parallel_reduction_test.go:71-91",
                  "op": "batch", "metric": "cosine", "query": [1, 0, 0, 0],
                  "vectors": [[1, 0, 0, 0], [0, 1, 0, 0]], "expected": [0.0, 1.0], "abs_tol": 1e-4})
    cases.append({"name": "dot_batch_dispatch", "src": "internal/simd/parallel_reduction_test.go:93-113",
                  "op": "batch", "metric": "dot_raw", "query": [1, 2, 3, 4],
                  "vectors": [[1, 0, 0, 0], [0, 1, 0, 0], [1, 1, 1, 1]], "expected": [1.0, 2.0, 10.0], "abs_tol": 1e-4})

    # (5) 768-dim, 10-vector batch, all three metrics (parallel_reduction_test.go:116-165)
    dim = 768
    q = (np.arange(dim) % 10).astype(F) / F(10.0)
    V = np.stack([((np.arange(dim) + j) % 10).astype(F) / F(10.0) for j in range(10)])
    cases.append({"name": "batch_768_highdim", "src": "internal/simd/parallel_reduction_test.go:116-165",
                  "op": "batch3", "gen": {"kind": "mod10", "dim": dim, "nvec": 10},
                  "expected": {"euclidean_unroll4": fl(onp.euclidean(q, V, "unroll4")),
                               "euclidean_seq": fl(onp.euclidean(q, V, "seq")),
                               "cosine": fl(onp.cosine(q, V)), "dot_raw": fl(onp.dot(q, V))},
                  "abs_tol": 1e-3})

    # (7) brute-force fixtures (internal/store/adaptive_index_test.go:105-164,280-315)
    def ds4(n):
        return (np.arange(n * 4, dtype=np.int64).astype(F) * F(0.01)).reshape(n, 4)
    X = ds4(100)
    qv = np.array([0.1, 0.2, 0.3, 0.4], F)
    ids, dist = onp.topk_canonical(onp.euclidean(qv, X), 10)
    cases.append({"name": "bruteforce_100x4_k10", "src": "internal/store/adaptive_index_test.go:105-131,280-315",
                  "op": "search", "metric": "euclidean", "gen": {"kind": "ds4", "n": 100},
                  "query": fl(qv), "k": 10, "expect_count": 10, "expected_ids": [int(i) for i in ids],
                  "expected_dist": fl(dist)})
    cases.append({"name": "bruteforce_5x4_k100", "src": "internal/store/adaptive_index_test.go:133-152",
                  "op": "search", "metric": "euclidean", "gen": {"kind": "ds4", "n": 5},
                  "query": [1.0, 0.0, 0.0, 0.0], "k": 100, "expect_count": 5})
    cases.append({"name": "bruteforce_empty", "src": "internal/store/adaptive_index_test.go:154-164",
                  "op": "search", "metric": "euclidean", "gen": {"kind": "ds4", "n": 0},
                  "query": [1.0, 2.0, 3.0, 4.0], "k": 10, "expect_count": 0})
    # 500x768 v[i][j]=i+j, q[j]=j (adaptive_index_zerocopy_test.go:135-180): nearest is row 0, dist 0
    cases.append({"name": "bruteforce_500x768_ramp", "src": "internal/store/adaptive_index_zerocopy_test.go:135-180",
                  "op": "search", "metric": "euclidean", "gen": {"kind": "ramp", "n": 500, "dim": 768},
                  "k": 10, "expect_count": 10, "expected_ids": list(range(10)),
                  "expected_dist": fl(onp.euclidean(np.arange(768, dtype=F),
                                                    (np.arange(10)[:, None] + np.arange(768)[None, :]).astype(F)))})

    # (8) gpu.Index tests (internal/gpu/gpu_test.go:24-46, :57-83)
    cases.append({"name": "gpu_index_basic", "src": "internal/gpu/gpu_test.go:12-46",
                  "op": "gpu_index", "gen": {"kind": "flat_scaled", "n": 10, "dim": 128, "scale": 0.01},
                  "k": 5, "expect_first_id": 0, "expect_first_dist_lt": 0.01})
    cases.append({"name": "gpu_index_bench_fixture", "src": "internal/gpu/gpu_test.go:57-83",
                  "op": "gpu_index", "gen": {"kind": "flat_scaled", "n": 10000, "dim": 128, "scale": 0.001},
                  "k": 10, "expect_first_id": 0, "expect_first_dist_lt": 1e-6})

    # (10) merge (internal/store/result_merger_test.go:10-83)
    cases.append({"name": "merge_three_streams", "src": "internal/store/result_merger_test.go:10-45",
                  "op": "merge", "k": 10,
                  "lists": [{"ids": [1, 2, 3], "scores": [0.1, 0.4, 0.7]},
                            {"ids": [4, 5], "scores": [0.2, 0.5]},
                            {"ids": [6, 7, 8], "scores": [0.3, 0.6, 0.9]}],
                  "expected_ids": [1, 4, 6, 2, 5, 7, 3, 8]})
    cases.append({"name": "merge_empty_channel", "src": "internal/store/result_merger_test.go:47-65",
                  "op": "merge", "k": 5, "lists": [{"ids": [], "scores": []}, {"ids": [1], "scores": [0.5]}],
                  "expected_ids": [1]})
    cases.append({"name": "merge_limit_k", "src": "internal/store/result_merger_test.go:67-83",
                  "op": "merge", "k": 2, "lists": [{"ids": [1, 2, 3], "scores": [0.1, 0.2, 0.3]}],
                  "expected_ids": [1, 2]})

    # (11) FNV-1a-32 standard vectors (hash/fnv; the ring's only arithmetic) -- published test vectors
    cases.append({"name": "fnv1a32_vectors", "src": "Go hash/fnv (FNV-1a 32 standard test vectors)",
                  "op": "fnv", "inputs": ["", "a", "foobar"],
                  "expected": [0x811C9DC5, 0xE40C292C, 0xBF9CF968]})

    # RRF literals (internal/store/hybrid_search_test.go:75-150): A=0 must rank first; expected
    # scores by the formula in the test's own comment, evaluated in f64 and narrowed
    d, sp = [0, 1, 2, 3], [2, 0, 4, 1]
    sc = {}
    for lst in (d, sp):
        for r, i in enumerate(lst):
            sc[i] = sc.get(i, 0.0) + 1.0 / float(60 + r + 1)
    order = sorted(sc, key=lambda i: (-np.float32(sc[i]), i))
    cases.append({"name": "rrf_basic_fusion", "src": "internal/store/hybrid_search_test.go:75-108",
                  "op": "rrf", "dense": d, "sparse": sp, "k": 60, "limit": 10, "expected_ids": order,
                  "expected_scores": [float(np.float32(sc[i])) for i in order], "expect_top": 0})
    cases.append({"name": "rrf_one_empty", "src": "internal/store/hybrid_search_test.go:111-124",
                  "op": "rrf", "dense": [0], "sparse": [], "k": 60, "limit": 10, "expected_ids": [0],
                  "expected_scores": [float(np.float32(1.0 / 61.0))], "expect_top": 0})
    cases.append({"name": "rrf_k1", "src": "internal/store/hybrid_search_test.go:126-150",
                  "op": "rrf", "dense": [0, 1], "sparse": [1, 0], "k": 1, "limit": 10, "expected_ids": [0, 1],
                  "expected_scores": [float(np.float32(0.5 + 1.0 / 3.0))] * 2, "expect_top": None})

    # calculateAdaptiveLimit table (internal/store/adaptive_search_test.go:7-72): (k, matches, total) -> want
    cases.append({"name": "adaptive_limit_table", "src": "internal/store/adaptive_search_test.go:7-72",
                  "op": "adaptive_limit",
                  "rows": [[10, 1000, 1000, 20], [10, 500, 1000, 20], [10, 100, 1000, 100], [10, 20, 1000, 500],
                           [10, 10, 1000, 500], [10, 0, 1000, 10], [10, 0, 0, 10], [100, 50, 200, 200]]})

    # pack/unpack 8 bytes <-> 2 floats (internal/store/hnsw_pq_test.go:37-70): layout only
    cases.append({"name": "pq_pack_8bytes", "src": "internal/store/hnsw_pq_test.go:61-69",
                  "op": "pack", "bytes": [1, 2, 3, 4, 250, 251, 252, 253], "n_floats": 2})

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "cases": cases}, f, indent=1)
    print(f"wrote {len(cases)} cases to {OUT} ({os.path.getsize(OUT)} bytes)")


if __name__ == "__main__":
    main()
