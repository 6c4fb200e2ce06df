"""Randomised parity sweep over shapes the fixed cases do not pin: odd dimensions, tiny and large k, corpora on
both sides of the sampled-threshold limit (64k rows), every batch-size regime, both accumulation orders, ids,
predicate masks on either side of the compaction limit.  Seeds are fixed: failures reproduce."""
import os

import numpy as np
import pytest

from tests.gpu_util import F, assert_same, gpu_or_skip, new_index

pytestmark = pytest.mark.gpu

CASES = [
    # (seed, rows, dim, k, batch sizes)
    (1, 1, 7, 5, (1, 6)),
    (2, 63, 1, 10, (1, 20)),
    (3, 300, 33, 1, (3, 40)),
    (4, 5000, 100, 100, (1, 9, 70)),
    (5, 65535, 31, 10, (2, 33)),
    (6, 65536, 32, 10, (1, 8, 33, 140)),
    (7, 70000, 129, 37, (4, 5, 64, 65)),
    (8, 120000, 64, 512, (1, 20)),
    (9, 200000, 48, 1000, (2, 12)),
    (10, 90000, 768, 100, (1, 8, 48)),
    (11, 131072, 96, 3, (1, 17, 400)),
    (12, 3_000_000, 16, 10, (1, 6, 40)),      # beyond one sampled span: the growing chunks continue
    (13, 2_700_000, 8, 100, (2, 130)),
    (14, 100_000, 257, 20, (1, 7, 30)),
    (15, 600_001, 32, 40, (130, 300)),         # enough 256-row tiles for the tall split tile on the default path
]


@pytest.mark.parametrize("seed,n,d,k,batches", CASES)
def test_random_shapes(oracle, seed, n, d, k, batches):
    gpu_or_skip()
    rng = np.random.default_rng(seed + 1000 * int(os.environ.get("LB_FUZZ_ROUND", "0")))  # extra rounds: other data
    X = rng.standard_normal((n, d)).astype(F) if seed % 2 else rng.random((n, d), dtype=F)
    Q = rng.standard_normal((max(batches), d)).astype(F) if seed % 2 else rng.random((max(batches), d), dtype=F)
    if n > 10:
        Q[0] = X[n // 2]                       # an exact hit
        X[n // 3] = 0                          # a zero row (cosine -> distance 1)
    ids = rng.permutation(n).astype(np.int64) * 7 + 3 if seed % 3 == 0 else None
    meta = rng.integers(0, 100, n)
    for metric in (0, 1, 2):
        order = (seed + metric) % 2
        idx = new_index(d, metric, order)
        idx.Add(ids, X)
        for mask in (None, (meta < 97).astype(np.uint8), (meta < 30).astype(np.uint8)):
            idx.set_filter(mask)
            for nq in batches:
                lab, dist = idx.SearchBatch(Q[:nq], k)
                oi, od = oracle.search_batch(metric, Q[:nq], X, k, order=order, mask=mask, ids=ids, nthreads=8)
                assert_same(lab, dist, oi, od, f"seed {seed} metric {metric} order {order} nq {nq} "
                                               f"mask {'none' if mask is None else int(mask.sum())}")
        idx.Close()
