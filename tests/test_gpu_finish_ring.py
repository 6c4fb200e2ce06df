"""The finish launch's tiled form (17 .. 256 queries: the member rows arrive by LDS-DMA into a ring of [256 rows][128 B]
stages, kernels_finish.hip) at its edges, against the CPU oracle bit for bit (BruteForceIndex.SearchVectors' ranking,
internal/store/adaptive_index.go:161-225; the metrics of internal/simd/simd.go:131-163,365-479):

* more members than one group of 256 (k beyond 256: the ring runs on across the groups);
* a ring of three and of two stages (k beyond 512 / 1024: the member arrays take the LDS of the others);
* dimensions that are not a multiple of 32 (the last chunk of a row is partial: its pieces beyond the row are clamped and
  never walked), dimensions below one chunk, and a dimension that is not a multiple of 4 (the generic walk);
* both accumulation orders, every metric, user ids, a filtered view (positions mapped back to rows);
* batches on both sides of the 256-query limit (beyond it the register-staged tile serves);
* 17 .. 128 queries with k x D from 128 Ki: several workgroups per query share its members through the ring and hand
  their exact values to the last one to arrive.
"""
import numpy as np
import pytest

from tests.gpu_util import assert_same, gpu_or_skip, new_index

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d,nq,k,order", [
    (768, 64, 300, 0),    # two groups of members, whole chunks
    (768, 17, 257, 1),    # the smallest tiled batch, one member beyond a group, four accumulators
    (100, 40, 280, 0),    # last chunk partial (100 = 3 x 32 + 4)
    (36, 33, 100, 1),     # one whole chunk + one piece
    (20, 24, 64, 0),      # less than one chunk
    (96, 256, 600, 0),    # smax 2048: three stages
    (64, 48, 1100, 0),    # smax 4096: two stages, 64 list entries per thread
    (70, 20, 50, 0),      # not a multiple of 4: generic walk
    (128, 257, 120, 0),   # beyond 256 queries: the register-staged tile
    (768, 24, 200, 0),    # k x D beyond 128 Ki at 17 .. 32 queries: eight workgroups per query share the members (split + ring)
    (1536, 40, 100, 1),   # ... four per query at 33 .. 64
    (1536, 100, 100, 0),  # ... two at 65 .. 128
])
def test_tiled_finish_matches_oracle(oracle, metric, d, nq, k, order):
    gpu_or_skip()
    n = 24000
    rng = np.random.default_rng(d * 131 + nq * 7 + k + metric)
    X = rng.random((n, d), dtype=F) - F(0.3)
    Q = rng.random((nq, d), dtype=F) - F(0.3)
    idx = new_index(d, metric, order)
    idx.Add(None, X)
    lab, dist = idx.SearchBatch(Q, k)
    oi, od = oracle.search_batch(metric, Q, X, k, order=order, nthreads=8)
    assert_same(lab, dist, oi, od, f"metric={metric} d={d} nq={nq} k={k} order={order}")
    idx.Close()


@pytest.mark.parametrize("metric", [0, 2])
def test_tiled_finish_with_ids_and_a_filtered_view(oracle, metric):
    """user ids in the labels; a selective predicate (the persistent kernels leave positions of the row list in the entries,
    the finish maps them back) with more than 256 members per query"""
    gpu_or_skip()
    n, d, nq, k = 300000, 512, 48, 260  # (k x D beyond 128 Ki: several workgroups per query)
    rng = np.random.default_rng(77 + metric)
    X = rng.random((n, d), dtype=F)
    Q = rng.random((nq, d), dtype=F)
    ids = (np.arange(n, dtype=np.int64) * 3 + 5)
    meta = rng.integers(0, 100, n).astype(np.int64)
    idx = new_index(d, metric)
    idx.Add(ids, X)
    idx.filter_column(meta, "<", 30)
    lab, dist = idx.SearchBatch(Q, k)
    vis = np.nonzero(meta < 30)[0]
    oi, od = oracle.search_batch(metric, Q, X[vis], k, nthreads=8)
    want = np.where(oi >= 0, ids[vis[np.clip(oi, 0, len(vis) - 1)]], -1)
    assert_same(lab, dist, want, od, f"filtered metric={metric}")
    idx.Close()
