"""Hybrid dense+sparse fusion on the GPU.

ReciprocalRankFusion mirrors store.ReciprocalRankFusion (internal/store/rrf.go:10-51): inputs are two
ranked result lists (best first), k defaults to 60, limit <= 0 means "all"; the sparse (BM25) ranking itself
stays on the CPU side as in the reference.  search_hybrid_candidates is the GPU -> HNSW hand-off rule of
ArrowHNSW.SearchHybrid (internal/store/hnsw_gpu.go:84-123): ask the GPU for min(10*k, Len) candidates and keep
the first k that are still live.
"""
import numpy as np

from . import _lib


def ReciprocalRankFusion(dense, sparse, k=60, limit=0, device=0):
    """dense / sparse: sequences of ids (or (id, score) pairs), best first.  Returns (ids, scores)."""
    def ids_of(lst):
        if lst is None or len(lst) == 0:
            return np.empty(0, np.int64)
        a = np.asarray([x[0] if isinstance(x, (tuple, list)) else x for x in lst], np.int64)
        return np.ascontiguousarray(a)
    d, s = ids_of(dense), ids_of(sparse)
    if d.size + s.size == 0:
        return np.empty(0, np.int64), np.empty(0, np.float32)  # rrf.go:30-32 returns nil
    lim = limit if limit > 0 else d.size + s.size
    ids, sc = fuse_batch(d[None, :], s[None, :], k, lim, device)
    keep = ids[0] >= 0
    return ids[0][keep], sc[0][keep]


def fuse_batch(dense_ids, sparse_ids, k=60, limit=10, device=0):
    """nq queries at once: dense_ids [nq, kd], sparse_ids [nq, ks] (int64, -1 padding) -> ([nq, limit], [nq, limit])"""
    lib = _lib.require_gpu(device)
    d = np.ascontiguousarray(dense_ids, np.int64)
    s = np.ascontiguousarray(sparse_ids, np.int64)
    nq = d.shape[0]
    out_i = np.empty((nq, limit), np.int64)
    out_s = np.empty((nq, limit), np.float32)
    _lib.check(lib.lb_gpu_rrf_fuse(device, nq, d.shape[1], d.ctypes.data if d.size else None, s.shape[1],
                                   s.ctypes.data if s.size else None, k, limit, out_i.ctypes.data, out_s.ctypes.data))
    return out_i, out_s


def search_hybrid_candidates(index, query, k, is_live=None):
    """candidateCount = min(k*10, Len); first k live candidates, GPU order trusted (hnsw_gpu.go:84-123)"""
    count = min(k * 10, index.ntotal)
    if count <= 0:
        return np.empty(0, np.int64), np.empty(0, np.float32)
    ids, dist = index.Search(query, count)
    keep = ids >= 0
    if is_live is not None:
        keep &= np.array([bool(is_live(int(i))) if i >= 0 else False for i in ids])
    ids, dist = ids[keep][:k], dist[keep][:k]
    return ids, dist


def calculate_adaptive_limit(k, matches, total):
    """calculateAdaptiveLimit (internal/store/adaptive_search.go:7-39): search depth for POST-filtered search,
    k * clamp(total / matches, 2, 50) clamped to [k, total].  The GPU path does not need it (predicates are
    applied inside the search: lb_gpu_index_set_filter / filter_int64), but callers that post-filter HNSW results
    keep the reference's rule."""
    if total == 0 or matches == 0:
        return int(k)
    factor = 1.0 / (float(matches) / float(total))
    factor = 50.0 if factor > 50.0 else factor
    factor = 2.0 if factor < 2.0 else factor
    limit = int(float(k) * factor)
    limit = total if limit > total else limit
    return int(k) if limit < k else int(limit)
