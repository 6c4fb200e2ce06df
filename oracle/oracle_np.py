"""Independent numpy-f32 restatement of the reference formulas.

TEST INFRASTRUCTURE ONLY.  Written separately from longbow_oracle.c (different
language, vectorised over rows instead of over dims) so that the two can check
each other; tests/test_oracle.py requires bit-equality between them.

Every arithmetic step is done on np.float32 arrays, so each operation rounds
once to binary32 exactly like the Go source on amd64 (no FMA).
"""
import numpy as np

F = np.float32


def _rows(x):
    x = np.asarray(x, dtype=F)
    return x[None, :] if x.ndim == 1 else x


def l2sq_seq(q, X):
    """referenceEuclidean before sqrt (internal/simd/simd_test.go:13-20)."""
    q, X = np.asarray(q, F), _rows(X)
    s = np.zeros(X.shape[0], F)
    for i in range(X.shape[1]):
        d = q[i] - X[:, i]
        s = s + d * d
    return s


def l2sq_unroll4(q, X):
    """L2SquaredFloat32 (internal/simd/distance_functions.go:195-227)."""
    q, X = np.asarray(q, F), _rows(X)
    n = X.shape[1]
    acc = [np.zeros(X.shape[0], F) for _ in range(4)]
    i = 0
    while i <= n - 4:
        for u in range(4):
            d = q[i + u] - X[:, i + u]
            acc[u] = acc[u] + d * d
        i += 4
    while i < n:
        d = q[i] - X[:, i]
        acc[0] = acc[0] + d * d
        i += 1
    return ((acc[0] + acc[1]) + acc[2]) + acc[3]


def _sqrt64(s):
    return np.sqrt(s.astype(np.float64)).astype(F)


def euclidean(q, X, order="seq"):
    return _sqrt64(l2sq_seq(q, X) if order == "seq" else l2sq_unroll4(q, X))


def _sums3(q, X, order):
    q, X = np.asarray(q, F), _rows(X)
    n = X.shape[1]
    m = X.shape[0]
    if order == "seq":
        dot = np.zeros(m, F); na = np.zeros(m, F); nb = np.zeros(m, F)
        for i in range(n):
            dot = dot + q[i] * X[:, i]
            na = na + q[i] * q[i]
            nb = nb + X[:, i] * X[:, i]
        return dot, na, nb
    d = [np.zeros(m, F) for _ in range(4)]
    a = [np.zeros(m, F) for _ in range(4)]
    b = [np.zeros(m, F) for _ in range(4)]
    i = 0
    while i <= n - 4:
        for u in range(4):
            d[u] = d[u] + q[i + u] * X[:, i + u]
            a[u] = a[u] + q[i + u] * q[i + u]
            b[u] = b[u] + X[:, i + u] * X[:, i + u]
        i += 4
    while i < n:
        d[0] = d[0] + q[i] * X[:, i]
        a[0] = a[0] + q[i] * q[i]
        b[0] = b[0] + X[:, i] * X[:, i]
        i += 1
    red = lambda v: ((v[0] + v[1]) + v[2]) + v[3]
    return red(d), red(a), red(b)


def cosine(q, X, order="seq"):
    """referenceCosine (simd_test.go:22-33) / cosineUnrolled4x (simd.go:399-450)."""
    X = _rows(X)
    if X.shape[1] == 0:
        return np.ones(X.shape[0], F)
    dot, na, nb = _sums3(q, X, order)
    denom = np.sqrt(na.astype(np.float64) * nb.astype(np.float64)).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = F(1.0) - dot / denom
    out = np.where((na == 0) | (nb == 0), F(1.0), out).astype(F)
    return out


def dot(q, X, order="seq"):
    d, _, _ = _sums3(q, X, order)
    return d


def distance(metric, q, X, order="seq"):
    if metric == 0:
        return euclidean(q, X, order)
    if metric == 1:
        return cosine(q, X, order)
    return -dot(q, X, order)


def topk_canonical(dist, k):
    """ascending (distance, index)"""
    dist = np.asarray(dist, F)
    idx = np.lexsort((np.arange(dist.size), dist))[:k]
    return idx.astype(np.int64), dist[idx]


def adc_batch(table, codes):
    """adcBatchGeneric (internal/simd/simd.go:345-355)"""
    table = np.asarray(table, F)
    codes = np.asarray(codes, np.uint8)
    s = np.zeros(codes.shape[0], F)
    for j in range(codes.shape[1]):
        s = s + table[j * 256 + codes[:, j].astype(np.int64)]
    return _sqrt64(s)


def build_adc_table(codebooks, q):
    """BuildADCTable (internal/pq/adc_table.go:15-51); codebooks [M][K][sub]."""
    cb = np.asarray(codebooks, F)
    M, K, sub = cb.shape
    q = np.asarray(q, F)
    t = np.empty((M, K), F)
    for i in range(M):
        t[i] = l2sq_unroll4(q[i * sub:(i + 1) * sub], cb[i])
    return t.reshape(-1)


def fnv1a32(data: bytes) -> int:
    h = 2166136261
    for b in data:
        h ^= b
        h = (h * 16777619) & 0xFFFFFFFF
    return h


def ring_points(num_shards, vnodes=40):
    """RingSharder construction (sharding_strategy.go:49-83)."""
    ring = {}
    hashes = []
    for s in range(num_shards):
        for v in range(vnodes):
            h = fnv1a32(f"{s}:{v}".encode())
            ring[h] = s
            hashes.append(h)
    hashes.sort()
    return hashes, ring


def ring_get_shard(hashes, ring, vid):
    import bisect
    h = fnv1a32(int(vid).to_bytes(8, "little"))
    i = bisect.bisect_left(hashes, h)
    if i == len(hashes):
        i = 0
    return ring[hashes[i]]


def splitmix64_uniform(n, seed, offset=0):
    idx = (np.arange(n, dtype=np.uint64) + np.uint64(offset) + np.uint64(1))
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32) * F(1.0 / 16777216.0)).astype(F)


def adaptive_limit(k, matches, total):
    """calculateAdaptiveLimit (internal/store/adaptive_search.go:7-39), independent restatement"""
    if total == 0 or matches == 0:
        return k
    factor = min(max(1.0 / (float(matches) / float(total)), 2.0), 50.0)
    return max(min(int(float(k) * factor), total), k)
