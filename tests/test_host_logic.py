"""Host-side logic of the product (no GPU): RingSharder, metric enum, blob layout."""
import struct

import numpy as np
import pytest

from longbow_amd import pq as lpq
from longbow_amd.sharded import RingSharder, fnv1a32_bytes, partition_rows
from longbow_amd.simd import MetricType


def test_metric_type_values_and_names():
    """internal/simd/registry.go:8-29"""
    assert [int(m) for m in MetricType] == [0, 1, 2]
    assert [str(m) for m in MetricType] == ["euclidean", "cosine", "dot"]


def test_ring_sharder_matches_oracle(oracle):
    ring = RingSharder(8, 40)
    oring = oracle.Ring(8, 40)
    assert np.array_equal(ring.sorted_hashes, oring.hashes)
    assert np.array_equal(ring.owners, oring.owners)
    ids = np.concatenate([np.arange(5000, dtype=np.uint64), np.array([2**32 - 1, 2**40 + 7, 2**63 + 11], np.uint64)])
    got = ring.GetShards(ids)
    exp = np.array([oring.get_shard(int(i)) for i in ids])
    assert np.array_equal(got, exp)
    assert ring.GetShard(12345) == oring.get_shard(12345)
    assert fnv1a32_bytes(b"foobar") == 0xBF9CF968
    assert RingSharder(3, 0).sorted_hashes.size == 60


def test_partition_skew_report():
    owner, counts = partition_rows(np.arange(200000, dtype=np.uint64), 8, 40)
    assert counts.sum() == 200000 and counts.min() > 0
    # FNV-1a-32 over the short "shard:vnode" keys disperses poorly: the reference's own ring puts
    # ~30 % of all ids on one of 8 shards (max/mean ~ 2.4).  Restated faithfully; see DESIGN.md.
    assert 2.0 < counts.max() / counts.mean() < 2.8


def test_codebook_blob_layout():
    """internal/pq/persistence.go:9-35"""
    cb = np.arange(2 * 256 * 4, dtype=np.float32).reshape(2, 256, 4)
    blob = lpq.serialize_codebooks(cb)
    assert struct.unpack("<III", blob[:12]) == (8, 2, 256)
    assert len(blob) == 12 + 2 * 256 * 4 * 4
    assert np.array_equal(np.frombuffer(blob[12:], "<f4").reshape(2, 256, 4), cb)


def test_calculate_adaptive_limit_matches_oracle(oracle):
    """host mirror of calculateAdaptiveLimit against the oracle over a grid (incl. the clamps)"""
    from longbow_amd import hybrid
    for k in (1, 10, 100):
        for total in (0, 1, 50, 1000, 10**6):
            for matches in (0, 1, 2, total // 50 if total else 0, total // 10 if total else 0, total):
                assert hybrid.calculate_adaptive_limit(k, matches, total) == oracle.adaptive_limit(k, matches, total)


def test_sample_plan_invariants():
    """the sampled-threshold plan (index.hip: sample_plan) over a grid of corpus sizes, k and list capacities:
    the sample fits one list, m is in [8, 64], the too-tight tail is below 1e-6 and mean + 5 sigma admitted
    rows fit the list -- checked on the host, no GPU needed"""
    import ctypes as C
    import math
    from longbow_amd import _lib
    try:
        lib = _lib.load_diag()  # host-logic probes are exported by the diagnostic build only
    except (RuntimeError, OSError) as e:  # CPU-only checkout without the built library / ROCm runtime
        pytest.skip(f"liblongbow_gpu_diag.so not loadable here: {e}")
    out = (C.c_longlong * 4)()
    seen_on = 0
    for n in (1000, 16383, 16384, 20000, 65535, 65536, 10**5, 10**6, 2_500_000, 10**7, 10**8, 4 * 10**9):
        for keep in (1, 10, 100, 256, 512, 1024, 4096):
            for cap in (8192, 16384, 32768):
                for count_max in (4096, 8192):
                    if keep > cap:
                        continue
                    lib.lb_debug_sample_plan(n, keep, cap, count_max, out)
                    on, span, count, m = out[0], out[1], out[2], out[3]
                    if not on:
                        continue
                    seen_on += 1
                    assert n >= 16384 and 0 < span <= n  # (sampled thresholds start at 16,384 rows since round 4: 65,536 before)
                    assert count <= min(cap, count_max) and span >= 8 * count
                    assert 8 <= m <= 64
                    lam = keep * count / span
                    # P(Poisson(lam) >= m): the threshold admits fewer than `keep` rows
                    tail = 1.0 - sum(math.exp(-lam) * lam ** i / math.factorial(i) for i in range(m))
                    assert tail < 1e-6, (n, keep, cap, m, lam, tail)
                    mean = m * span / count
                    assert mean * (1 + 5 / math.sqrt(m)) <= (cap - keep) * 1.0001, (n, keep, cap, m, mean)
    assert seen_on > 50


def test_long_result_lists_get_one_sampled_span():
    """k = 241 .. 512 keeps 1,024 candidates, twice that on the fp16 routes.  With the 8,192-entry lists of rounds 1-3 the
    sampled span of such a search ended short of a 1M-row corpus (the rest ran the classic schedule: 0.71 ms a query where
    k = 100 took 0.29); with the 16,384-entry lists the index allots from 1,024 candidates one span covers it."""
    import ctypes as C
    from longbow_amd import _lib
    try:
        lib = _lib.load_diag()
    except (RuntimeError, OSError) as e:
        pytest.skip(f"liblongbow_gpu_diag.so not loadable here: {e}")
    out = (C.c_longlong * 4)()
    lib.lb_debug_sample_plan(1_000_000, 2048, 8192, 8192, out)
    assert not out[0] or out[1] < 1_000_000      # the old geometry: no plan, or a span that ends short
    lib.lb_debug_sample_plan(1_000_000, 2048, 16384, 8192, out)
    assert out[0] and out[1] >= 1_000_000        # the new one: a single span


def test_gpu_partition_packs_ring_shards_evenly():
    """GpuPartition: the reference's RingSharder with 8 ring shards per GPU, shards packed onto GPUs by the
    share of the hash space they own.  The id -> ring shard map stays the reference's; only shard -> GPU is
    chosen; the expected shares (from the ring's arcs) match a count over real ids."""
    from longbow_amd.sharded import GpuPartition
    ids = np.arange(2_000_000, dtype=np.uint64)
    for gpus in (1, 2, 4, 8):
        p = GpuPartition(gpus, 8, 40)
        assert p.ring.num_shards == gpus * 8
        g = p.GetGpus(ids)
        assert g.min() >= 0 and g.max() < gpus
        assert np.array_equal(g, p.shard_gpu[RingSharder(gpus * 8, 40).GetShards(ids)])
        cnt = np.bincount(g, minlength=gpus)
        assert cnt.max() / cnt.mean() < 1.03, (gpus, cnt)
        assert abs(p.skew() - cnt.max() / cnt.mean()) < 0.01
        assert abs(p.shard_fraction.sum() - 1.0) < 1e-9
    assert GpuPartition(8, 1, 40).skew() > 2.0  # one ring shard per GPU: the skew the packing removes


def test_comm_entry_points_validate_arguments_without_a_gpu():
    import ctypes as C
    from longbow_amd import _lib
    lib = _lib.load()
    st = C.c_int(0)
    assert not lib.lb_gpu_comm_init_all(0, None, C.byref(st)) and st.value == 1
    assert not lib.lb_gpu_comm_init_host(0, 2, 5, None, None, C.byref(st)) and st.value == 1
    assert not lib.lb_gpu_comm_init_rank(0, 2, 0, None, C.byref(st)) and st.value == 1
    if lib.lb_gpu_device_count() == 0:
        assert not lib.lb_gpu_comm_init_all(2, None, C.byref(st)) and st.value == 3
    assert lib.lb_gpu_comm_nranks(None) == 0
    assert lib.lb_gpu_comm_search_device(None, None, 1, None, 1, None, None, None) == 1


def test_kernel_registry_lookup_rule_and_metric_names():
    """internal/simd/registry.go:94-124: exact (metric, type, dims) first, then dims = 0; nil when neither;
    core.DistanceMetric strings (core/enums.go:6-13) map onto MetricType; the HIP batch kernels sit under
    their own dims key and never shadow the per-pair kernels DispatchDistance looks up"""
    from longbow_amd import simd
    r = simd.KernelRegistry()
    f0, f128 = object(), object()
    r.Register(simd.MetricType.Euclidean, simd.SIMDDataType.Float32, 0, f0)
    r.Register(simd.MetricType.Euclidean, simd.SIMDDataType.Float32, 128, f128)
    assert r.Get(simd.MetricType.Euclidean, simd.SIMDDataType.Float32, 128) is f128
    assert r.Get(simd.MetricType.Euclidean, simd.SIMDDataType.Float32, 384) is f0
    assert r.Get(simd.MetricType.Cosine, simd.SIMDDataType.Float32, 128) is None
    assert r.Get(simd.MetricType.Euclidean, simd.SIMDDataType.Float16, 128) is None
    r.Register(simd.MetricType.Euclidean, simd.SIMDDataType.Float32, 128, f0)  # re-registering replaces
    assert r.Get(simd.MetricType.Euclidean, simd.SIMDDataType.Float32, 128) is f0
    for m in simd.MetricType:
        batch = simd.Registry.Get(m, simd.SIMDDataType.Float32, simd.BatchFlatDims)
        pair = simd.Registry.Get(m, simd.SIMDDataType.Float32, 768)
        assert batch is not None and pair is not None and batch is not pair and batch.metric == m
    assert [str(m) for m in simd.MetricType] == ["euclidean", "cosine", "dot"]
    assert str(simd.SIMDDataType.Float32) == "float32" and int(simd.SIMDDataType.Complex128) == 12
    assert simd.MetricFromCore("euclidean") == 0 and simd.MetricFromCore("cosine") == 1
    assert simd.MetricFromCore("dot_product") == 2 and simd.MetricFromCore("dot") == 2
    with pytest.raises(ValueError):
        simd.MetricFromCore("manhattan")
    with pytest.raises(ValueError):
        simd.DispatchDistance(0, np.zeros(3, np.float32), np.zeros(4, np.float32))  # dimension mismatch
    assert simd.DispatchDistance(1, np.zeros(0, np.float32), np.zeros(0, np.float32)) == 0  # len 0 -> 0, nil
