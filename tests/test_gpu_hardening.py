"""What a Go server links against: cancellation (the ctx of SearchVectors, internal/store/adaptive_index.go:182), the
slice-of-slices batch entry with the reference's per-vector rules (internal/simd/batch_operations.go:29-60,131-157),
argument caps that are checked before any staging is sized, the default candidate mode, and a communicator in which a
failing rank still takes part in the exchange."""
import os
import socket
import threading
import time

import numpy as np
import pytest

from tests.gpu_util import assert_same, gpu_or_skip, new_index

pytestmark = pytest.mark.gpu
F = np.float32
FLT_MAX = np.finfo(F).max


def test_default_mode_is_auto_and_large_batches_take_the_split_route(oracle):
    """a fresh index answers 512 / 1024-query batches on the split contraction (no f32-MFMA cliff beyond 384 queries) with
    results bit-equal to the strict mode and to the oracle; per-query cost must not jump between 384 and 512 queries"""
    gpu_or_skip()
    rng = np.random.default_rng(5)
    n, d, k = 200_000, 128, 30
    X = rng.random((n, d), dtype=F)
    Q = rng.random((1024, d), dtype=F)
    idx = new_index(d, 1)
    idx.Add(None, X)
    oi, od = oracle.search_batch(1, Q[:48], X, k, nthreads=8)
    per_query = {}
    res = {}
    for nq in (384, 512, 1024):
        lab, dist = idx.SearchBatch(Q[:nq], k)      # default = LB_CAND_AUTO
        assert_same(lab[:48], dist[:48], oi, od, f"auto nq={nq}")
        res[nq] = (lab, dist)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            idx.SearchBatch(Q[:nq], k)
            ts.append(time.perf_counter() - t0)
        per_query[nq] = sorted(ts)[2] / nq
    idx.set_candidate_mode(0)                        # strict: f32 MFMA beyond 384 queries
    for nq in (512, 1024):
        lab, dist = idx.SearchBatch(Q[:nq], k)
        assert np.array_equal(lab, res[nq][0]) and np.array_equal(dist, res[nq][1])
    assert per_query[512] < 1.5 * per_query[384], per_query
    assert per_query[1024] < 1.5 * per_query[384], per_query
    idx.Close()


def test_cancel_and_deadline(oracle):
    gpu_or_skip()
    from longbow_amd import gpu
    rng = np.random.default_rng(8)
    n, d, k = 300_000, 256, 10
    X = rng.random((n, d), dtype=F)
    Q = rng.random((2048, d), dtype=F)
    idx = new_index(d, 0)
    idx.Add(None, X)
    want = idx.SearchBatch(Q[:64], k)
    # fired before the call: nothing runs
    c = gpu.Cancel()
    c.fire()
    with pytest.raises(gpu.Canceled):
        idx.SearchBatch(Q[:64], k, ctx=c)
    assert c.state == 8
    # a deadline that has passed
    c2 = gpu.Cancel(deadline_ms=0)
    time.sleep(0.002)
    with pytest.raises(gpu.DeadlineExceeded):
        idx.Search(Q[0], k, ctx=c2)
    # a live context changes nothing
    c3 = gpu.Cancel(deadline_ms=60_000)
    lab, dist = idx.SearchBatch(Q[:64], k, ctx=c3)
    assert np.array_equal(lab, want[0]) and np.array_equal(dist, want[1])
    # fired from another thread while a long series of searches runs: the series ends early with Canceled, and
    # the index keeps answering (workspaces, fused-launch tickets and streams are left consistent)
    c4 = gpu.Cancel()
    threading.Timer(0.05, c4.fire).start()
    t0 = time.perf_counter()
    with pytest.raises(gpu.Canceled):
        for _ in range(2000):
            idx.SearchBatch(Q, k, ctx=c4)
            idx.SearchBatch(Q[:9], k, ctx=c4)        # (the fused launch)
    assert time.perf_counter() - t0 < 5.0
    for nq in (1, 9, 64):
        lab, dist = idx.SearchBatch(Q[:nq], k)
        assert np.array_equal(lab, want[0][:nq]) and np.array_equal(dist, want[1][:nq])
    for c_ in (c, c2, c3, c4):
        c_.close()
    idx.Close()


def test_cancel_token_fire_races_the_return(oracle):
    """The lifetime pattern of hip_gpu.go: SearchBatchContext (round-3 advisor, use-after-free): a watcher thread may fire
    the token at any moment around the return of the call; the token is freed only after the watcher has been joined.
    Many short searches, the watcher's fire placed right at the return: answers stay right, nothing crashes, and a token
    fired after the return has no effect on that call's results."""
    gpu_or_skip()
    from longbow_amd import gpu
    rng = np.random.default_rng(18)
    n, d, k = 20_000, 64, 5
    X = rng.random((n, d), dtype=F)
    q = rng.random((1, d), dtype=F)
    idx = new_index(d, 0)
    idx.Add(None, X)
    want = idx.SearchBatch(q, k)
    for it in range(200):
        c = gpu.Cancel(deadline_ms=60_000)
        returned = threading.Event()

        def watcher(c=c, returned=returned):
            returned.wait()
            c.fire()              # "defer cancel()" of the caller, racing the binding's clean-up

        th = threading.Thread(target=watcher)
        th.start()
        lab, dist = idx.SearchBatch(q, k, ctx=c)
        returned.set()
        th.join()                  # stop, JOIN, then free: the order hip_gpu.go keeps
        c.close()
        assert np.array_equal(lab, want[0]) and np.array_equal(dist, want[1]), it
    idx.Close()


def test_pq_search_cancel(oracle):
    gpu_or_skip()
    from longbow_amd import gpu, pq
    rng = np.random.default_rng(9)
    M, dims, n = 8, 64, 200_000
    cb = rng.random((M, 256, dims // M), dtype=F)
    enc = pq.PQEncoder(pq.serialize_codebooks(cb))
    enc.add_codes(rng.integers(0, 256, (n, M), dtype=np.uint8))
    Q = rng.random((6, dims), dtype=F)
    want = enc.Search(Q, 10)
    c = gpu.Cancel()
    lab, dist = enc.Search(Q, 10, ctx=c)
    assert np.array_equal(lab, want[0]) and np.array_equal(dist, want[1])
    c.fire()
    with pytest.raises(gpu.Canceled):
        enc.Search(Q, 10, ctx=c)
    lab, dist = enc.Search(Q, 10)
    assert np.array_equal(lab, want[0])
    c.close()
    enc.Close()


def test_k_is_checked_before_any_staging_is_sized():
    """k = INT32_MAX used to size a pinned slab and a device buffer of nq*k*12 bytes before being rejected"""
    lib = gpu_or_skip()
    idx = new_index(16, 0)
    idx.Add(None, np.zeros((10, 16), F))
    q = np.zeros((1, 16), F)
    out_d, out_l = np.zeros(4, F), np.zeros(4, np.int64)
    import torch
    free0, _ = torch.cuda.mem_get_info(0)
    rc = lib.lb_gpu_index_search(idx._h, 1, q.ctypes.data, 2**31 - 1, out_d.ctypes.data, out_l.ctypes.data)
    assert rc == 6  # LB_ERR_UNSUPPORTED
    rc = lib.lb_gpu_index_search(idx._h, 1, q.ctypes.data, 2049, out_d.ctypes.data, out_l.ctypes.data)
    assert rc == 6
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 < (64 << 20)
    idx.Close()


def test_concurrent_single_query_searches_are_combined_and_identical(oracle):
    """gpu.Index.Search is one query per call, from many goroutines (internal/gpu/faiss_gpu.go:108-145): calls that overlap are
    answered by ONE batched device search (index.hip: combined_search).  Every caller must get exactly what a search on its own
    returns -- same labels, same distance bits --, callers with another k are never mixed in, errors reach their caller only,
    and the counters show that batches were in fact combined."""
    gpu_or_skip()
    rng = np.random.default_rng(515)
    n, d = 300_000, 128
    X = rng.random((n, d), dtype=F)
    Q = rng.random((96, d), dtype=F)
    idx = new_index(d, 1)
    idx.Add(None, X)
    ks = (10, 37)
    idx.set_search_combining(False)
    want = {k: [idx.Search(Q[i], k) for i in range(len(Q))] for k in ks}
    oi, od = oracle.search_batch(1, Q[:4], X, ks[0], nthreads=8)
    for i in range(4):
        assert_same(want[ks[0]][i][0][None], want[ks[0]][i][1][None], oi[i][None], od[i][None], f"sequential search {i}")
    assert idx.combining_stats == (0, 0)
    idx.set_search_combining(True)
    errors = []

    def caller(t):
        try:
            k = ks[t % 2] if t >= 6 else ks[0]          # two of the eight callers ask for another k
            for rep in range(25):
                for i in range(t, len(Q), 8):
                    lab, dist = idx.Search(Q[i], k)
                    if not (np.array_equal(lab, want[k][i][0]) and np.array_equal(dist, want[k][i][1])):
                        errors.append(f"thread {t} query {i} k {k} rep {rep}: differs from the search on its own")
                if t == 3 and rep % 5 == 0:             # a bad call in the middle of the traffic fails alone
                    with pytest.raises(ValueError):
                        idx.Search(Q[0][: d - 1], k)
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {t}: {type(e).__name__}: {e}")

    ths = [threading.Thread(target=caller, args=(t,)) for t in range(8)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors[:5]
    batches, requests = idx.combining_stats
    assert batches > 0 and requests >= 2 * batches, (batches, requests)
    # a lone caller is never held back: nothing is combined when calls do not overlap
    before = idx.combining_stats
    for i in range(8):
        lab, dist = idx.Search(Q[i], ks[0])
        assert np.array_equal(lab, want[ks[0]][i][0]) and np.array_equal(dist, want[ks[0]][i][1])
    assert idx.combining_stats == before
    idx.Close()


@pytest.mark.parametrize("order", [0, 1])
def test_batch_over_slices_follows_the_reference_per_vector_rules(oracle, order):
    gpu_or_skip()
    from longbow_amd import simd
    rng = np.random.default_rng(21 + order)
    d = 48
    q = rng.standard_normal(d).astype(F)
    vecs = [rng.standard_normal(d).astype(F) for _ in range(9)]
    ragged = list(vecs)
    ragged[2] = None                                  # nil
    ragged[5] = rng.standard_normal(d - 1).astype(F)  # length mismatch
    full = np.stack(vecs)
    want = {m: oracle.batch_flat(m, q, full, order) for m in (0, 1, 2)}
    want[2] = -want[2]  # the oracle reports the value the search ranks by (negated); simd.DotProductBatch is raw
    # Euclidean: nil / mismatched -> math.MaxFloat32, everything else computed (batch_operations.go:39-42,51)
    r = np.full(9, -7.0, F)
    simd.EuclideanDistanceBatch(q, ragged, r, order=order)
    exp = want[0].copy()
    exp[[2, 5]] = FLT_MAX
    assert np.array_equal(r, exp)
    # Cosine / Dot: nil skipped (slot untouched); the mismatch stops the loop there, error swallowed (simd.go:241-267)
    for m, fn in ((1, simd.CosineDistanceBatch), (2, simd.DotProductBatch)):
        r = np.full(9, -7.0, F)
        fn(q, ragged, r, order=order)
        exp = np.full(9, -7.0, F)
        exp[[0, 1, 3, 4]] = want[m][[0, 1, 3, 4]]
        assert np.array_equal(r, exp), (m, r, exp)
    # a rectangular list of vectors equals the flat call
    r = np.empty(9, F)
    simd.EuclideanDistanceBatch(q, vecs, r, order=order)
    assert np.array_equal(r, want[0])
    with pytest.raises(ValueError):
        simd.EuclideanDistanceBatch(q, vecs, np.empty(8, F))


def test_literal_batches_of_the_reference_through_the_rerank_entry(oracle):
    """internal/simd/parallel_reduction_test.go:13-68,116-165: the 8-dim literal batches and the 768-dim 10-vector
    batch, answered from RESIDENT rows by lb_gpu_index_rerank (processChunkInternal's distance step)"""
    gpu_or_skip()
    q8 = np.array([1, 2, 3, 4, 5, 6, 7, 8], F)
    V8 = np.array([[1, 2, 3, 4, 5, 6, 7, 8], [8, 7, 6, 5, 4, 3, 2, 1], [1, 1, 1, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 0, 0, 1]], F)
    for metric in (0, 1, 2):
        idx = new_index(8, metric)
        idx.Add(None, V8)
        for order in (0, 1):
            d, sc = idx.Rerank(q8, np.arange(4), order=order)
            w = oracle.batch_flat(metric, q8, V8, order)  # (dot: the negated value, what the index ranks by)
            assert np.array_equal(d, w.astype(F))
        idx.Close()
    # dot literals: 204, 120, 36, 8 (parallel_reduction_test.go:44-68); cosine of identical vectors 0
    idx = new_index(8, 2)
    idx.Add(None, V8)
    d, _ = idx.Rerank(q8, np.arange(4))
    assert np.array_equal(-d, np.array([204, 120, 36, 8], F))
    idx.Close()
    # 768-dim, 10 vectors: v[i][j] = (i + j) * 0.001, query[j] = j * 0.001 (parallel_reduction_test.go:116-165)
    dim = 768
    q = (np.arange(dim) * 0.001).astype(F)
    V = ((np.arange(10)[:, None] + np.arange(dim)[None, :]) * 0.001).astype(F)
    for metric in (0, 1, 2):
        idx = new_index(dim, metric)
        idx.Add(None, V)
        rows = np.array([9, 0, 3, 3, 7], np.int64)
        d, sc = idx.Rerank(q, rows, order=1)
        w = oracle.batch_flat(metric, q, V[rows], 1)
        assert np.array_equal(d, w.astype(F))
        assert np.array_equal(sc, (F(1) / (F(1) + d)).astype(F))
        idx.Close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _failing_rank_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from longbow_amd import _lib, gpu
    from longbow_amd.sharded import CommSearcher
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = _lib.load_diag()                       # the forced failure is a diagnostic-build hook
        rng = np.random.default_rng(4)
        X = rng.random((20000, 32), dtype=np.float32)
        Q = torch.from_numpy(rng.random((7, 32), dtype=np.float32)).cuda()
        idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=0, Dimension=32, Metric=0), lib=lib)
        idx.Add(None, X[rank::world])
        cs = CommSearcher(idx, rank, world, device_index=0, transport="host", lib=lib)
        lab0, _ = cs.search(Q, 5)                    # a healthy round first
        if rank == 1:
            lib.lb_debug_search_fail_next(1)         # THIS rank's shard search fails; rank 0's does not
        code, msg = 0, ""
        try:
            cs.search(Q, 5)
        except _lib.LongbowGPUError as e:
            code, msg = e.code, str(e)
        lab2, _ = cs.search(Q, 5)                    # and the communicator keeps working afterwards
        q.put((rank, int(code), msg, bool(np.array_equal(lab0.cpu().numpy(), lab2.cpu().numpy()))))
        cs.close()
        idx.Close()
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_still_takes_part_in_the_exchange():
    """rank 1's local search fails: it ships the canonical empty block + its status word instead of leaving the
    collective, so rank 0 does not hang in the all-gather, and BOTH ranks report the failure (results without a shard
    are never returned as if complete)"""
    gpu_or_skip()
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, m0, same0), (r1, c1, m1, same1) = res
    assert c1 == 7 and c0 == 7, res                  # LB_ERR_INTERNAL on the failing rank AND on its peer
    assert "rank 1" in m0, m0
    assert same0 and same1


def test_l2_proof_is_not_inflated_by_a_few_rows_of_a_much_larger_norm(oracle):
    """unnormalised data: five rows 4096 times longer than the rest.  The containment bound of the L2 metric covers key errors
    only up to the norm |q| + d(c_last) (a longer row is farther than c_last by the triangle inequality), so the corpus'
    maximum norm no longer sends every query to the exact scan; the answers are the oracle's either way."""
    gpu_or_skip()
    rng = np.random.default_rng(5)
    n, d, k = 300_000, 128, 10
    X = rng.random((n, d), dtype=F)
    X[rng.integers(0, n, 5)] *= F(4096.0)
    Q = rng.random((64, d), dtype=F)
    idx = new_index(d, 0)
    idx.Add(None, X)
    for nq in (8, 64):
        lab, dist = idx.SearchBatch(Q[:nq], k)
        oi, od = oracle.search_batch(0, Q[:nq], X, k, nthreads=8)
        assert_same(lab, dist, oi, od, f"nq={nq}")
        assert idx.last_fallbacks == 0, (nq, idx.last_fallbacks)
    # a query that IS one of the long rows: its neighbours are the other long rows, far away -- still exact
    long_rows = np.argsort(-np.einsum("ij,ij->i", X, X))[:5]
    Ql = np.ascontiguousarray(X[long_rows[:2]] * F(0.999))
    Qm = np.concatenate([Ql, Q[:6]])
    lab, dist = idx.SearchBatch(Qm, k)
    oi, od = oracle.search_batch(0, Qm, X, k, nthreads=8)
    assert_same(lab, dist, oi, od, "long-row queries")
    idx.Close()


def test_searches_during_adds_see_a_whole_prefix_of_the_corpus(oracle):
    """the reference's gpu.Index serialises Add and Search with one mutex (internal/gpu/faiss_gpu.go:76-145): a search that runs
    while another goroutine adds rows answers for the corpus as it stood between two Adds, never for half a chunk.  One writer
    adds 24 chunks while three readers search; every answer must equal the oracle's over SOME whole prefix whose length lies
    between the index size read before and after the call."""
    gpu_or_skip()
    rng = np.random.default_rng(77)
    d, k, chunk, nchunks = 32, 10, 6000, 24
    X = rng.random((chunk * nchunks, d), dtype=F)
    Q = rng.random((5, d), dtype=F)
    prefix = {m: oracle.search_batch(0, Q, X[:m * chunk], k, nthreads=8) for m in range(1, nchunks + 1)}
    idx = new_index(d, 0)
    idx.Add(None, X[:chunk])
    errs, seen = [], set()
    stop = threading.Event()

    def reader(t):
        try:
            while not stop.is_set():
                nq = (1, 3, 5)[t]
                n0 = idx.ntotal
                lab, dist = idx.SearchBatch(Q[:nq], k)
                n1 = idx.ntotal
                ok = False
                for m in range(n0 // chunk, n1 // chunk + 1):
                    oi, od = prefix[m]
                    if np.array_equal(lab, oi[:nq]) and np.array_equal(dist, od[:nq]):
                        ok = True
                        seen.add(m)
                        break
                if not ok:
                    errs.append((t, n0, n1, lab[0, :3].tolist()))
                    return
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ths = [threading.Thread(target=reader, args=(t,)) for t in range(3)]
    [t.start() for t in ths]
    for m in range(1, nchunks):
        idx.Add(None, X[m * chunk:(m + 1) * chunk])
        time.sleep(0.002)
    stop.set()
    [t.join() for t in ths]
    assert not errs, errs[:3]
    assert idx.ntotal == chunk * nchunks
    lab, dist = idx.SearchBatch(Q, k)
    assert_same(lab, dist, *prefix[nchunks], "after the last Add")
    assert len(seen) >= 2, seen  # the readers did overlap the writer
    idx.Close()


@pytest.mark.parametrize("nq", [16, 256])
def test_query_batch_that_is_not_16_byte_aligned(oracle, nq):
    """round-3 advisor: with the fp16 copy present the fp16 route is offered to a device batch whose pointer is only 4-byte
    aligned (dim % 32 == 0); every launch of such a search must read the queries through paths that take any pointer"""
    gpu_or_skip()
    import torch
    from longbow_amd import gpu
    rng = np.random.default_rng(77)
    n, d, k = 300_000, 64, 10
    X = rng.random((n, d), dtype=F)
    Q = rng.random((nq, d), dtype=F)
    for metric in (0, 1):
        idx = new_index(d, metric)
        idx.Add(None, X)
        assert idx.f16_image_bytes > 0
        dev = torch.device("cuda", 0)
        flat = torch.zeros(nq * d + 8, dtype=torch.float32, device=dev)
        qv = flat[1:1 + nq * d]  # 4 bytes past a 16-byte boundary
        qv.copy_(torch.from_numpy(Q.reshape(-1)).to(dev))
        assert qv.data_ptr() % 16 == 4
        od = torch.empty((nq, k), device=dev)
        ol = torch.empty((nq, k), dtype=torch.int64, device=dev)
        idx.search_device(nq, qv.data_ptr(), k, od.data_ptr(), ol.data_ptr())
        oi, odist = oracle.search_batch(metric, Q, X, k, nthreads=8)
        assert np.array_equal(ol.cpu().numpy(), oi) and np.array_equal(od.cpu().numpy(), odist), (metric, nq)
        idx.Close()


def test_add_sheds_the_fp16_copy_when_the_device_is_full(oracle):
    """round-3 advisor: the fp16 copy of the corpus (half the corpus's bytes again) only speeds searches up.  An Add that
    runs out of device memory while the copy is resident must give the copy back and go through; searches stay exact
    (they stage the f32 rows), and the copy is not rebuilt while the device is that full."""
    gpu_or_skip()
    import torch
    dev = torch.device("cuda", 0)
    d, n0 = 1024, 1_000_000
    idx = new_index(d, 1)
    idx.reserve(n0)
    buf = torch.empty((250_000, d), device=dev)
    from longbow_amd import _lib
    lib = _lib.require_gpu(0)
    for r0 in range(0, n0, 250_000):
        _lib.check(lib.lb_gpu_fill_uniform_device(0, buf.data_ptr(), buf.numel(), 12345, r0 * d, None))
        idx.add_device(250_000, buf.data_ptr())
    copy = idx.f16_image_bytes
    assert copy >= n0 * d * 2
    q = torch.empty((40, d), device=dev)
    _lib.check(lib.lb_gpu_fill_uniform_device(0, q.data_ptr(), q.numel(), 42, 0, None))
    od = torch.empty((40, 10), device=dev)
    ol = torch.empty((40, 10), dtype=torch.int64, device=dev)
    idx.search_device(40, q.data_ptr(), 10, od.data_ptr(), ol.data_ptr())
    want_l, want_d = ol.cpu().numpy().copy(), od.cpu().numpy().copy()
    # somebody else fills the device: less than the next Add needs is left, more than it needs once the copy is gone
    torch.cuda.empty_cache()
    free_b, _ = torch.cuda.mem_get_info(dev)
    add_rows = 250_000                       # 1 GB of rows
    leave = 256 << 20
    blocker = torch.empty(free_b - leave, dtype=torch.uint8, device=dev)
    went_through = True
    try:
        _lib.check(lib.lb_gpu_fill_uniform_device(0, buf.data_ptr(), buf.numel(), 12345, n0 * d, None))
        try:
            idx.add_device(add_rows, buf.data_ptr())      # grows past the reservation: needs ~1 GB, 256 MB are free
        except _lib.LongbowGPUError as e:
            # (the driver sometimes refuses to extend a mapped range -- hipMemSetAccess: invalid argument --, and the index then
            # has to move its rows into a fresh allocation, old + new resident at once: that cannot fit here either way)
            assert e.code == 5, e
            went_through = False
        assert idx.f16_image_bytes == 0                    # the copy went, whatever became of the Add
        assert idx.ntotal == (n0 + add_rows if went_through else n0)
    finally:
        del blocker
        torch.cuda.empty_cache()
    if not went_through:                                   # room again: the same Add succeeds, nothing was half-done
        idx.add_device(add_rows, buf.data_ptr())
        assert idx.ntotal == n0 + add_rows
    # exactness after the shed: strict mode == default mode on the grown corpus, and the old answers are a subset relation
    idx.search_device(40, q.data_ptr(), 10, od.data_ptr(), ol.data_ptr())
    got_l, got_d = ol.cpu().numpy().copy(), od.cpu().numpy().copy()
    idx.set_candidate_mode(0)
    idx.search_device(40, q.data_ptr(), 10, od.data_ptr(), ol.data_ptr())
    assert np.array_equal(got_l, ol.cpu().numpy()) and np.array_equal(got_d, od.cpu().numpy())
    for i in range(40):  # rows of the first million that are still in a list keep their distance
        old = dict(zip(want_l[i].tolist(), want_d[i].tolist()))
        for lab, dist in zip(got_l[i].tolist(), got_d[i].tolist()):
            if lab < n0:
                assert lab in old and old[lab] == dist
    idx.Close()
