"""N>1 path on CPU: two gloo ranks, RingSharder partition, all-gather of per-shard top-k, merge.
The GPU search/merge steps are replaced by the oracle here (test infrastructure) -- what is
under test is the partitioning and the collective plumbing of longbow_amd.sharded."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, metric, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from longbow_amd.sharded import RingSharder, ShardedSearcher
    from oracle import oracle_c as oc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)
        n, d, nq, k = 4000, 32, 9, 10
        X = rng.random((n, d), dtype=np.float32)
        Q = rng.random((nq, d), dtype=np.float32)
        ids = np.arange(n, dtype=np.int64) * 3 + 5  # user ids != positions
        owner = RingSharder(world, 40).GetShards(ids.astype(np.uint64))
        mine = np.nonzero(owner == rank)[0]
        Xl, idl = X[mine], ids[mine]

        def local_search(queries, kk):
            oi, od = oc.search_batch(metric, queries.numpy(), Xl, kk, ids=idl)
            return oi, od

        def merge(nshards, nqq, kk, dist_all, lab_all, dist_out, lab_out, stream):
            for b in range(nqq):
                dd = dist_all[:, b, :].reshape(-1).numpy()
                ll = lab_all[:, b, :].reshape(-1).numpy().astype(np.uint64)
                order = np.lexsort((ll, dd))[:kk]
                dist_out[b] = torch.from_numpy(dd[order])
                lab_out[b] = torch.from_numpy(ll[order].astype(np.int64))

        s = ShardedSearcher(None, rank, world, device=torch.device("cpu"), local_search=local_search, merge=merge)
        lab, dd = s.search(torch.from_numpy(Q), k)
        gi, gd = oc.search_batch(metric, Q, X, k, ids=ids)
        ok = bool(np.array_equal(lab.numpy(), gi) and np.array_equal(dd.numpy(), gd))
        q.put((rank, ok, int(len(mine))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric", [0, 2])
def test_two_rank_sharded_search_equals_global(metric, oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, metric, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sum(n for _, _, n in res) == 4000


class _NoRcclLib:
    """stand-in for liblongbow_gpu.so on a host where RCCL cannot be loaded: the id call fails, init must never be reached"""

    def lb_gpu_comm_get_unique_id(self, buf):
        return 6  # LB_ERR_UNSUPPORTED

    def lb_gpu_comm_init_rank(self, *a):
        raise AssertionError("init_rank reached without a unique id")


def _uid_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from longbow_amd import _lib
    from longbow_amd.sharded import CommSearcher
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        try:
            CommSearcher(None, rank, world, transport="rccl", lib=_NoRcclLib())
            raised = False
        except _lib.LongbowGPUError:
            raised = True
        # the ranks are still in step: the agreement all-reduce bench.py's make_searcher does next goes through
        ok = torch.tensor([0 if raised else 1])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        q.put((rank, raised, int(ok.item())))
    finally:
        dist.destroy_process_group()


def test_unique_id_failure_on_rank0_is_raised_on_every_rank_without_a_hang():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, 0), (1, True, 0)], res


def _small_worker(rank, world, port, n, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from longbow_amd.sharded import RingSharder, ShardedSearcher
    from oracle import oracle_c as oc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)
        d, nq, k = 16, 5, 10
        X = rng.random((n, d), dtype=np.float32)
        Q = rng.random((nq, d), dtype=np.float32)
        ids = np.arange(n, dtype=np.int64) * 3 + 5
        owner = RingSharder(world, 40).GetShards(ids.astype(np.uint64))
        mine = np.nonzero(owner == rank)[0]
        Xl, idl = X[mine], ids[mine]

        def local_search(queries, kk):
            if len(mine) == 0:  # an empty shard answers with the canonical padding (what lb_gpu_index_search does for n = 0)
                return (np.full((queries.shape[0], kk), -1, np.int64),
                        np.full((queries.shape[0], kk), np.finfo(np.float32).max, np.float32))
            return oc.search_batch(0, queries.numpy(), Xl, kk, ids=idl)

        def merge(nshards, nqq, kk, dist_all, lab_all, dist_out, lab_out, stream):
            for b in range(nqq):
                dd = dist_all[:, b, :].reshape(-1).numpy()
                ll = lab_all[:, b, :].reshape(-1).numpy()
                key = np.where(ll < 0, np.iinfo(np.int64).max, ll)  # padding sorts behind every real entry
                order = np.lexsort((key, dd))[:kk]
                dist_out[b] = torch.from_numpy(dd[order])
                lab_out[b] = torch.from_numpy(ll[order])

        s = ShardedSearcher(None, rank, world, device=torch.device("cpu"), local_search=local_search, merge=merge)
        lab, dd = s.search(torch.from_numpy(Q), k)
        gi, gd = oc.search_batch(0, Q, X, k, ids=ids)
        q.put((rank, bool(np.array_equal(lab.numpy(), gi) and np.array_equal(dd.numpy(), gd)), int(len(mine))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(4, 4000), (4, 3), (6, 7)])
def test_more_ranks_and_empty_shards(world, n, oracle):
    """4 and 6 gloo ranks; with fewer rows than ranks some shards are empty and k exceeds every shard's row count:
    the merged answer is still the global one (padding -1 / FLT_MAX behind the real entries)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_small_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sum(m for _, _, m in res) == n
