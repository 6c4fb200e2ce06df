"""Multi-GPU search through the C ABI's own communicator (lb_gpu_comm_*: shard search + one all-gather of the
packed per-shard top-k + device merge; semantics of internal/store/sharded_hnsw.go:414-503 and
result_merger.go:34-101) on ONE MI355X: RCCL with a single rank / single device, and two ranks sharing
cuda:0 with the host-staged transport over gloo.  The 8-GPU RCCL run is the driver's."""
import os
import socket

import numpy as np
import pytest

from tests.gpu_util import gpu_or_skip, new_index

pytestmark = pytest.mark.gpu
F = np.float32
torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_single_process_node_searcher_and_single_rank_rccl(oracle):
    gpu_or_skip()
    from longbow_amd.sharded import CommSearcher, NodeSearcher
    rng = np.random.default_rng(3)
    n, d, nq, k = 30000, 96, 40, 25
    X = rng.random((n, d), dtype=F)
    Q = rng.random((nq, d), dtype=F)
    ids = (np.arange(n, dtype=np.int64) * 7 + 3)
    idx = new_index(d, 0)
    idx.Add(ids, X)
    want_l, want_d = idx.SearchBatch(Q, k)
    # lb_gpu_comm_init_all(1): the one-process-per-node form with a single device
    node = NodeSearcher([idx])
    lab, dd = node.search(Q, k)
    assert np.array_equal(lab, want_l) and np.array_equal(dd, want_d)
    lab1, dd1 = node.search(Q[:1], 1)
    assert np.array_equal(lab1, want_l[:1, :1])
    node.close()
    # lb_gpu_comm_init_rank with nranks = 1: RCCL really initialises (unique id, communicator)
    cs = CommSearcher(idx, 0, 1, device_index=0, transport="rccl")
    lab, dd = cs.search(torch.from_numpy(Q).cuda(), k)
    assert np.array_equal(lab.cpu().numpy(), want_l) and np.array_equal(dd.cpu().numpy(), want_d)
    cs.close()
    oi, od = oracle.search_batch(0, Q, X, k, ids=ids, nthreads=4)
    assert np.array_equal(want_l, oi) and np.array_equal(want_d, od)
    idx.Close()


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch  # first: see tests/conftest.py
    from longbow_amd import gpu
    from longbow_amd.sharded import CommSearcher, GpuPartition
    from oracle import oracle_c as oc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)
        n, d, nq, k = 50000, 64, 33, 40
        X = rng.random((n, d), dtype=np.float32)
        X[1000:1010] = X[0:10]  # equal distances across shards: the label order decides
        Q = rng.random((nq, d), dtype=np.float32)
        ids = np.arange(n, dtype=np.int64) * 2 + 1
        part = GpuPartition(world, 8, 40)
        mine = np.nonzero(part.GetGpus(ids.astype(np.uint64)) == rank)[0]
        idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=0, Dimension=d, Metric=1))
        idx.Add(ids[mine], X[mine])
        cs = CommSearcher(idx, rank, world, device_index=0, transport="host")
        lab, dd = cs.search(torch.from_numpy(Q).cuda(), k)
        torch.cuda.synchronize()
        gi, gd = oc.search_batch(1, Q, X, k, ids=ids, nthreads=4)
        # ties: the oracle orders by (distance, row position), the merge by (distance, id) -- identical here
        # because ids ascend with rows
        ok = bool(np.array_equal(lab.cpu().numpy(), gi) and np.array_equal(dd.cpu().numpy(), gd))
        q.put((rank, ok, int(len(mine))))
        cs.close()
        idx.Close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_through_lb_gpu_comm_host_transport(oracle):
    gpu_or_skip()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sum(n for _, _, n in res) == 50000


def test_fill_rows_by_id_matches_the_generator(oracle):
    gpu_or_skip()
    from longbow_amd import _lib
    lib = _lib.load()
    ids = np.array([0, 5, 123456789, 7, 999_999_999], np.int64)
    d = 48
    out = torch.empty((ids.size, d), device="cuda")
    assert lib.lb_gpu_fill_uniform_rows_device(0, out.data_ptr(), torch.from_numpy(ids).cuda().data_ptr(), ids.size, d, 12345, None) == 0
    got = out.cpu().numpy()
    for r, i in enumerate(ids):
        assert np.array_equal(got[r], oracle.fill_uniform(d, 12345, int(i) * d))
