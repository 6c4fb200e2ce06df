"""The N>1 path with the REAL GPU pieces (HIP search per shard + HIP merge kernel) on one MI355X:
two ranks share cuda:0, transport is gloo (RCCL needs one GPU per rank; the 8-GPU run is the
driver's).  Checks that the sharded result equals the oracle's global result."""
import os
import socket

import numpy as np
import pytest

from tests.gpu_util import gpu_or_skip

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch  # first: see tests/conftest.py
    from longbow_amd import gpu
    from longbow_amd.sharded import RingSharder, ShardedSearcher
    from oracle import oracle_c as oc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        n, d, nq, k = 40000, 64, 48, 20
        X = rng.random((n, d), dtype=np.float32)
        Q = rng.random((nq, d), dtype=np.float32)
        ids = np.arange(n, dtype=np.int64) * 2 + 1
        owner = RingSharder(world, 40).GetShards(ids.astype(np.uint64))
        mine = np.nonzero(owner == rank)[0]
        idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=0, Dimension=d, Metric=2))
        idx.Add(ids[mine], X[mine])
        dev = torch.device("cuda", 0)
        s = ShardedSearcher(idx, rank, world, device=dev)
        lab, dd = s.search(torch.from_numpy(Q).to(dev), k)
        torch.cuda.synchronize()
        gi, gd = oc.search_batch(2, Q, X, k, ids=ids, nthreads=4)
        ok = bool(np.array_equal(lab.cpu().numpy(), gi) and np.array_equal(dd.cpu().numpy(), gd))
        q.put((rank, ok, int(len(mine))))
        idx.Close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_hip_search_and_merge(oracle):
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
    assert sum(n for _, _, n in res) == 40000
