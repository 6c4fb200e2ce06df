"""Corpus sharding across the GPUs of one node + cross-shard top-k merge.

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI).  The corpus is
partitioned by Longbow's RingSharder (internal/store/sharding_strategy.go:40-127, built as
NewRingSharder(numShards, 40) in internal/store/sharded_hnsw.go:158); every rank searches the
same query batch over its shard with the HIP index, ONE all-gather moves the per-shard top-k
(B*k*(4+8) bytes per rank), and every rank merges with the HIP merge kernel -- the device form
of ShardedHNSW's concat+sort (sharded_hnsw.go:494-503) / MergeSortedStreams
(internal/store/result_merger.go:34-101).

The search and merge steps are injectable so that the collective plumbing can be exercised on
CPU (gloo) in tests with the oracle standing in for the GPU; the defaults are the HIP path and
raise without a GPU.
"""
import numpy as np

from . import _lib
from .simd import MetricType

FNV_OFFSET = np.uint32(2166136261)
FNV_PRIME = np.uint32(16777619)


def fnv1a32_bytes(data: bytes) -> int:
    h = 2166136261
    for b in data:
        h ^= b
        h = (h * 16777619) & 0xFFFFFFFF
    return h


class RingSharder:
    """store.RingSharder: consistent hashing of VectorIDs onto a fixed set of shards."""

    def __init__(self, num_shards, vnodes=40):
        if vnodes <= 0:
            vnodes = 20  # sharding_strategy.go:51-53
        self.num_shards = num_shards
        self.vnodes = vnodes
        ring = {}
        hashes = []
        for s in range(num_shards):           # addShard order (sharding_strategy.go:58-75)
            for v in range(vnodes):
                h = fnv1a32_bytes(f"{s}:{v}".encode())  # strconv.Itoa(shard)+":"+strconv.Itoa(vnode)
                ring[h] = s                    # Go map assignment: a later equal hash overwrites
                hashes.append(h)
        hashes.sort()
        self.sorted_hashes = np.array(hashes, np.uint32)
        self.owners = np.array([ring[h] for h in hashes], np.int32)

    @staticmethod
    def hash_ids(ids):
        """hashID (sharding_strategy.go:86-101): FNV-1a-32 over the 8 little-endian bytes of uint64(id)."""
        v = np.asarray(ids).astype(np.uint64)
        h = np.full(v.shape, FNV_OFFSET, np.uint32)
        with np.errstate(over="ignore"):
            for i in range(8):
                b = ((v >> np.uint64(8 * i)) & np.uint64(0xFF)).astype(np.uint32)
                h = (h ^ b) * FNV_PRIME
        return h

    def GetShards(self, ids):
        if self.sorted_hashes.size == 0:
            return np.zeros(np.asarray(ids).shape, np.int32)
        h = self.hash_ids(ids)
        idx = np.searchsorted(self.sorted_hashes, h, side="left")  # sort.Search(first >= h)
        idx[idx == self.sorted_hashes.size] = 0                     # wrap (sharding_strategy.go:113-115)
        return self.owners[idx]

    def GetShard(self, vid):
        return int(self.GetShards(np.array([vid], np.uint64))[0])

    def ActiveShards(self):
        return self.num_shards


def _hip_merge(device):
    def merge(nshards, nq, k, dist_all, lab_all, dist_out, lab_out, stream):
        lib = _lib.require_gpu(device)
        _lib.check(lib.lb_gpu_merge_topk_device(device, nshards, nq, k, dist_all.data_ptr(), lab_all.data_ptr(),
                                                dist_out.data_ptr(), lab_out.data_ptr(), stream))
    return merge


class ShardedSearcher:
    """One rank's view of a corpus sharded over `world_size` GPUs."""

    def __init__(self, index, rank, world_size, group=None, device=None, local_search=None, merge=None,
                 force_collective=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.index = index
        self.rank, self.world = rank, world_size
        self.group = group
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._local_search = local_search
        self._merge = merge
        self._force = force_collective  # run the all-gather even with one rank (rehearsal)
        self._bufs = {}

    def _buffers(self, nq, k):
        key = (nq, k)
        if key not in self._bufs:
            t = self.torch
            dev = self.device
            self._bufs[key] = dict(
                d=t.empty((nq, k), dtype=t.float32, device=dev), l=t.empty((nq, k), dtype=t.int64, device=dev),
                da=t.empty((self.world, nq, k), dtype=t.float32, device=dev),
                la=t.empty((self.world, nq, k), dtype=t.int64, device=dev),
                do=t.empty((nq, k), dtype=t.float32, device=dev), lo=t.empty((nq, k), dtype=t.int64, device=dev))
        return self._bufs[key]

    def search(self, queries, k):
        """queries: [nq, dim] float32 tensor on this rank's device (identical on all ranks).
        Returns (labels [nq,k] int64, dist [nq,k] float32): global top-k, identical on all ranks."""
        t = self.torch
        nq = queries.shape[0]
        b = self._buffers(nq, k)
        if self._local_search is None and self._merge is None:
            return self._search_hip(queries, k, nq, b)
        # injectable path (CPU/gloo tests, or mixed): separate tensors
        if self._local_search is not None:
            lab, d = self._local_search(queries, k)
            b["l"].copy_(t.as_tensor(lab))
            b["d"].copy_(t.as_tensor(d))
            stream = None
        else:
            stream = t.cuda.current_stream(self.device).cuda_stream
            self.index.search_device(nq, queries.data_ptr(), k, b["d"].data_ptr(), b["l"].data_ptr(), stream)
        if self.world > 1 or self._force:
            if b["d"].is_cuda:  # gloo transport with device tensors: stage via host
                hd = [t.empty(b["d"].shape, dtype=t.float32) for _ in range(self.world)]
                hl = [t.empty(b["l"].shape, dtype=t.int64) for _ in range(self.world)]
                self.dist.all_gather(hd, b["d"].cpu(), group=self.group)
                self.dist.all_gather(hl, b["l"].cpu(), group=self.group)
                b["da"].copy_(t.stack(hd))
                b["la"].copy_(t.stack(hl))
            else:
                self.dist.all_gather([b["da"][r] for r in range(self.world)], b["d"], group=self.group)
                self.dist.all_gather([b["la"][r] for r in range(self.world)], b["l"], group=self.group)
        else:
            b["da"][0].copy_(b["d"])
            b["la"][0].copy_(b["l"])
        merge = self._merge if self._merge is not None else _hip_merge(self.device.index or 0)
        merge(self.world, nq, k, b["da"], b["la"], b["do"], b["lo"], stream)
        return b["lo"], b["do"]

    def _search_hip(self, queries, k, nq, b):
        """product path: HIP search -> ONE all-gather of the packed (labels | distances) block -> HIP merge"""
        t = self.torch
        lib = _lib.require_gpu(self.device.index or 0)
        nk = nq * k
        block_words = nk + (nk * 4 + 7) // 8  # int64 words per shard block
        key = ("packed", nq, k)
        if key not in self._bufs:
            self._bufs[key] = (t.empty(block_words, dtype=t.int64, device=self.device),
                               t.empty(self.world * block_words, dtype=t.int64, device=self.device))
        mine, allb = self._bufs[key]
        stream = t.cuda.current_stream(self.device).cuda_stream
        self.index.search_device(nq, queries.data_ptr(), k, mine.data_ptr() + nk * 8, mine.data_ptr(), stream)
        backend = self.dist.get_backend(self.group) if (self.world > 1 or self._force) else None
        if backend == "nccl":      # RCCL over xGMI: the one exchange step of the path
            self.dist.all_gather_into_tensor(allb, mine, group=self.group)
        elif backend is not None:  # gloo with device tensors (single-GPU rehearsal): stage via host
            parts = [t.empty(block_words, dtype=t.int64) for _ in range(self.world)]
            self.dist.all_gather(parts, mine.cpu(), group=self.group)
            allb.copy_(t.cat(parts))
        else:
            allb.copy_(mine)
        _lib.check(lib.lb_gpu_merge_topk_packed_device(self.device.index or 0, self.world, nq, k, allb.data_ptr(),
                                                       b["do"].data_ptr(), b["lo"].data_ptr(), stream))
        return b["lo"], b["do"]


def partition_rows(ids, num_shards, vnodes=40):
    """row -> shard for a whole corpus; also returns per-shard counts (skew report)."""
    ring = RingSharder(num_shards, vnodes)
    owner = ring.GetShards(ids)
    counts = np.bincount(owner, minlength=num_shards)
    return owner, counts


# ---------------------------------------------------------------------------------------------------
# Shards -> GPUs.  The reference's ring is badly skewed at one shard per GPU (FNV-1a-32 over the short keys
# "<shard>:<vnode>" clusters: with 8 shards x 40 vnodes one shard owns 30 % of the hash space, max/mean 2.4),
# and the largest shard sets the step time.  ShardedHNSW's NumShards is a free parameter
# (internal/store/sharded_hnsw.go:150-165), so a node runs MORE ring shards than GPUs and packs them onto
# the GPUs by size: same RingSharder, same id -> shard map, balanced devices.
# ---------------------------------------------------------------------------------------------------
class GpuPartition:
    """RingSharder(num_gpus * shards_per_gpu, vnodes) + a largest-first packing of the shards onto the GPUs
    by the fraction of the hash space each shard owns (exact, from the ring's arcs; no data needed)."""

    def __init__(self, num_gpus, shards_per_gpu=8, vnodes=40):
        self.num_gpus = num_gpus
        self.ring = RingSharder(num_gpus * shards_per_gpu, vnodes)
        h = self.ring.sorted_hashes.astype(np.uint64)
        arcs = np.empty(h.size, np.float64)
        arcs[1:] = (h[1:] - h[:-1]).astype(np.float64)
        arcs[0] = float(h[0]) + float((1 << 32) - int(h[-1]))  # keys above the last point wrap to point 0
        frac = np.bincount(self.ring.owners, weights=arcs, minlength=self.ring.num_shards) / float(1 << 32)
        self.shard_fraction = frac
        load = np.zeros(num_gpus)
        self.shard_gpu = np.zeros(self.ring.num_shards, np.int32)
        for s in np.argsort(-frac, kind="stable"):  # largest first onto the least loaded GPU
            g = int(np.argmin(load))
            self.shard_gpu[s] = g
            load[g] += frac[s]
        self.gpu_fraction = load

    def GetGpus(self, ids):
        return self.shard_gpu[self.ring.GetShards(ids)]

    def skew(self):
        """max / mean of the GPUs' expected shares"""
        return float(self.gpu_fraction.max() / self.gpu_fraction.mean())


class CommSearcher:
    """Sharded search through the library's own communicator (lb_gpu_comm_*): no torch in the data path.
    transport "rccl": ncclCommInitRank inside liblongbow_gpu.so, the unique id travels over `group`;
    transport "host": the exchange is `group`'s all_gather between host buffers (gloo), staged by the library."""

    def __init__(self, index, rank, world_size, device_index=0, transport="rccl", group=None, lib=None):
        import ctypes as C
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.C = torch, dist, C
        self.index, self.rank, self.world, self.group = index, rank, world_size, group
        self.device_index = device_index
        lib = lib or _lib.require_gpu(device_index)
        self._lib = lib
        st = C.c_int(0)
        self._cb = None
        if transport == "rccl":
            uid = (C.c_char * 128)()
            rc0 = lib.lb_gpu_comm_get_unique_id(uid) if rank == 0 else 0
            # (rank 0 takes part in the broadcast even when it has no id to give: its peers are already waiting in it)
            box = [bytes(uid) if rc0 == 0 else b""]
            if world_size > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            if len(box[0]) != 128:
                raise _lib.LongbowGPUError(rc0 or 7, "rank 0 could not create the RCCL unique id (lb_gpu_comm_get_unique_id)")
            buf = C.create_string_buffer(box[0], 128)
            h = lib.lb_gpu_comm_init_rank(device_index, world_size, rank, buf, C.byref(st))
        elif transport == "host":
            FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)

            def allgather(ctx, send, recv, nbytes):
                try:
                    mine = torch.frombuffer((C.c_char * nbytes).from_address(send), dtype=torch.uint8).clone()
                    parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world_size)]
                    dist.all_gather(parts, mine, group=group)
                    out = torch.frombuffer((C.c_char * (nbytes * world_size)).from_address(recv), dtype=torch.uint8)
                    out.copy_(torch.cat(parts))
                    return 0
                except Exception:  # never let an exception cross the C boundary
                    return 1
            self._cb = FN(allgather)
            h = lib.lb_gpu_comm_init_host(device_index, world_size, rank, self._cb, None, C.byref(st))
        else:
            raise ValueError(transport)
        if not h:
            _lib.check(st.value or 7)
        self._h = C.c_void_p(h)

    def prepare(self, nq_max, k_max):
        """size the exchange buffers for searches of up to nq_max queries / k_max results (lb_gpu_comm_prepare).  Not a
        collective: call it on every rank right after construction and agree on the outcome BEFORE the first search -- a
        search within the prepared size allocates nothing between the shard search and the all-gather, so a rank that
        runs out of memory cannot strand its peers there."""
        rc = self._lib.lb_gpu_comm_prepare(self._h, int(nq_max), int(k_max))
        if rc != 0:
            raise _lib.LongbowGPUError(rc, (self._lib.lb_gpu_comm_last_error(self._h) or b"").decode())

    def search(self, queries, k):
        """queries: [nq, dim] float32 CUDA tensor, identical on all ranks -> (labels, dist) CUDA tensors"""
        t = self.torch
        nq = queries.shape[0]
        d = t.empty((nq, k), dtype=t.float32, device=queries.device)
        lab = t.empty((nq, k), dtype=t.int64, device=queries.device)
        stream = t.cuda.current_stream(queries.device).cuda_stream
        rc = self._lib.lb_gpu_comm_search_device(self._h, self.index._h, nq, queries.data_ptr(), k, d.data_ptr(),
                                                 lab.data_ptr(), stream)
        if rc != 0:
            raise _lib.LongbowGPUError(rc, (self._lib.lb_gpu_comm_last_error(self._h) or b"").decode())
        return lab, d

    def close(self):
        if self._h:
            self._lib.lb_gpu_comm_free(self._h)
            self._h = None


class NodeSearcher:
    """ONE process driving every GPU of the node (what the Go server does): lb_gpu_comm_init_all + one index
    per device; host queries in, host results out."""

    def __init__(self, indexes, devices=None):
        import ctypes as C
        self.C = C
        lib = _lib.require_gpu(0)
        self._lib = lib
        self.indexes = list(indexes)
        n = len(self.indexes)
        devs = (C.c_int * n)(*(devices if devices is not None else range(n)))
        st = C.c_int(0)
        h = lib.lb_gpu_comm_init_all(n, devs, C.byref(st))
        if not h:
            _lib.check(st.value or 7)
        self._h = C.c_void_p(h)
        self._shards = (C.c_void_p * n)(*[ix._h for ix in self.indexes])

    def search(self, queries, k):
        q = np.ascontiguousarray(queries, np.float32)
        nq = q.shape[0]
        dist = np.empty((nq, k), np.float32)
        lab = np.empty((nq, k), np.int64)
        rc = self._lib.lb_gpu_comm_search_all(self._h, self._shards, nq, q.ctypes.data, k, dist.ctypes.data, lab.ctypes.data)
        if rc != 0:
            raise _lib.LongbowGPUError(rc, (self._lib.lb_gpu_comm_last_error(self._h) or b"").decode())
        return lab, dist

    def close(self):
        if self._h:
            self._lib.lb_gpu_comm_free(self._h)
            self._h = None
