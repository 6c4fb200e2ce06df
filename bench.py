#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

Metric (BASELINE.json): k-NN queries/sec at batch = 1024, k = 100 on 1M x 768 f32 (cosine,
brute force), plus p50 single-query latency.  One "step" = one batch of 1024 queries searched
over the rank's 1M x 768 shard, inputs resident in HBM.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) the corpus is N shards of 1M rows partitioned by the
reference's RingSharder; every step ends with the RCCL all-gather of the per-shard top-k and the
device merge.  A unit of `value` is one query searched over one 1M x 768 shard, so the whole-job
rate is N * 1024 / step_time (weak scaling: per-GPU work fixed).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ROWS, DIM, BATCH, K = 1_000_000, 768, 1024, 100
METRIC_COSINE = 1
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense f32 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0          # HBM3E spec


def host_cores():
    """CPU threads this process may really use: min(affinity, cgroup v2 cpu.max quota)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def shard_ids(rank, world, rows):
    """first `rows` vector ids (ascending) that Longbow's ring assigns to shard `rank`"""
    from longbow_amd.sharded import RingSharder
    ring = RingSharder(world, 40)
    out = []
    have = 0
    start = 0
    step = 4 * rows
    while have < rows:
        ids = np.arange(start, start + step, dtype=np.uint64)
        mine = ids[ring.GetShards(ids) == rank]
        out.append(mine)
        have += mine.size
        start += step
    return np.concatenate(out)[:rows].astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=N_ROWS)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast", action="store_true", help="skip the split-bf16 candidate-mode leg")
    ap.add_argument("--cpu-queries", type=int, default=0, help="CPU baseline sample size (0 = 4 per core)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from longbow_amd import _lib, gpu
    from longbow_amd.sharded import ShardedSearcher

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LB_BENCH_FORCE_DIST=1 exercises the RCCL path (process group, all-gather, merge) even with one
    # rank, so the multi-GPU code can be rehearsed on a single-GPU box.
    use_dist = world > 1 or os.environ.get("LB_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    lib = _lib.require_gpu(local_rank)
    rows, B = args.rows, args.batch

    # ---- synthetic inputs, generated in HBM (uniform [0,1), corpus seed 12345, queries 42)
    X = torch.empty((rows, DIM), device=dev)
    Q = torch.empty((B, DIM), device=dev)
    _lib.check(lib.lb_gpu_fill_uniform_device(local_rank, X.data_ptr(), X.numel(), 12345, rank * rows * DIM, None))
    _lib.check(lib.lb_gpu_fill_uniform_device(local_rank, Q.data_ptr(), Q.numel(), 42, 0, None))
    idx = gpu.NewIndexWithConfig(gpu.GPUConfig(DeviceID=local_rank, Dimension=DIM, Metric=METRIC_COSINE))
    idx.reserve(rows)
    d_ids = None
    if use_dist:
        ids = torch.from_numpy(shard_ids(rank, world, rows)).to(dev)
        d_ids = ids.data_ptr()
    idx.add_device(rows, X.data_ptr(), d_ids)
    searcher = ShardedSearcher(idx, rank, world, device=dev, force_collective=use_dist) if use_dist else None
    out_d = torch.empty((B, K), device=dev)
    out_l = torch.empty((B, K), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        if searcher is not None:
            return searcher.search(Q, K)
        idx.search_device(B, Q.data_ptr(), K, out_d.data_ptr(), out_l.data_ptr(), stream)
        return out_l, out_d

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    idx.set_profiling(True)  # HIP events on the search stream around each kernel class
    gemm_ms, gemm_launches, fallbacks = 0.0, 0, 0
    cls_ms = {"gemm": 0.0, "select": 0.0, "rerank": 0.0, "scan": 0.0, "total": 0.0}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = idx.last_timing()
        gemm_ms += tm["gemm"][0]
        gemm_launches += tm["gemm"][1]
        for c in cls_ms:
            cls_ms[c] += tm[c][0]
        fallbacks += idx.last_fallbacks
    barrier()
    elapsed = time.perf_counter() - t0
    idx.set_profiling(False)
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    lab, dd = step()
    lab_h, dist_h = lab.cpu().numpy(), dd.cpu().numpy()

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed

    flops_per_step = 2.0 * B * rows * DIM
    result = {
        "metric": "k-NN queries/sec (batch=1024, k=100), 1Mx768 f32", "value": round(value, 1),
        "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "1Mx768 float32 cosine, batch=1024 queries, k=100, brute-force",
                   "shard_rows": rows, "dim": DIM, "batch": B, "k": K, "metric": "cosine",
                   "global_rows": rows * world,
                   "sharding": "RingSharder(n_gpus, 40) + RCCL all-gather merge" if use_dist else "single shard",
                   "unit_of_value": "one query searched over one 1Mx768 shard"},
        "fallback_queries": int(fallbacks),
        "device_ms_per_step": {c: round(v / args.steps, 4) for c, v in cls_ms.items()},
    }

    flops_per_step = 2.0 * B * rows * DIM
    bytes_per_step = 4.0 * rows * DIM + 4.0 * B * DIM + 12.0 * B * K
    if gemm_ms > 0:
        achieved = flops_per_step * args.steps / (gemm_ms * 1e-3) / 1e12
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("gemm_filter_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result["roofline"] = {
            "bound": "mfma", "kernel": "gemm_filter_kernel", "achieved": round(achieved, 2),
            "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
            "traffic": traffic,
            "launches_per_step": gemm_launches / args.steps,
            "avg_launch_ms": round(gemm_ms / max(gemm_launches, 1), 4),
            "kernel_ms_per_step": round(gemm_ms / args.steps, 4),
            "flops_per_step": flops_per_step,
            "hbm_algorithmic_bytes_per_step": bytes_per_step,
            "hbm_frac_of_8TBs_at_step_rate": round(bytes_per_step / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
        }

    # ---- secondary leg: split-bf16 candidate contraction (3 x bf16 MFMA), same exact results ----------
    if not args.no_fast and not use_dist:
        try:
            idx.set_candidate_mode(1)
            for _ in range(2):
                step()
            idx.set_profiling(True)
            f_ms, f_launch, f_fb = 0.0, 0, 0
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
                tm = idx.last_timing()
                f_ms += tm["gemm"][0]
                f_launch += tm["gemm"][1]
                f_fb += idx.last_fallbacks
            torch.cuda.synchronize(dev)
            f_el = time.perf_counter() - t1
            idx.set_profiling(False)
            labf, ddf = step()
            same = bool(np.array_equal(labf.cpu().numpy(), lab_h) and np.array_equal(ddf.cpu().numpy(), dist_h))
            ach = 3.0 * flops_per_step * args.steps / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0
            result["split_bf16_candidates"] = {
                "value": round(B * args.steps / f_el, 1), "unit": "queries/s", "ms_per_step": round(1e3 * f_el / args.steps, 4),
                "results_identical_to_f32_mode": same, "fallback_queries": int(f_fb),
                "roofline": {"bound": "mfma", "kernel": "gemm_filter_kernel<split>", "achieved": round(ach, 1),
                             "peak": 2500.0, "unit": "TFLOP/s (bf16, 3 MFMA passes counted)", "frac": round(ach / 2500.0, 4),
                             "kernel_ms_per_step": round(f_ms / args.steps, 4)},
                "note": "opt-in lb_gpu_index_set_candidate_mode(1): candidates from hi*hi+hi*lo+lo*hi on bf16 MFMA; "
                        "reported distances/ids still come from the exact f32 re-rank"}
        except Exception as e:  # keep the primary line even if the optional leg fails
            result["split_bf16_candidates"] = {"error": str(e)}
        finally:
            try:
                idx.set_candidate_mode(0)
            except Exception:
                pass

    if rank == 0 and world == 1:
        # ---- p50 single-query latency (the DoExchange path is single-query) ------------------------
        lat = []
        d1 = torch.empty((1, K), device=dev)
        l1 = torch.empty((1, K), dtype=torch.int64, device=dev)
        for i in range(24):  # wall-clock latency, library profiling off
            q1 = Q[i:i + 1].contiguous()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            idx.search_device(1, q1.data_ptr(), K, d1.data_ptr(), l1.data_ptr(), stream)
            lat.append(1e3 * (time.perf_counter() - t1))
        idx.set_profiling(True)  # second pass: HIP events around the scan kernel itself
        scan_ms = []
        for i in range(24):
            q1 = Q[i:i + 1].contiguous()
            idx.search_device(1, q1.data_ptr(), K, d1.data_ptr(), l1.data_ptr(), stream)
            scan_ms.append(idx.last_timing()["scan"][0])
        idx.set_profiling(False)
        lat = sorted(lat[4:])
        p50 = lat[len(lat) // 2]
        scan_p50 = sorted(scan_ms[4:])[len(scan_ms[4:]) // 2]
        result["p50_latency_ms"] = round(p50, 4)
        result["latency_roofline"] = {
            "bound": "hbm", "kernel": "scan_kernel", "achieved": round(4.0 * rows * DIM / (scan_p50 * 1e-3) / 1e9, 1),
            "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(4.0 * rows * DIM / (scan_p50 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "scan_kernels_ms": round(scan_p50, 4)}

        # ---- parity gate + CPU baseline (oracle = test/bench infrastructure, never the product) -----
        from oracle import oracle_c as oc
        Xh = X.cpu().numpy()
        Qh = Q.cpu().numpy()
        cores = host_cores()
        sub = np.arange(0, B, max(1, B // 16))[:16]
        oi, od = oc.search_batch(METRIC_COSINE, Qh[sub], Xh, K, nthreads=cores)
        ok = bool(np.array_equal(oi, lab_h[sub]) and np.array_equal(od, dist_h[sub]))
        result["parity"] = {"checked_queries": int(len(sub)), "index_sets_equal": bool(np.array_equal(oi, lab_h[sub])),
                            "distances_bit_equal": bool(np.array_equal(od, dist_h[sub])), "ok": ok}
        if not args.no_cpu_baseline:
            nqc = args.cpu_queries or 16 * cores  # 16 full-corpus scans per thread: tens of core-seconds
            nqc = min(nqc, B)
            secs, bi, bd = oc.cpu_baseline(METRIC_COSINE, Qh[:nqc], Xh, K, nthreads=cores, simd=1)
            agree = float((bi == lab_h[:nqc]).mean())
            result["cpu_baseline"] = {
                "value": round(nqc / secs, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                "sample": f"{nqc} of the {B} queries, each scanned over the full {rows}x{DIM} corpus "
                          f"(reference idiom: queries partitioned over {cores} threads, SIMD C port of internal/simd)",
                "seconds": round(secs, 2), "label_agreement_with_gpu": round(agree, 5)}

    if rank == 0:
        print(json.dumps(result), flush=True)
    idx.Close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
