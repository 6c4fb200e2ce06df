#!/usr/bin/env python3
"""Which kernel should generate the candidates of a batch?  Times every route the library can be forced onto
(diagnostic build: LB_FORCE_ROUTE) over a grid of dimensions, corpus sizes and batch sizes, next to the route the
library picks by itself, and prints how far the pick is from the best forced route.  The cost-model constants of
index.hip (choose_route) are fitted to this table; tests/test_gpu_routes.py asserts the <= 10 % bound on a sub-grid.
usage: LB_GPU_SO=longbow_amd/liblongbow_gpu_diag.so python tools/route_grid.py [--quick] [--json out.json]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from longbow_amd import _lib, gpu

NAMES = {1: "narrow32", 2: "narrow64", 4: "wide_f32", 5: "tall2", 6: "tall16", 7: "narrow16"}  # (3, the 256 x 128 split tile, went in round 4)


def grid(quick):
    dims = (128, 768) if quick else (128, 384, 768, 1536)
    ns = (100_000, 1_000_000) if quick else (100_000, 1_000_000, 4_000_000)
    bs = (64, 256, 1024) if quick else (8, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024)
    return dims, ns, bs


def time_search(idx, Q, B, K, od, ol, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


def main():
    quick = "--quick" in sys.argv
    out_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    lib = _lib.load_diag() if "diag" not in _lib.SO_PATH else _lib.load()
    raw = C.CDLL(_lib.DIAG_SO_PATH if "diag" not in _lib.SO_PATH else _lib.SO_PATH)
    raw.lb_debug_last_route.restype = C.c_int
    dims, ns, bs = grid(quick)
    K = 100
    rows_out = []
    worst = 0.0
    for D in dims:
        for n in ns:
            X = torch.empty((n, D), device="cuda"); Q = torch.empty((1024, D), device="cuda")
            lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
            lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
            idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1), lib=lib)
            idx.add_device(n, X.data_ptr())
            del X
            for B in bs:
                od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
                os.environ["LB_FORCE_ROUTE"] = "0"
                for _ in range(2):
                    idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
                t_auto = time_search(idx, Q, B, K, od, ol)
                picked = raw.lb_debug_last_route() // 10
                forced = {}
                for r in (1, 2, 5, 6, 7, 4):
                    os.environ["LB_FORCE_ROUTE"] = str(r)
                    idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
                    if raw.lb_debug_last_route() // 10 != r:
                        continue  # not available for this batch
                    if r == 4 and B * n * D > 4e11:
                        continue  # (the f32 tile at the large end: seconds per point, never the pick)
                    forced[r] = time_search(idx, Q, B, K, od, ol, reps=3)
                os.environ["LB_FORCE_ROUTE"] = "0"
                best = min(forced.values())
                ratio = forced.get(picked, t_auto) / best
                worst = max(worst, ratio)
                rows_out.append({"D": D, "n": n, "B": B, "picked": NAMES[picked], "ms_auto": round(t_auto, 4),
                                 "forced_ms": {NAMES[r]: round(v, 4) for r, v in forced.items()}, "picked_over_best": round(ratio, 3)})
                f = "  ".join(f"{NAMES[r]}={v:.3f}" for r, v in forced.items())
                print(f"D={D:5d} n={n:8d} B={B:5d}  pick {NAMES[picked]:9s} {t_auto:8.3f} ms  x{ratio:.2f} of best   [{f}]", flush=True)
            idx.Close()
            torch.cuda.empty_cache()
    print(f"worst picked/best ratio: {worst:.3f}")
    if out_path:
        json.dump(rows_out, open(out_path, "w"), indent=0)


if __name__ == "__main__":
    main()
