import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from longbow_amd import _lib, gpu
rows, D, B, K = 1_000_000, 768, 1024, 100
lib = _lib.require_gpu(0)
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((B, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
raw = C.CDLL(_lib.SO_PATH); raw.lb_debug_set_gemm_ablation.argtypes = [C.c_int]; raw.lb_debug_set_gemm_glds.argtypes = [C.c_int]
probe = (C.c_ulonglong * 8)()
for phase, abl in (("warm (default kernel) x40", 0), ("clock-stamped kernel (register-staged) x40", 5)):
    raw.lb_debug_set_gemm_glds(-1 if abl else 0); raw.lb_debug_set_gemm_ablation(abl)
    raw.lb_debug_read_clock_probe(probe, 1)
    t0 = time.perf_counter()
    for _ in range(40):
        idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
    dt = (time.perf_counter() - t0) / 40
    raw.lb_debug_read_clock_probe(probe, 1)
    msg = f"{phase}: {dt*1e3:.3f} ms/batch"
    if probe[1]:
        msg += (f"; shader clock {probe[0] / probe[1] * 100:.0f} MHz over {probe[2]} workgroups; mean cycles per workgroup: "
                f"prologue {probe[3] / probe[2]:.0f}, main loop {probe[0] / probe[2]:.0f}, epilogue {probe[4] / probe[2]:.0f}")
    print(msg, flush=True)
raw.lb_debug_set_gemm_ablation(0); raw.lb_debug_set_gemm_glds(0)
