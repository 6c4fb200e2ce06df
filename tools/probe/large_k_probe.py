import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from longbow_amd import _lib, gpu
lib = _lib.require_gpu(0)
D = 768
for rows in (500000, 1000000):
    X = torch.empty((rows, D), device="cuda"); Q = torch.empty((1024, D), device="cuda")
    lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
    lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
    for image in (1, 0):
        idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1))
        if not image: idx.set_f16_image(0)
        idx.add_device(rows, X.data_ptr())
        for K in (100, 300, 500):
            line = []
            for B in (1, 8, 32, 64, 256, 1024):
                od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
                ts = []
                for i in range(6):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
                    ts.append(time.perf_counter() - t0)
                t = sorted(ts[2:])[2] * 1e3
                line.append(f"{B}:{t:.3f}" + (f"/fb{idx.last_fallbacks}" if idx.last_fallbacks else "") + (f"/gu{idx.fused_giveups}" if idx.fused_giveups else ""))
            print(f"rows {rows} image {image} k {K}  " + "  ".join(line), flush=True)
        idx.Close()
