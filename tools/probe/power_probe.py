"""Shader clock and socket power while one search shape runs in a loop (rocm-smi polled from a side thread).
usage: python tools/probe/power_probe.py [batch] [seconds]   (CAND_MODE as in tools/bench_sweep.py)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.getcwd())
import torch
from longbow_amd import _lib, gpu
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
rows, D, K = 1_000_000, 768, 100
lib = _lib.require_gpu(0)
X = torch.empty((rows, D), device="cuda"); Q = torch.empty((B, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, X.data_ptr(), X.numel(), 12345, 0, None)
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
idx = gpu.NewIndexWithConfig(gpu.GPUConfig(0, D, 1)); idx.reserve(rows); idx.add_device(rows, X.data_ptr())
if os.environ.get('CAND_MODE'): idx.set_candidate_mode(int(os.environ['CAND_MODE']))
od = torch.empty((B, K), device="cuda"); ol = torch.empty((B, K), dtype=torch.int64, device="cuda")
samples, stop = [], False
def poll():
    while not stop:
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True)
        samples.append([l.strip() for l in r.stdout.splitlines() if any(t in l for t in ("sclk", "Power", "mclk", "junction"))])
        time.sleep(0.3)
for _ in range(3): idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
print("idle:", (poll.__call__ if False else None)); 
r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True); print(r.stdout[-1500:], r.stderr[-300:])
th = threading.Thread(target=poll); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    idx.search_device(B, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr()); n += 1
torch.cuda.synchronize(); dt = time.perf_counter() - t0
stop = True; th.join()
print(f"B={B} route={idx.last_route} {dt / n * 1e3:.3f} ms per search over {n} searches")
for s in samples: print(" | ".join(s))
