#!/usr/bin/env python3
"""C4-style measurement: N x PQ(m=96, 8-bit) ADC k-NN on one MI355X (wall clock around search_device).
usage: python tools/bench_pq.py [N_millions=100] [nq=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from longbow_amd import _lib, pq
N = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 100_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
M, D, K = 96, 768, 100
lib = _lib.require_gpu(0)
cb = torch.empty((M, 256, D // M), device="cuda")
lib.lb_gpu_fill_uniform_device(0, cb.data_ptr(), cb.numel(), 7, 0, None)
enc = pq.PQEncoder(pq.serialize_codebooks(cb.cpu().numpy()))
if os.environ.get("PQ_REAL") == "1":
    CH = 2_000_000
    buf = torch.empty((CH, D), device="cuda")
    enc.reserve(N)
    t0 = time.perf_counter()
    for r0 in range(0, N, CH):
        c = min(CH, N - r0)
        lib.lb_gpu_fill_uniform_device(0, buf.data_ptr(), c * D, 12345, r0 * D, None)
        enc.add_vectors_device(c, buf.data_ptr())
    torch.cuda.synchronize()
    print(f"encoded {N} x {D} vectors on the GPU in {time.perf_counter()-t0:.2f} s (incl. generating them)", flush=True)
    del buf
else:
    codes = torch.empty((N, M), dtype=torch.uint8, device="cuda")
    lib.lb_gpu_fill_codes_device(0, codes.data_ptr(), codes.numel(), 99, 0, None)
    enc.add_codes_device(N, codes.data_ptr())
    del codes
Q = torch.empty((nq, D), device="cuda")
lib.lb_gpu_fill_uniform_device(0, Q.data_ptr(), Q.numel(), 42, 0, None)
od = torch.empty((nq, K), device="cuda"); ol = torch.empty((nq, K), dtype=torch.int64, device="cuda")
# codes either uniform random bytes (default) or PQ_REAL=1: encoded on the GPU from uniform vectors
for pre, name in ((1, "byte-table prefilter + exact survivors (default)"), (0, "exact f32 table pass"), (1, "byte-table prefilter + exact survivors (default)")):
  enc.set_prefilter(bool(pre))
  ts = []
  for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc.search_device(nq, Q.data_ptr(), K, od.data_ptr(), ol.data_ptr())
    ts.append(time.perf_counter() - t0)
  t = sorted(ts[2:])[len(ts[2:]) // 2]
  res = (ol.cpu().numpy().copy(), od.cpu().numpy().copy())
  if pre == 0: ref = res
  print(f"[{name}] N={N} nq={nq}: {t*1e3:.3f} ms per batch, {t*1e3/nq:.3f} ms/query, codes stream {N*M*nq/t/1e9:.0f} GB/s "
      f"({N*M*nq/t/8e12*100:.1f}% of 8 TB/s)", flush=True)
enc.set_prefilter(True)
print("prefilter == exact pass:", bool(np.array_equal(res[0], ref[0]) and np.array_equal(res[1], ref[1])))
print("top-3", ol[0, :3].tolist(), od[0, :3].tolist())
