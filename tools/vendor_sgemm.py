import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
for (M, N, K) in ((1024, 131072, 768), (1024, 262144, 768), (4096, 4096, 4096), (8192, 8192, 768)):
    A = torch.rand((M, K), device=dev); B = torch.rand((N, K), device=dev)
    C = torch.empty((M, N), device=dev)
    for _ in range(3): torch.matmul(A, B.t(), out=C)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 10
    for _ in range(it): torch.matmul(A, B.t(), out=C)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / it
    print(f"torch f32 matmul {M}x{N}x{K}: {t*1e3:.3f} ms = {2.0*M*N*K/t/1e12:.1f} TFLOP/s", flush=True)
