"""Sustained library GEMM (torch.matmul -> hipBLASLt) for tools/power_probe.sh: M N K iters.  GPU box only."""
import sys, time, torch
M, N, K, iters = (int(x) for x in sys.argv[1:5])
a = torch.randn(M, K, device="cuda", dtype=torch.float16)
b = torch.randn(N, K, device="cuda", dtype=torch.float16)
for _ in range(10):
    torch.matmul(a, b.t())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    torch.matmul(a, b.t())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"library GEMM M={M} N={N} K={K}: {dt*1e6:.1f} us  {2.0*M*N*K/dt/1e12:.1f} TF/s")
