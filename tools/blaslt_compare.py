"""Measurement aid only (never on the product path): time the library GEMM PyTorch-ROCm dispatches to (hipBLASLt /
rocBLAS) on the step's shapes next to this repo's kernel with the plain 16-bit-store epilogue.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

for M, N, K in [(50432, 2304, 768), (50432, 768, 768), (50432, 3072, 768), (50432, 768, 3072), (8192, 8192, 8192), (74368, 4096, 1024), (74368, 1024, 4096)]:
    A = torch.randn(M, K, device="cuda").half()
    Bt = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    t_lib = timeit(lambda: torch.matmul(A, Bt.t(), out=out))
    t_own = timeit(lambda: E.op_gemm(A, Bt, 0, bias=bias))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: library {t_lib*1e3:7.1f} us ({fl/t_lib/1e9:6.1f} TF)   this repo {t_own*1e3:7.1f} us ({fl/t_own/1e9:6.1f} TF)")
