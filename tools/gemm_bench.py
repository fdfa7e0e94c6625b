"""Time the MFMA GEMM (through the C ABI) on the shapes of the ViT-B/16 bs=256 step. GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E

def bench(M, N, K, epi=0, iters=20, dtype=torch.float16):
    A = torch.randn(M, K, device="cuda").to(dtype)
    Bt = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dtype)
    bias = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda") if epi == 2 else None
    aux = torch.randn(M, N, device="cuda").to(dtype) if epi == 3 else None
    for _ in range(3):
        E.op_gemm(A, Bt, epi, bias=bias, resid=resid, aux=aux)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        E.op_gemm(A, Bt, epi, bias=bias, resid=resid, aux=aux)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return ms, 2.0 * M * N * K / ms / 1e9

shapes = [(50432, 2304, 768, 0), (50432, 768, 768, 2), (50432, 3072, 768, 1), (50432, 768, 3072, 2),
          (7700, 1536, 512, 0), (7700, 512, 512, 2), (7700, 2048, 512, 1), (7700, 512, 2048, 2),
          (7700, 2048, 512, 3), (7700, 512, 1536, 4), (8192, 8192, 8192, 0), (4096, 4096, 4096, 0)]
if len(sys.argv) > 1:      # e.g. "50432,2304,768,0;4096,4096,4096,0" [iters]
    shapes = [tuple(int(x) for x in sh.split(",")) for sh in sys.argv[1].split(";")]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for M, N, K, epi in shapes:
    ms, tf = bench(M, N, K, epi, iters)
    print(f"M={M:6d} N={N:5d} K={K:5d} epi={epi}: {ms*1e3:8.1f} us  {tf:7.1f} TF")
