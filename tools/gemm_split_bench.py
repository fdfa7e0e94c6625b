"""Time the split-operand GEMMs (A = hi|lo pair, 2K MFMA depth) on the step's shapes through the C ABI. GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E


def bench(M, N, K, epi, iters=20, dtype=torch.float16):
    A2 = E.split_pair(torch.randn(M, K, device="cuda"), dtype)
    Bt = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dtype)
    bias = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda") if epi == 2 else None
    aux = torch.randn(M, N, device="cuda").to(dtype) if epi == 6 else None
    for _ in range(3):
        E.op_gemm_split(A2, Bt, epi, bias=bias, resid=resid, aux=aux)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        E.op_gemm_split(A2, Bt, epi, bias=bias, resid=resid, aux=aux)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return ms, 4.0 * M * N * K / ms / 1e9


shapes = [(52480, 2304, 768, 7), (52480, 768, 768, 2), (52480, 3072, 768, 5), (52480, 768, 3072, 2), (52480, 3072, 768, 6), (52480, 768, 2304, 4),
          (7700, 1536, 512, 7), (7700, 512, 512, 2), (7700, 2048, 512, 5), (7700, 512, 2048, 2), (7700, 2048, 512, 6), (7700, 512, 1536, 4),
          (8192, 8192, 4096, 4)]
for M, N, K, epi in shapes:
    ms, tf = bench(M, N, K, epi)
    print(f"M={M:6d} N={N:5d} K={K:5d} (pair: 2K) epi={epi}: {ms*1e3:8.1f} us  {tf:7.1f} TF executed")
