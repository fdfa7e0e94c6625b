"""How much of the MLP-up GEMM is its QuickGELU epilogue?  Same shape with EPI_STORE16 / EPI_GELU (one output) / EPI_GELU + saved u."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E
L = E._lib
M, N, K = 50432, 3072, 768
A = torch.randn(M, K, device="cuda").half(); Bt = (torch.randn(N, K, device="cuda") * K ** -0.5).half(); bias = torch.randn(N, device="cuda")
def t(epi, out2):
    for _ in range(3): E.op_gemm(A, Bt, epi, bias=bias, out2=out2)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): E.op_gemm(A, Bt, epi, bias=bias, out2=out2)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 20 * 1e3
for r in range(2):
    print(f"store16 {t(L.EPI_STORE16, False):.1f} us | gelu {t(L.EPI_GELU, False):.1f} us | gelu + u {t(L.EPI_GELU, True):.1f} us")
