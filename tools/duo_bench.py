"""gemm_duo.hip against the 256x256 kernel of gemm.hip: run once per setting of MVLPT_GEMM_DUO (read once per process).
Checks every output against an fp32 torch GEMM and times the launch.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mvlpt_amd import engine as E

L = E._lib


def run(M, N, K, epi, iters=20, dtype=torch.float16, check=True):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    Bt = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dtype)
    bias = torch.randn(N, device="cuda", generator=g)
    out2 = epi == L.EPI_GELU
    res = E.op_gemm(A, Bt, epi, bias=bias, out2=out2)
    err = -1.0
    if check:
        ref = A.float() @ Bt.float().t() + bias
        if epi == L.EPI_GELU:
            out, u = res
            e1 = float((u.float() - ref).abs().max()) / float(ref.abs().max())
            want = ref * torch.sigmoid(1.702 * ref)
            e2 = float((out.float() - want).abs().max()) / float(want.abs().max())
            err = max(e1, e2)
        else:
            err = float((res.float() - ref).abs().max()) / float(ref.abs().max())
        del ref
    for _ in range(3):
        E.op_gemm(A, Bt, epi, bias=bias, out2=out2)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        E.op_gemm(A, Bt, epi, bias=bias, out2=out2)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return ms, 2.0 * M * N * K / ms / 1e9, err


shapes = [(50432, 2304, 768, 0), (50432, 3072, 768, 1), (50432, 768, 3072, 0), (50432, 768, 768, 0), (12608, 2304, 768, 0),
          (50000, 2304, 768, 0), (33333, 3072, 768, 1), (148736, 3072, 1024, 1), (8192, 8192, 8192, 0)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in sh.split(",")) for sh in sys.argv[1].split(";")]
print("MVLPT_GEMM_DUO =", os.environ.get("MVLPT_GEMM_DUO", "0"), " E_STORE", os.environ.get("MVLPT_DUO_E_STORE", "-"),
      " E_GELU", os.environ.get("MVLPT_DUO_E_GELU", "-"))
bad = False
for M, N, K, epi in shapes:
    ms, tf, err = run(M, N, K, epi)
    flag = "" if err < 2e-3 else "   <-- WRONG"
    bad = bad or err >= 2e-3
    print(f"M={M:6d} N={N:5d} K={K:5d} epi={epi}: {ms*1e3:8.1f} us  {tf:7.1f} TF   err {err:.2e}{flag}")
sys.exit(1 if bad else 0)
