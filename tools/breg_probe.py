"""A/B of the experimental "B fragments in registers" GEMM (debug build -DMVLPT_BREG) against the shipped 256x256
kernel, on the image tower's shapes through the op ABI: correctness against an fp32 torch matmul on the device, then event-timed launches.
Run once per setting (the switch is read once per process):
  MVLPT_HIP_LIB=$PWD/mvlpt_amd/libvar_breg.so MVLPT_GEMM_BREG={0,1} python tools/breg_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E

L = E._lib
shapes = [(50432, 2304, 768, L.EPI_STORE16, "QKV, plain store"), (50432, 3072, 768, L.EPI_GELU, "MLP up + GELU"),
          (50432, 768, 3072, L.EPI_RESID32, "MLP down + fp32 residual"), (50432, 2304, 768, L.EPI_STORE32, "QKV, fp32 store")]
print("MVLPT_GEMM_BREG =", os.environ.get("MVLPT_GEMM_BREG", "0"))
for M, N, K, epi, name in shapes:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).half()
    Bt = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g) if epi == L.EPI_RESID32 else None
    ref = A.float() @ Bt.float().t() + bias
    if resid is not None: ref = ref + resid
    if epi == L.EPI_GELU: ref = ref * torch.sigmoid(1.702 * ref)
    run = lambda: E.op_gemm(A, Bt, epi, bias=bias, resid=resid)
    out = run()
    out = out[0] if isinstance(out, tuple) else out
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    outs = [run() for _ in range(5)]
    same = all(torch.equal((o[0] if isinstance(o, tuple) else o), out) for o in outs)
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9; tot = 0.0
    for rep in range(5):
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 20 * 1e3
        best = min(best, t); tot += t
    print(f"{name:28s} M={M} N={N} K={K}: rel err {err:.2e} bit-stable {same}  avg {tot/5:7.1f} us  best {best:7.1f} us  {2.0*M*N*K/best/1e6:7.1f} TF/s")
