"""Packed-residual-stream GEMM epilogue (EPI_RESIDP_LN) on the image tower's two shapes, through the op ABI: fingerprints of the new
planes and of the row partials + event-timed launches, for A/B runs of two libraries (MVLPT_HIP_LIB)."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E
name = os.path.basename(os.environ.get("MVLPT_HIP_LIB", "libmvlpt_hip.so"))
for M, N, K, what in [(50432, 768, 768, "out-projection"), (50432, 768, 3072, "MLP down")]:
    g = torch.Generator(device="cuda").manual_seed(K)
    A = torch.randn(M, K, device="cuda", generator=g).half()
    Bt = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    x = torch.randn(M, N, device="cuda", generator=g) * 3
    hi = x.half()
    lo = torch.zeros(M, N, device="cuda", dtype=torch.int8)
    h2, l2, part, nt = E.op_gemm_residp(A, Bt, bias, hi, lo)
    torch.cuda.synchronize()
    fp = hashlib.sha256(h2.cpu().numpy().tobytes() + l2.cpu().numpy().tobytes()).hexdigest()[:12]
    ref = (A.float() @ Bt.float().t() + bias + hi.float())
    got = h2.float() + l2.float() * 0  # hi plane alone is within fp16 rounding of the sum
    err = float((got - ref).abs().max() / ref.abs().max())
    rs = ref.double().sum(1); ps = part[:, :nt, 0].double().sum(1)
    serr = float((rs - ps).abs().max() / rs.abs().max())
    run = lambda: E.op_gemm_residp(A, Bt, bias, hi, lo)
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 20 * 1e3)
    print(f"{name}: {what:15s} planes sha256 {fp}  hi-plane err {err:.1e}  row-sum err {serr:.1e}  {min(ts):6.1f} us best {sorted(ts)[2]:6.1f} median (incl. the allocation of its outputs)")
