"""Time the mixed-pair GEMM (through the C ABI) on given shapes: python tools/gemm_mixed_bench.py "M,N,K,epi;..." [iters]  (epi 2 / 4 / 5 / 6 / 7)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E
shapes = [tuple(int(x) for x in sh.split(",")) for sh in (sys.argv[1] if len(sys.argv) > 1 else "7700,512,2048,2;7700,512,512,2;7700,1536,512,7;7700,2048,512,5").split(";")]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for M, N, K, epi in shapes:
    A2 = E.op_cast_mixed(torch.randn(M, K, device="cuda"), torch.float16)
    Wp, e8 = E.op_pack_weight_mixed((torch.randn(N, K, device="cuda") * K ** -0.5).half().float(), torch.float16)
    resid = torch.randn(M, N, device="cuda") if epi == 2 else None
    aux = torch.randn(M, N, device="cuda").half() if epi == 6 else None
    f = lambda: E.op_gemm_mixed(A2, Wp, e8, epi, resid=resid, aux=aux)
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"M={M:6d} N={N:5d} K={K:5d} epi={epi}: {ms * 1e3:7.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF algorithmic (output allocation included)")
