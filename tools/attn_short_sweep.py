"""Text-tower attention (causal, L = 77, 8 heads, mixed pairs as the engine runs it) over the number of sequences: how much of a launch is
the ragged last round of workgroups (100 classes x 8 heads = 800 workgroups on 768 resident slots)?  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E

L, H = 77, 8
d = H * 64
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / it * 1e3)
    return best
for N in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "64,90,96,100,104,128,160,192,200".split(","))]:
    qkv = E.split_pair(torch.randn(N * L, 3 * d, device="cuda"), torch.float16)
    out, lse = E.op_attention32_fwd_mixed(qkv, N, L, H, True)
    dout = E.split_pair(torch.randn(N * L, d, device="cuda"), torch.float16)
    f = t(lambda: E.op_attention32_fwd_mixed(qkv, N, L, H, True))
    b = t(lambda: E.op_attention32_bwd_mixed(qkv, out, dout, lse, N, L, H, True))
    print(f"N = {N:4d} sequences ({N * H:5d} workgroups): forward {f:6.1f} us  backward {b:6.1f} us   (includes torch.zeros of the outputs)")
