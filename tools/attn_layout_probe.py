import sys, os
sys.path.insert(0, "/root/repo")
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
for (N, L, H) in [(256, 197, 12), (3072, 197, 1), (1536, 197, 2), (768, 197, 4)]:
    d = H * 64
    qkv = torch.randn(N * L, 3 * d, device="cuda").half()
    out, lse = E.op_attention_fwd(qkv, N, L, H, False)
    dout = torch.randn(N * L, d, device="cuda").half()
    tf = timeit(lambda: E.op_attention_fwd(qkv, N, L, H, False))
    tb = timeit(lambda: E.op_attention_bwd(qkv, out, dout, lse, N, L, H, False))
    print(f"N={N} L={L} H={H}: fwd {tf*1e3:7.1f} us  bwd {tb*1e3:7.1f} us")
