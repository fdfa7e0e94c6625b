"""Time LayerNorm fwd/bwd (through the C ABI) on the step's shapes. GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

for rows, d in [(50432, 768), (52480, 768), (7700, 512), (74368, 1024)]:
    x = torch.randn(rows, d, device="cuda")
    g, b = torch.randn(d, device="cuda"), torch.randn(d, device="cuda")
    dy = torch.randn(rows, d, device="cuda").half()
    tf = timeit(lambda: E.op_layernorm_fwd(x, g, b, torch.float16))
    tb = timeit(lambda: E.op_layernorm_bwd(dy, x, g, resid=x, want16=True))
    bf, bb = rows * d * 6, rows * d * (4 + 2 + 4 + 4 + 2)
    print(f"rows={rows} d={d}: fwd {tf*1e3:6.1f} us ({bf/tf/1e9:5.2f} TB/s)  bwd {tb*1e3:6.1f} us ({bb/tb/1e9:5.2f} TB/s)")
