"""Time attention fwd/bwd kernels (through the C ABI) on the step's shapes. GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

for (N, L, H, causal) in [(256, 197, 12, False), (256, 205, 12, False), (256, 50, 12, False), (100, 77, 8, True), (1000, 77, 8, True),
                          (128, 261, 16, False), (128, 581, 16, False)]:
    d = H * 64
    qkv = torch.randn(N * L, 3 * d, device="cuda").half()
    out, lse = E.op_attention_fwd(qkv, N, L, H, causal)
    dout = torch.randn(N * L, d, device="cuda").half()
    tf = timeit(lambda: E.op_attention_fwd(qkv, N, L, H, causal))
    tb = timeit(lambda: E.op_attention_bwd(qkv, out, dout, lse, N, L, H, causal))
    fl = 4.0 * L * L * 64 * N * H * (0.5 if causal else 1.0)
    print(f"N={N} L={L} H={H} causal={causal}: fwd {tf*1e3:7.1f} us ({fl/tf/1e9:6.1f} TF)  bwd {tb*1e3:7.1f} us ({2.5*fl/tb/1e9:6.1f} TF)")

print("split-precision attention (hi|lo pair operands, three-term products; kernels only, outputs preallocated):")
from mvlpt_amd import _lib
for (N, L, H, causal) in [(256, 205, 12, False), (100, 77, 8, True), (2191, 77, 8, True), (128, 581, 16, False)]:
    d = H * 64
    qkv = E.split_pair(torch.randn(N * L, 3 * d, device="cuda"), torch.float16)
    out, lse = E.op_attention32_fwd_pair(qkv, N, L, H, causal)
    dout = E.split_pair(torch.randn(N * L, d, device="cuda"), torch.float16)
    dqkv = torch.empty(N * L, 6 * d, device="cuda", dtype=torch.float16)
    delta = torch.empty(N * H * L, device="cuda", dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    tf = timeit(lambda: _lib.lib.mvlpt_op_attention32_fwd(1, P(qkv), P(out), P(lse), N, L, H, int(causal), 0, st))
    tb = timeit(lambda: _lib.lib.mvlpt_op_attention32_bwd(1, P(qkv), P(out), P(dout), P(lse), P(delta), P(dqkv), N, L, H, int(causal), st))
    fl = 4.0 * L * L * 64 * N * H * (0.5 if causal else 1.0)
    print(f"N={N} L={L} H={H} causal={causal}: fwd {tf*1e3:7.1f} us ({fl/tf/1e9:6.1f} TF)  bwd {tb*1e3:7.1f} us ({2.5*fl/tb/1e9:6.1f} TF)")
