"""Run only the split-precision attention of one ViT-B/16 layer (B = 256, L = 205) a few times: target of PMC passes. GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E, _lib
N, L, H = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 205, int(sys.argv[3]) if len(sys.argv) > 3 else 12
d = H * 64
qkv = E.split_pair(torch.randn(N * L, 3 * d, device="cuda"), torch.float16)
out, lse = E.op_attention32_fwd_pair(qkv, N, L, H, False)
dout = E.split_pair(torch.randn(N * L, d, device="cuda"), torch.float16)
dqkv = torch.empty(N * L, 6 * d, device="cuda", dtype=torch.float16)
delta = torch.empty(N * H * L, device="cuda", dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
for _ in range(6):
    _lib.lib.mvlpt_op_attention32_fwd(1, P(qkv), P(out), P(lse), N, L, H, 0, 0, st)
    _lib.lib.mvlpt_op_attention32_bwd(1, P(qkv), P(out), P(dout), P(lse), P(delta), P(dqkv), N, L, H, 0, st)
torch.cuda.synchronize()
