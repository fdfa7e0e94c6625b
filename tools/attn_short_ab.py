"""Output fingerprint + timing of the pair-attention backward (default: the text tower's: causal, L = 77, 8 heads, 100 sequences, mixed pairs; or N L H causal) for A/B runs of
two libraries (MVLPT_HIP_LIB): same seed -> the sha256 of dqkv must agree when a change is meant to be bit-neutral."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E
N, L, H, CAUSAL = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (100, 77, 8, 1)
d = H * 64
torch.manual_seed(1)
qkv = E.split_pair(torch.randn(N * L, 3 * d, device="cuda"), torch.float16)
out, lse = E.op_attention32_fwd_mixed(qkv, N, L, H, bool(CAUSAL))
dout = E.split_pair(torch.randn(N * L, d, device="cuda"), torch.float16)
dq = E.op_attention32_bwd_mixed(qkv, out, dout, lse, N, L, H, bool(CAUSAL))
torch.cuda.synchronize()
fp = hashlib.sha256(dq.cpu().numpy().tobytes()).hexdigest()[:16]
fpo = hashlib.sha256(out.cpu().numpy().tobytes() + lse.cpu().numpy().tobytes()).hexdigest()[:16]
dqkv = torch.zeros_like(dq); delta = torch.empty(N * H * L, device="cuda")
from mvlpt_amd import _lib
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
run = lambda: _lib.lib.mvlpt_op_attention32_bwd_mixed(1, P(qkv), P(out), P(dout), P(lse), P(delta), P(dqkv), N, L, H, CAUSAL, st)
for _ in range(10): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(7):
    s.record()
    for _ in range(100 if L < 100 else 10): run()
    e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) * (10 if L < 100 else 100))
o2 = torch.zeros_like(out); l2 = torch.zeros_like(lse)
runf = lambda: _lib.lib.mvlpt_op_attention32_fwd_mixed(1, P(qkv), P(o2), P(l2), N, L, H, CAUSAL, 0, st)
for _ in range(10): runf()
torch.cuda.synchronize()
tf = []
for _ in range(7):
    s.record()
    for _ in range(100 if L < 100 else 10): runf()
    e.record(); torch.cuda.synchronize()
    tf.append(s.elapsed_time(e) * (10 if L < 100 else 100))
print(f"   forward: out+lse sha256 {fpo}  {min(tf):.1f} us best, {sorted(tf)[3]:.1f} median   (MVLPT_ATTN32T_PERSIST={os.environ.get('MVLPT_ATTN32T_PERSIST', '1')})")
print(f"{os.path.basename(os.environ.get('MVLPT_HIP_LIB', 'libmvlpt_hip.so'))}: dqkv sha256 {fp}  backward {min(ts):.1f} us best, {sorted(ts)[3]:.1f} median (N = {N}, L = {L}, H = {H}, causal = {CAUSAL})")
