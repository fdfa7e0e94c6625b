"""Timeline of ONE workgroup (block 1024: mid-launch, chip busy) of the streamed pair-attention forward (debug build with
-DMVLPT_ATTN_TRACE, loaded through MVLPT_HIP_LIB).  Usage on the GPU box:
    MVLPT_HIP_LIB=$PWD/mvlpt_amd/libvar_attn_trace.so python tools/attn_trace.py [N L H]
Points: 10 kernel start, 11 own rows requested / first chunk issued, per chunk 1 loop top, 2 after the slot-free barrier, 3 after the
DMA issue of the next chunk, 4 after the vmcnt wait, 5 after the data barrier, 6 after S = Q.K^T (3 terms), 7 after the softmax,
8 after O += P.V (incl. the split of P), 12 epilogue start, 13 end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVLPT_ATTN_TRACE_FILE"] = "/tmp/attn_trace.bin"
import numpy as np
import torch
from mvlpt_amd import engine as E, _lib
BWD = len(sys.argv) > 1 and sys.argv[1] == "bwd"
if BWD:
    sys.argv.pop(1)
N, L, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 and sys.argv[1] != "short" else (256, 205, 12)
if BWD:
    # the resident pair-attention BACKWARD (attn32r_bwd_kernel, cfg3's first kernel): workgroup (head 5, image N/2).  Points: 20 start,
    # 21 K / V staging + own delta rows requested, 22 landed + barrier; phase A per own query tile: 23 own Q / dO pairs loaded, 24 S, dP,
    # exp, dS of 13 key tiles done, 25 dQ accumulated, 26 dQ stored; 27 barrier behind phase A, 28 Q / dO staging requested, 29 landed +
    # barrier; phase B per own key tile: 30 own K / V pairs loaded, 31 S, dP, P, dS of 13 query tiles, 32 dV accumulated, 33 dK
    # accumulated, 34 stored; 35 end
    d = H * 64
    qkv = E.split_pair(torch.randn(N * L, 3 * d, device="cuda"), torch.float16)
    out, lse = E.op_attention32_fwd_pair(qkv, N, L, H, False)
    dout = E.split_pair(torch.randn(N * L, d, device="cuda"), torch.float16)
    dqkv = torch.empty(N * L, 6 * d, device="cuda", dtype=torch.float16)
    delta = torch.empty(N * H * L, device="cuda", dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.lib.mvlpt_op_attention32_bwd(1, qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), N, L, H, 0, st)
    torch.cuda.synchronize()
    raw = np.fromfile("/tmp/attn_trace.bin", dtype=np.int64).reshape(8, 256)
    names = {(20, 21): "prologue: K / V staging requests, delta of the own rows", (21, 22): "wait for K, V + barrier",
             (22, 23): "phase A: own Q / dO pairs from memory", (26, 23): "phase A: own Q / dO pairs from memory (2nd tile)",
             (23, 24): "phase A: S, dP (78 MFMAs), exp, dS of 13 key tiles", (24, 25): "phase A: dQ = dS K (39 x 3 MFMAs)", (25, 26): "phase A: dQ stores",
             (26, 27): "barrier behind phase A (waits for the slowest wave)", (22, 27): "no phase-A tile", (27, 28): "Q / dO staging requests", (28, 29): "wait for Q, dO + barrier",
             (29, 30): "phase B: own K / V pairs from memory", (34, 30): "phase B: own K / V pairs from memory (2nd tile)",
             (30, 31): "phase B: S, dP (78 MFMAs), exp, P, dS of 13 query tiles", (31, 32): "phase B: dV = P^T dO", (32, 33): "phase B: dK = dS^T Q",
             (33, 34): "phase B: dK / dV stores", (34, 35): "end"}
    for w in range(7):
        r = raw[w][raw[w] != 0]
        if len(r) == 0:
            continue
        p, t = (r >> 56) & 0xff, r & ((1 << 56) - 1)
        tot = int(t[-1] - t[0])
        print(f"wave {w}: {tot} ticks (s_memtime; ~ shader cycles on this box: 12 heads per CU x this = the launch)")
        for k in range(1, len(p)):
            key = (int(p[k - 1]), int(p[k]))
            print(f"    {names.get(key, str(key)):62s} {int(t[k] - t[k - 1]):7d} ticks  ({100 * (t[k] - t[k - 1]) / tot:4.1f} %)")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "short":
    # the text tower's causal backward (attn32t_bwd_kernel, L <= 80): workgroup (head 5, sequence N/2).  Points: 40 start, 41 K / V staging +
    # own rows + delta requested, 42 landed + barrier, 43 phase A: S, dP, exp, dS of the own query tile, 44 dQ accumulated, 45 dQ stores
    # issued, 46 barrier behind phase A, 47 Q / dO staging requested, 48 landed + barrier, 49 phase B: S, dP, P, dS, 50 dV / dK accumulated, 51 end
    N, L, H = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (100, 77, 8)
    d = H * 64
    qkv = E.split_pair(torch.randn(N * L, 3 * d, device="cuda"), torch.float16)
    out, lse = E.op_attention32_fwd_pair(qkv, N, L, H, True)
    dout = E.split_pair(torch.randn(N * L, d, device="cuda"), torch.float16)
    dqkv = torch.empty(N * L, 6 * d, device="cuda", dtype=torch.float16)
    delta = torch.empty(N * H * L, device="cuda", dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.lib.mvlpt_op_attention32_bwd(1, qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), N, L, H, 1, st)
    torch.cuda.synchronize()
    raw = np.fromfile("/tmp/attn_trace.bin", dtype=np.int64).reshape(8, 256)
    names = {(40, 41): "requests: K / V staging, own rows, delta", (41, 42): "wait + barrier", (42, 43): "phase A: S, dP, exp, dS (own query tile)",
             (43, 44): "phase A: dQ accumulation", (44, 45): "phase A: dQ stores issued", (45, 46): "barrier behind phase A", (42, 45): "no phase-A tile",
             (46, 47): "Q / dO staging requests", (47, 48): "wait + barrier", (48, 49): "phase B: S, dP, P, dS (own key tile)", (49, 50): "phase B: dV, dK accumulation",
             (50, 51): "phase B: stores issued", (48, 51): "no phase-B tile"}
    t0 = min(int(raw[w][raw[w] != 0][0] & ((1 << 56) - 1)) for w in range(5))
    for w in range(5):
        r = raw[w][raw[w] != 0]
        p, t = (r >> 56) & 0xff, r & ((1 << 56) - 1)
        print(f"wave {w}: {int(t[-1] - t[0])} ticks from its start to its end (s_memtime, 100 MHz x 21 ~ shader cycles?)  start +{int(t[0]) - t0}")
        for k in range(1, len(p)):
            print(f"    {names.get((int(p[k - 1]), int(p[k])), str((int(p[k - 1]), int(p[k])))):50s} {int(t[k] - t[k - 1]):7d} ticks")
    sys.exit(0)
d = H * 64
qkv = E.split_pair(torch.randn(N * L, 3 * d, device="cuda"), torch.float16)
out = torch.zeros(N * L, 2 * d, device="cuda", dtype=torch.float16)
lse = torch.zeros(N * H * L, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _lib.lib.mvlpt_op_attention32_fwd(1, qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), N, L, H, 0, 0, st)
torch.cuda.synchronize()
raw = np.fromfile("/tmp/attn_trace.bin", dtype=np.int64).reshape(8, 256)
names = {(10, 11): "prologue (Q loads, DMA chunk 0)", (11, 1): "to loop", (1, 2): "slot-free barrier", (2, 3): "DMA issue", (3, 4): "vmcnt wait",
         (2, 4): "vmcnt wait (last)", (4, 5): "data barrier", (5, 6): "S = Q.K^T", (6, 7): "softmax", (7, 8): "P split + P.V", (8, 1): "loop",
         (8, 12): "to epilogue", (12, 13): "epilogue"}
for w in range(8):
    r = raw[w][raw[w] != 0]
    if len(r) == 0:
        continue
    p, t = (r >> 56) & 0xff, r & ((1 << 56) - 1)
    seg = {}
    for k in range(1, len(p)):
        seg.setdefault((int(p[k - 1]), int(p[k])), []).append(int(t[k] - t[k - 1]))
    tot = int(t[-1] - t[0])
    print(f"wave {w}: {tot} ticks total (100 MHz ticks x ~20 = shader cycles? see below)")
    for key, v in seg.items():
        print(f"    {names.get(key, str(key)):34s} {np.sum(v):7d} ticks  ({100 * np.sum(v) / tot:4.1f} %)  per occurrence {np.mean(v):7.0f} x{len(v)}")
