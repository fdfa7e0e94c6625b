"""Stage timeline of workgroup 0 of one gemm_duo launch (debug build -DMVLPT_GEMM_TRACE through MVLPT_HIP_LIB, MVLPT_GEMM_DUO=1).
Usage on the GPU box:  MVLPT_GEMM_DUO=1 MVLPT_HIP_LIB=$PWD/mvlpt_amd/libvar_trace.so python tools/duo_trace.py M N K epi
Points: 1 / 2 / 3 = behind the closing barrier of a stage the wave spent multiplying / in its epilogue / idle; 4 = in front of its vmcnt wait."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVLPT_GEMM_TRACE_FILE"] = "/tmp/duo_trace.bin"
import numpy as np
import torch
from mvlpt_amd import engine as E

M, N, K, epi = [int(v) for v in sys.argv[1:5]]
A = torch.randn(M, K, device="cuda").half()
Bt = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
bias = torch.randn(N, device="cuda")
for _ in range(3):
    E.op_gemm(A, Bt, epi, bias=bias)
torch.cuda.synchronize()
raw = np.fromfile("/tmp/duo_trace.bin", dtype=np.int64).reshape(16, 2048)
W = {}
for w in (0, 4):
    r = raw[w][raw[w] != 0]
    W[w] = ((r >> 56) & 0xff, r & ((1 << 56) - 1))
# stage ends (points 1-3) of wave 0 and wave 4 are the same barriers: classify every stage by the pair of roles
def stages(w):
    p, t = W[w]
    ends = [(int(pi), int(ti)) for pi, ti in zip(p, t) if pi in (1, 2, 3)]
    waits = {}
    last4 = None
    k = 0
    for pi, ti in zip(p, t):
        if pi == 4:
            last4 = int(ti)
        elif pi in (1, 2, 3):
            waits[k] = (int(ti) - last4) if last4 is not None else 0
            last4 = None
            k += 1
    return ends, waits
e0, w0 = stages(0)
e4, w4 = stages(4)
n = min(len(e0), len(e4))
print(f"{n} stages traced; total {e0[n-1][1] - e0[0][1]} ticks")
names = {1: "mul", 2: "epi", 3: "idle"}
acc = {}
for k in range(1, n):
    key = (names[e0[k][0]], names[e4[k][0]])
    acc.setdefault(key, []).append(e0[k][1] - e0[k - 1][1])
for key, v in sorted(acc.items()):
    print(f"group0 {key[0]:4s} group1 {key[1]:4s}: {np.mean(v):7.0f} ticks x{len(v)}  (min {min(v)}, max {max(v)})")
for w, (e, wt) in ((0, (e0, w0)), (4, (e4, w4))):
    byrole = {}
    for k in range(1, n):
        byrole.setdefault(names[e[k][0]], []).append(wt.get(k, 0))
    print(f"wave {w}: ticks from 'before vmcnt wait' to 'behind the barrier' by role: " + ", ".join(f"{r} {np.mean(v):.0f}" for r, v in sorted(byrole.items())))
seq = " ".join(f"{names[e0[k][0]][0]}{names[e4[k][0]][0]}:{e0[k][1]-e0[k-1][1]}" for k in range(1, min(n, 45)))
print("first stages (group0 group1 : ticks):", seq)
