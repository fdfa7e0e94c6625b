"""Timeline of workgroup 0 of one GEMM launch (debug build with -DMVLPT_GEMM_TRACE, loaded through MVLPT_HIP_LIB).
Usage on the GPU box:  MVLPT_HIP_LIB=$PWD/mvlpt_amd/libvar_trace.so python tools/gemm_trace.py M N K epi
Points: 1 stage start, 2 after a DMA issue, 3 before the mid-stage DMA issue, 4 after the MFMA groups, 5 after the
vmcnt wait, 7 after the closing barrier, 8 epilogue start, 9 epilogue end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVLPT_GEMM_TRACE_FILE"] = "/tmp/gemm_trace.bin"
import numpy as np
import torch
from mvlpt_amd import engine as E

M, N, K, epi = [int(v) for v in sys.argv[1:5]]
A = torch.randn(M, K, device="cuda").half()
Bt = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
bias = torch.randn(N, device="cuda")
resid = torch.randn(M, N, device="cuda") if epi == 2 else None
for _ in range(3):
    E.op_gemm(A, Bt, epi, bias=bias, resid=resid)
torch.cuda.synchronize()
raw = np.fromfile("/tmp/gemm_trace.bin", dtype=np.int64).reshape(16, 2048)
waves = []
for w in range(16):
    r = raw[w][raw[w] != 0]
    if len(r) == 0:
        continue
    waves.append((w, (r >> 56) & 0xff, r & ((1 << 56) - 1)))
t0 = min(t[0] for _, _, t in waves)
print(f"{len(waves)} waves traced, {len(waves[0][1])} records each")
# per-wave breakdown of a steady-state stage (skip the first tile)
for w, p, t in waves:
    t = t - t0
    seg = {}
    last_p, last_t = None, None
    n_stage = 0
    for pi, ti in zip(p, t):
        if last_p is not None:
            key = (int(last_p), int(pi))
            seg.setdefault(key, []).append(int(ti - last_t))
        if pi == 1:
            n_stage += 1
        last_p, last_t = pi, ti
    tot = int(t[-1] - t[0])
    desc = "  ".join(f"{a}->{b}: {np.mean(v):7.0f} x{len(v)}" for (a, b), v in sorted(seg.items()))
    print(f"wave {w}: {n_stage} stages, {tot} ticks total, {tot / max(n_stage, 1):.0f} per stage | {desc}")

# vmcnt-wait (4->5) of the FIRST stage after an epilogue vs the other stages (is the wait draining the epilogue's stores?)
for w, p, t in waves[:1] + waves[4:5]:
    first, rest, after_epi = [], [], False
    for k in range(1, len(p)):
        if p[k - 1] == 9:
            after_epi = True
        if p[k - 1] == 4 and p[k] == 5:
            (first if after_epi else rest).append(int(t[k] - t[k - 1]))
            after_epi = False
    if first and rest:
        print(f"wave {w}: vmcnt wait of a tile's first stage {np.mean(first):.0f} ticks (n={len(first)}) vs {np.mean(rest):.0f} elsewhere")
