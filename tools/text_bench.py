"""Time the text tower fwd+bwd alone (C=100, L=77, ViT-B/16 text width 512) through the engine. GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd.model import FrozenCLIP, build_prompt_layout
from mvlpt_amd.weights import ARCHS, make_state_dict
arch = ARCHS["ViT-B/16"]
clip = FrozenCLIP(make_state_dict(arch, 1), precision=os.environ.get("PREC", "split_grad"))
eng = clip.engine
C, L, n = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 77, 16
nl = [1 + (i % 3) for i in range(C)]
layout = build_prompt_layout(nl, n, L, "middle").cuda()
eot = torch.tensor([n + x + 2 for x in nl], dtype=torch.int32).cuda()
pre = torch.randn(C, 1, 512, device="cuda") * 0.02
suf = torch.randn(C, L - 1 - n, 512, device="cuda") * 0.02
ctx = torch.randn(n, 512, device="cuda") * 0.02
dfeat = torch.randn(C, 512, device="cuda") * 1e-3
def run():
    eng.text_fwd(pre, suf, ctx, layout, eot, save_for_bwd=True)
    eng.text_bwd(dfeat)
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize()
print(f"text tower C={C} L={L}: fwd+bwd {(time.perf_counter()-t0)/20*1e3:.3f} ms")
def runf():
    eng.text_fwd(pre, suf, ctx, layout, eot, save_for_bwd=True)
t0 = time.perf_counter()
for _ in range(20): runf()
torch.cuda.synchronize()
print(f"   fwd only {(time.perf_counter()-t0)/20*1e3:.3f} ms")
