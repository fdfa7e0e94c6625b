"""Host time to ENQUEUE each tower (no device sync in between) vs its device time: is the step's critical path a host launch loop?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd.model import FrozenCLIP, build_prompt_layout
from mvlpt_amd.weights import ARCHS, make_state_dict
arch = ARCHS["ViT-B/16"]
eng = FrozenCLIP(make_state_dict(arch, 1)).engine
x = torch.randn(256, 3, 224, 224, device="cuda").half()
C, L, n = 100, 77, 16
nl = [1 + (i % 3) for i in range(C)]
layout = build_prompt_layout(nl, n, L, "middle").cuda()
eot = torch.tensor([n + v + 2 for v in nl], dtype=torch.int32).cuda()
pre, suf, ctx = torch.randn(C, 1, 512, device="cuda") * 0.02, torch.randn(C, L - 1 - n, 512, device="cuda") * 0.02, torch.randn(n, 512, device="cuda") * 0.02
dfeat = torch.randn(C, 512, device="cuda") * 1e-3
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    h = d = 0.0
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        h += t1 - t0; d += t2 - t0
    return h / reps * 1e3, d / reps * 1e3
for name, fn in (("image_fwd", lambda: eng.image_fwd(x)),
                 ("text_fwd", lambda: eng.text_fwd(pre, suf, ctx, layout, eot, save_for_bwd=True)),
                 ("text_bwd", lambda: eng.text_bwd(dfeat))):
    h, d = timed(fn)
    print(f"{name:10s}: host enqueue {h:6.3f} ms, enqueue + device {d:6.3f} ms")
