"""Time the image tower forward alone (ViT-B/16, batch 256, no prompts, no backward) through the engine. GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd.model import FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict
arch = ARCHS[sys.argv[1] if len(sys.argv) > 1 else "ViT-B/16"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eng = FrozenCLIP(make_state_dict(arch, 1)).engine
x = torch.randn(B, 3, arch.image_resolution, arch.image_resolution, device="cuda").half()
for _ in range(3): eng.image_fwd(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): eng.image_fwd(x)
torch.cuda.synchronize()
print(f"image tower fwd {arch.name if hasattr(arch, 'name') else ''} B={B}: {(time.perf_counter()-t0)/20*1e3:.3f} ms")
