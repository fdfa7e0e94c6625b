import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import clip_oracle as O
from mvlpt_amd.weights import ARCHS, make_state_dict
arch = ARCHS["ViT-B/16"]; sd = make_state_dict(arch, 1)
img = torch.randn(16, 3, 224, 224)
print("cpu count", os.cpu_count())
for th in (128, 64, 32, 16):
    torch.set_num_threads(th)
    with torch.no_grad():
        O.image_encoder_fwd(sd, img[:4], None, None, heads=arch.vision_heads, need_bwd=False)
        t0 = time.perf_counter()
        O.image_encoder_fwd(sd, img, None, None, heads=arch.vision_heads, need_bwd=False)
        print(th, "threads: image fwd B=16", round(time.perf_counter() - t0, 2), "s")
