"""Bit-stability of the text tower (forward + backward: features and the context gradient) while another stream runs the image tower of the
same engine.  Usage (GPU box): [ITERS=n] python tools/text_determinism_probe.py image_batch [precision]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd.class_prompts import load_class_prompts
from mvlpt_amd.config import get_cfg_default
from mvlpt_amd.model import CustomCLIP, FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict
B = int(sys.argv[1]); prec = sys.argv[2] if len(sys.argv) > 2 else "split_grad"
ITERS = int(os.environ.get("ITERS", "40"))
arch = ARCHS["ViT-B/16"]
cfg = get_cfg_default(); cfg.TRAINER.MVLPT.COOP.N_CTX = 16
pre, C = load_class_prompts("caltech101", 16)
torch.manual_seed(0)
model = CustomCLIP(cfg, ["c"] * C, FrozenCLIP(make_state_dict(arch, 3), "fp16", precision=prec), pretokenized=pre).cuda()
pl, eng = model.prompt_learner, model.engine
ctx = pl.ctx.detach()
dfeat = torch.randn(C, arch.embed_dim, device="cuda") * 1e-3
x = torch.randn(B, 3, 224, 224, device="cuda").half()
def text():
    f = eng.text_fwd(pl.token_prefix, pl.token_suffix, ctx, pl.layout, pl.eot, save_for_bwd=True).clone()
    g = eng.text_bwd(dfeat).clone()
    return f, g
side = torch.cuda.Stream()
bad_f = bad_g = 0
with torch.no_grad():
    f0, g0 = text(); torch.cuda.synchronize()
    for it in range(ITERS):
        with torch.cuda.stream(side):
            eng.image_fwd(x)
        f, g = text(); torch.cuda.synchronize()
        if not torch.equal(f, f0):
            bad_f += 1
            if bad_f <= 4:
                d = (f - f0).abs(); print("   feature mismatch it", it, "rows", (d.max(1).values > 0).nonzero().flatten().tolist()[:8], "rel", float(d.max() / f0.abs().max()))
        if not torch.equal(g, g0):
            bad_g += 1
            if bad_g <= 4: print("   gradient mismatch it", it, "rel", float((g - g0).abs().max() / g0.abs().max()))
print(f"text tower under a concurrent image tower of {B} images ({prec}): feature mismatches {bad_f}/{ITERS}, gradient mismatches {bad_g}/{ITERS}", flush=True)
