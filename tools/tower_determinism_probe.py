"""Bit-stability of the split image tower (forward + backward) while another stream runs the text tower (or unrelated traffic).
Usage (GPU box): [MVLPT_LN_FOLD_MIN_ROWS=1] python tools/tower_determinism_probe.py B deep(-1 no prompts, 0 shallow, 1 deep) concurrent(0 none, 1 text forward, 2 matmuls, 3 text forward + backward) [save(0/1)]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd.class_prompts import load_class_prompts
from mvlpt_amd.config import get_cfg_default
from mvlpt_amd.model import CustomCLIP, FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict
B = int(sys.argv[1]); deep_on = int(sys.argv[2]); conc = int(sys.argv[3]); save = int(sys.argv[4]) if len(sys.argv) > 4 else 1
arch = ARCHS["ViT-B/16"]
cfg = get_cfg_default(); cfg.TRAINER.MVLPT.COOP.N_CTX = 16
pre, C = load_class_prompts("caltech101", 16)
torch.manual_seed(0)
model = CustomCLIP(cfg, ["c"] * C, FrozenCLIP(make_state_dict(arch, 3), "fp16", precision="split_grad"), pretokenized=pre).cuda()
pl, eng = model.prompt_learner, model.engine
ctx = pl.ctx.detach()
n, dv = 8, arch.vision_width
vpt = torch.randn(n, dv, device="cuda") * 0.05 if deep_on >= 0 else None      # deep = -1: no prompts at all (CoOp image tower)
deep = torch.randn(arch.vision_layers - 1, n, dv, device="cuda") * 0.05 if deep_on > 0 else None
x = torch.randn(B, 3, 224, 224, device="cuda").half()
dfeat = torch.randn(B, arch.embed_dim, device="cuda") * 1e-3
def image():
    f = eng.image_fwd(x, vpt, deep, save_for_bwd=bool(save) and vpt is not None).clone()
    if save and vpt is not None:
        a, b = eng.image_bwd(dfeat)
        return f, a.clone(), (b.clone() if b is not None else a.clone())
    return f, f, f
side = torch.cuda.Stream()
ITERS = int(os.environ.get('ITERS', '40'))
tdfeat = torch.randn(C, arch.embed_dim, device='cuda') * 1e-3
ma = torch.randn(3000, 512, device='cuda').half(); mb = torch.randn(512, 2048, device='cuda').half(); mc = torch.empty(3000, 2048, device='cuda').half()
bad_f = bad_g = 0
with torch.no_grad():
    f0, a0, b0 = image(); torch.cuda.synchronize()
    for it in range(ITERS):
        if conc == 1:
            with torch.cuda.stream(side):
                eng.text_fwd(pl.token_prefix, pl.token_suffix, ctx, pl.layout, pl.eot, save_for_bwd=False)
        elif conc == 3:      # text tower forward + backward, as in the CoOp step
            with torch.cuda.stream(side):
                eng.text_fwd(pl.token_prefix, pl.token_suffix, ctx, pl.layout, pl.eot, save_for_bwd=True)
                eng.text_bwd(tdfeat)
        elif conc == 2:      # unrelated traffic instead of the text tower
            with torch.cuda.stream(side):
                for _ in range(30): torch.mm(ma, mb, out=mc)
        f, a, b = image(); torch.cuda.synchronize()
        if not torch.equal(f, f0):
            bad_f += 1
            if bad_f <= 6:
                d = (f - f0).abs(); rows = (d.max(1).values > 0).nonzero().flatten().tolist()
                print("   first mismatch it", it, "rows", rows[:10], "max diff", float(d.max()), "rel", float(d.max() / f0.abs().max()))
        if not (torch.equal(a, a0) and torch.equal(b, b0)): bad_g += 1
print(f"B={B} deep={deep_on} concurrent={conc} save={save} env={ {k: v for k, v in os.environ.items() if k.startswith('MVLPT_')} }: feature mismatches {bad_f}/{ITERS}, gradient mismatches {bad_g}/{ITERS}", flush=True)
