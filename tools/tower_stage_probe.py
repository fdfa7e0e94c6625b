"""Which kernel of the packed image tower is not bit-stable under a concurrent text tower?  (manifestation (b))
The engine fingerprints every intermediate of the image tower (mvlpt_debug_checksums); the tower is repeated on the same input while the
text tower (forward + backward) runs on another stream; at every run whose fingerprints differ from the reference run the FIRST differing
stage is recorded.  Usage (GPU box): [MVLPT_RESID_PACKED=1] ITERS=6000 python tools/tower_stage_probe.py [B=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd.class_prompts import load_class_prompts
from mvlpt_amd.config import get_cfg_default
from mvlpt_amd.model import CustomCLIP, FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ITERS = int(os.environ.get("ITERS", "2000"))
arch = ARCHS["ViT-B/16"]
cfg = get_cfg_default(); cfg.TRAINER.MVLPT.COOP.N_CTX = 16
pre, C = load_class_prompts("caltech101", 16)
torch.manual_seed(0)
model = CustomCLIP(cfg, ["c"] * C, FrozenCLIP(make_state_dict(arch, 3), "fp16", precision="split_grad"), pretokenized=pre).cuda()
pl, eng = model.prompt_learner, model.engine
ctx = pl.ctx.detach()
x = torch.randn(B, 3, 224, 224, device="cuda").half()
tdfeat = torch.randn(C, arch.embed_dim, device="cuda") * 1e-3
side = torch.cuda.Stream()
# stage names in the order mvlpt_image_fwd records them (packed path, engine.hip)
names = ["patches", "patch embedding (conv GEMM)", "assemble: stream hi|lo", "assemble: ln_1 statistics"]
for l in range(arch.vision_layers - 1):
    names += [f"block {l}: qkv", f"block {l}: attention out", f"block {l}: stream behind out-projection", f"block {l}: ln_2 statistics",
              f"block {l}: MLP activations", f"block {l}: stream behind MLP down", f"block {l}: ln_1 statistics of block {l + 1}"]
names += ["last block: qkv", "last block: CLS attention rows", "last block: CLS stream rows (unpacked)"]
with torch.no_grad():
    eng.debug_checksums(True)
    f0 = eng.image_fwd(x, None, None, save_for_bwd=False).clone()
    ref = eng.debug_checksums(True)
    print(f"# {len(ref)} stages recorded ({len(names)} named); packed = {os.environ.get('MVLPT_RESID_PACKED', '0')}", flush=True)
    first_bad, feat_bad, n_bad = {}, 0, 0
    for it in range(ITERS):
        with torch.cuda.stream(side):
            eng.text_fwd(pl.token_prefix, pl.token_suffix, ctx, pl.layout, pl.eot, save_for_bwd=True)
            eng.text_bwd(tdfeat)
        f = eng.image_fwd(x, None, None, save_for_bwd=False)
        ck = eng.debug_checksums(True)
        if ck != ref:
            n_bad += 1
            i = next(k for k in range(min(len(ck), len(ref))) if ck[k] != ref[k])
            nm = names[i] if i < len(names) else f"stage {i}"
            first_bad[nm] = first_bad.get(nm, 0) + 1
            later = [k for k in range(i, len(ref)) if ck[k] != ref[k]]
            if n_bad <= 12:
                rows = ((f - f0).abs().max(1).values > 0).nonzero().flatten().tolist()
                print(f"   it {it}: first differing stage {i} = {nm}; {len(later)} stages differ from there on; feature rows {rows[:8]}", flush=True)
        if not torch.equal(f, f0): feat_bad += 1
    print(f"B={B} packed={os.environ.get('MVLPT_RESID_PACKED', '0')} lib={os.path.basename(os.environ.get('MVLPT_HIP_LIB', 'libmvlpt_hip.so'))}: "
          f"{n_bad}/{ITERS} towers with differing fingerprints, {feat_bad} with differing features; first differing stage: {first_bad}", flush=True)
