"""Kernel-level stress of the packed residual update (mvlpt_op_gemm_residp) while another stream runs the text tower forward + backward:
every launch must reproduce the first one bit for bit.  GPU box only.
   python tools/residp_repro.py M K launches in_place(0/1) [fp32(0/1): the fp32-stream producer instead]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E
from mvlpt_amd.class_prompts import load_class_prompts
from mvlpt_amd.config import get_cfg_default
from mvlpt_amd.model import CustomCLIP, FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict
M, K, launches, in_place = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
fp32 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
N = 768
dev = "cuda"
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K, generator=g).half().to(dev)
W = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
b = (torch.randn(N, generator=g) * 0.1).to(dev)
resid = (torch.randn(M, N, generator=g) * 2).to(dev)
gamma = (1 + 0.2 * torch.randn(N, generator=g)).to(dev)
hi0, lo0, _ = E.op_respk_pack(resid)
arch = ARCHS["ViT-B/16"]; cfg = get_cfg_default(); cfg.TRAINER.MVLPT.COOP.N_CTX = 16
pre, C = load_class_prompts("caltech101", 16)
model = CustomCLIP(cfg, ["c"] * C, FrozenCLIP(make_state_dict(arch, 3), "fp16", precision="split_grad"), pretokenized=pre).cuda()
pl, eng = model.prompt_learner, model.engine; ctx = pl.ctx.detach()
tdfeat = torch.randn(C, arch.embed_dim, device=dev) * 1e-3
side = torch.cuda.Stream()
def run():
    if fp32:
        out32, x16, part, nt = E.op_gemm_ln_producer(A, W, b, resid, gamma)
        return out32, x16, part
    if in_place:
        h, l = hi0.clone(), lo0.clone()
        hi, lo, part, nt = E.op_gemm_residp(A, W, b, h, l, in_place=True)
    else:
        hi, lo, part, nt = E.op_gemm_residp(A, W, b, hi0, lo0)
    return hi, lo, part
with torch.no_grad():
    ref = [t.clone() for t in run()]
    bad = torch.zeros(3, device=dev, dtype=torch.int64)
    first = None
    done = 0
    while done < launches:
        with torch.cuda.stream(side):
            eng.text_fwd(pl.token_prefix, pl.token_suffix, ctx, pl.layout, pl.eot, save_for_bwd=True)
            eng.text_bwd(tdfeat)
        for _ in range(40):
            o = run()
            for k in range(3):
                neq = (o[k] != ref[k])
                bad[k] += neq.any().long()
            done += 1
        torch.cuda.synchronize()
    print(f"M={M} K={K} in_place={in_place} fp32={fp32}: launches {done}, mismatching [hi/out32, lo/x16, part] = {bad.tolist()}", flush=True)
