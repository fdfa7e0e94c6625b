"""Kernel-level reproducer of a timing-dependent result (round 5): the folded MLP-up consumer on mixed pairs at the 128x128 geometry
(gemm_bt_kernel<f16, EPI_GELU_SPLIT_FOLD, 128, 128, 4 waves, 2-deep ring, MIXED>), 2 460 x 3072 x 768, repeated while another stream runs the
text tower: with the fold arithmetic compiled to v_pk_fma_f32 + op_sel ~40 % of the launches differed from the first one (lanes 48-63 of a row
segment, the LOW element of a packed pair); on scalar fmas (gemm_epi.h fold_apply) 0 of 360.  Kept as the stress test for that class of bug.
Usage (GPU box): [NOFOLD=1] python tools/fold_consumer_repro.py M N2 mixed(0/1) gelu|store save_u(0/1) concurrent(0/1)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E
from mvlpt_amd.class_prompts import load_class_prompts
from mvlpt_amd.config import get_cfg_default
from mvlpt_amd.model import CustomCLIP, FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict
L_ = E._lib
M, N2, mixed, epi_name, want_u, conc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
N1 = K1 = 768
dev = "cuda"
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K1, generator=g); W1 = (torch.randn(N1, K1, generator=g) * K1 ** -0.5).half().float()
b1 = torch.randn(N1, generator=g) * 0.1; resid = torch.randn(M, N1, generator=g) * 2
gamma = 1 + 0.2 * torch.randn(N1, generator=g); beta = 0.1 * torch.randn(N1, generator=g)
W2 = (torch.randn(N2, N1, generator=g) * N1 ** -0.5).half().float(); b2 = torch.randn(N2, generator=g) * 0.1
if mixed:
    A2 = E.op_cast_mixed(A.to(dev), torch.float16)
    W1p, e1 = E.op_pack_weight_mixed(W1.to(dev), torch.float16); W2p, e2 = E.op_pack_weight_mixed(W2.to(dev), torch.float16)
    ldb1, ldb2, sp = W1p.shape[1], W2p.shape[1], 2
    W2_16 = W2p[:, :N1].contiguous()
else:
    A2 = A.half().to(dev); W1p = W1.half().to(dev); W2p = W2.half().to(dev); ldb1 = ldb2 = 0; e1 = e2 = 0; sp = 0; W2_16 = W2p
out32, x16, part, nt = E.op_gemm_ln_producer(A2, W1p, b1.to(dev), resid.to(dev), gamma.to(dev), a_split=sp, x16_split=sp, ldb=ldb1, w8_exp=e1)
cs, bias2 = E.op_fold_vectors(W2_16, N1, gamma.to(dev), beta.to(dev), b2.to(dev))
epi = {"store": L_.EPI_STORE_SPLIT if mixed else L_.EPI_STORE16, "gelu": L_.EPI_GELU_SPLIT if mixed else L_.EPI_GELU}[epi_name]
nofold = int(os.environ.get("NOFOLD", "0"))
h16m = None
if nofold:
    xn = torch.nn.functional.layer_norm(out32.float().cpu(), (N1,), gamma, beta, 1e-5)
    h16m = E.op_cast_mixed(xn.to(dev), torch.float16) if mixed else xn.half().to(dev)
def run():
    if nofold:
        r = E.op_gemm_mixed(h16m, W2p, e2, epi=epi, bias=b2.to(dev), out2=bool(want_u) and epi_name == "gelu")
        return r if isinstance(r, tuple) else (r,)
    r = E.op_gemm_folded(x16, W2p, cs, bias2, part, nt, epi=epi, a_split=sp, ldb=ldb2, w8_exp=e2, out2=bool(want_u) and epi_name == "gelu")
    return r if isinstance(r, tuple) else (r,)
# concurrent text tower
arch = ARCHS["ViT-B/16"]; cfg = get_cfg_default(); cfg.TRAINER.MVLPT.COOP.N_CTX = 16
pre, C = load_class_prompts("caltech101", 16)
model = CustomCLIP(cfg, ["c"] * C, FrozenCLIP(make_state_dict(arch, 3), "fp16", precision="split_grad"), pretokenized=pre).cuda()
pl, eng = model.prompt_learner, model.engine; ctx = pl.ctx.detach()
side = torch.cuda.Stream()
with torch.no_grad():
    ref = [t.clone() for t in run()]; torch.cuda.synchronize()
    bad = 0; first = None
    for it in range(60):
        if conc:
            with torch.cuda.stream(side):
                eng.text_fwd(pl.token_prefix, pl.token_suffix, ctx, pl.layout, pl.eot, save_for_bwd=False)
        outs = []
        for _ in range(6): outs.append([t.clone() for t in run()])
        torch.cuda.synchronize()
        for o in outs:
            if not all(torch.equal(a, b) for a, b in zip(o, ref)):
                bad += 1
                if first is None:
                    d = (o[0].float() - ref[0].float()).abs()
                    rows = (d.max(1).values > 0).nonzero().flatten()
                    cols = (d.max(0).values > 0).nonzero().flatten()
                    first = f"rows {rows[:8].tolist()} (n={len(rows)}) cols {cols[:8].tolist()} (n={len(cols)}) max {float(d.max()):.3e}"
print(f"M={M} N={N2} mixed={mixed} epi={epi_name} u={want_u} concurrent={conc}: mismatching launches {bad}/360  {first}", flush=True)
