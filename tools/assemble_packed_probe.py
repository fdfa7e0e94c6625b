"""Kernel-level probe of the packed tower entry (assemble_tokens_kernel<.., PK>): the stage the tower fingerprints name as the first one
that differs under a concurrent text tower (tools/tower_stage_probe.py, round 6).  The kernel is launched 6 x ITERS times on the same
input while a partner runs on another stream; differing launches are compared element by element with the first launch AND with a
float64 host recomputation of ln_pre, so that the report says which of the two is the wrong one.
Usage: [MVLPT_HIP_LIB=..] python tools/assemble_packed_probe.py partner(none|text|mfma|valu|mem) [iters=200]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E
partner = sys.argv[1]; ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B, G2, d = 256, 196, 768
g_ = torch.Generator().manual_seed(3)
pe = (torch.randn(B * G2, d, generator=g_) * 0.5).cuda()
cls = torch.randn(d, generator=g_).cuda() * 0.3; pos = (torch.randn(1 + G2, d, generator=g_) * 0.1).cuda()
lg = (1 + 0.2 * torch.randn(d, generator=g_)).cuda(); lb = (0.1 * torch.randn(d, generator=g_)).cuda()
side = torch.cuda.Stream(); sink = torch.zeros(64, device="cuda")
P = None; text = None
if partner in ("mfma", "valu", "mem"):
    P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libpartners.so"))
    membuf = torch.zeros(64 << 20, device="cuda")
if partner == "text":
    from mvlpt_amd.class_prompts import load_class_prompts
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.model import CustomCLIP, FrozenCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    arch = ARCHS["ViT-B/16"]; cfg = get_cfg_default(); cfg.TRAINER.MVLPT.COOP.N_CTX = 16
    pre, Cn = load_class_prompts("caltech101", 16)
    model = CustomCLIP(cfg, ["c"] * Cn, FrozenCLIP(make_state_dict(arch, 3), "fp16", precision="split_grad"), pretokenized=pre).cuda()
    pl, eng = model.prompt_learner, model.engine; ctx = pl.ctx.detach(); tdf = torch.randn(Cn, arch.embed_dim, device="cuda") * 1e-3
def launch_partner():
    s = C.c_void_p(side.cuda_stream)
    if partner == "text":
        with torch.cuda.stream(side):
            eng.text_fwd(pl.token_prefix, pl.token_suffix, ctx, pl.layout, pl.eot, save_for_bwd=True); eng.text_bwd(tdf)
    elif partner == "mfma": P.partner_mfma(s, 16384 * 4, 300, C.c_void_p(sink.data_ptr()))
    elif partner == "valu": P.partner_valu(s, 16384 * 4, 400, C.c_void_p(sink.data_ptr()))
    elif partner == "mem": P.partner_mem(s, 2048, C.c_void_p(membuf.data_ptr()), C.c_long(membuf.numel() // 4), 4)
run = lambda: E.op_assemble_packed(pe, cls, pos, lg, lb, B)
with torch.no_grad():
    ref = [t.clone() for t in run()]; torch.cuda.synchronize()
    # float64 host recomputation of the rows (ln_pre of cls / patch embedding + positional embedding)
    def host_rows(rows):
        out = []
        for r in rows:
            b, i = divmod(r, 1 + G2)
            v = (cls if i == 0 else pe[b * G2 + i - 1]).double().cpu() + pos[i].double().cpu()
            m = v.mean(); var = ((v - m) ** 2).mean()
            out.append(((v - m) / torch.sqrt(var + 1e-5)) * lg.double().cpu() + lb.double().cpu())
        return torch.stack(out)
    def unpack(hi, lo):
        h = hi.float()
        return (h.view(torch.int32) + (lo.to(torch.int32) << 5)).view(torch.float32)
    bad = 0; shown = 0
    for it in range(ITERS):
        launch_partner()
        outs = [[t.clone() for t in run()] for _ in range(6)]
        torch.cuda.synchronize()
        for o in outs:
            neq = [not torch.equal(a, b) for a, b in zip(o, ref)]
            if any(neq):
                bad += 1
                if shown < 8:
                    shown += 1
                    dh = (o[0] != ref[0]) | (o[1] != ref[1])
                    rows = dh.any(1).nonzero().flatten().tolist()
                    dp = (o[2] != ref[2]).any(-1).any(-1).nonzero().flatten().tolist()
                    msg = f"   launch differs in [hi, lo, part] = {neq}: stream rows {rows[:6]} ({len(rows)} rows, {int(dh.sum())} elements), part rows {dp[:6]} ({len(dp)})"
                    if rows:
                        r = rows[0]; cols = dh[r].nonzero().flatten()
                        want = host_rows([r])[0]
                        xb = unpack(o[0][r].cpu(), o[1][r].cpu()).double(); xr = unpack(ref[0][r].cpu(), ref[1][r].cpu()).double()
                        eb = float((xb - want).abs().max()); er = float((xr - want).abs().max())
                        msg += (f"\n      row {r} (image {r // (1 + G2)}, token {r % (1 + G2)}): {len(cols)} columns differ, first {cols[:8].tolist()}; "
                                f"max |x - host fp64|: this launch {eb:.3e}, first launch {er:.3e}; "
                                f"part: this {o[2][r, 0].tolist()} first {ref[2][r, 0].tolist()} host {[float(want.sum()), float((want * want).sum())]}")
                    print(msg, flush=True)
    print(f"{os.path.basename(os.environ.get('MVLPT_HIP_LIB', 'libmvlpt_hip.so'))} partner={partner}: differing launches {bad}/{6 * ITERS}", flush=True)
