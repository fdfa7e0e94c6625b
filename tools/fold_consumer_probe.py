"""Kernel-level stress probe with FORENSICS for the timing-dependent folded consumer of round 5.

The folded MLP-up consumer on mixed pairs at the 128x128 geometry (gemm_bt_kernel<f16, EPI_GELU_SPLIT_FOLD, 128, 128, 4, 2, MIXED>),
2 460 x 3072 x 768, is launched 6 x ITERS times while a PARTNER runs on another stream.  Every launch is compared bit for bit with
the first one; every differing element of the saved pre-activation u = rstd*acc + (-rstd*mean*colsum + bias2) is matched against the
values a list of candidate mechanisms would produce (stale accumulator row, stale row coefficients, the raw accumulator, another
row's result, ...), so that a mismatch names what the kernel read instead of what it should have read.

partner: none | text (the CoOp text tower, as in round 5) | vgpr64 | vgpr128 | vgpr256 | vgpr512 | sgpr | lds<bytes> | mem | valu | mfma |
         ldsrw   (tools/partners_gen.py: each dirties ONE resource)
Usage (GPU box): [MVLPT_HIP_LIB=...] python tools/fold_consumer_probe.py M N2 partner [iters=60] [epi=gelu] [u=1]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E

L_ = E._lib
M, N2, partner = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ITERS = int(sys.argv[4]) if len(sys.argv) > 4 else 60
epi_name = sys.argv[5] if len(sys.argv) > 5 else "gelu"
want_u = int(sys.argv[6]) if len(sys.argv) > 6 else 1
N1 = K1 = 768
dev = "cuda"
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K1, generator=g); W1 = (torch.randn(N1, K1, generator=g) * K1 ** -0.5).half().float()
b1 = torch.randn(N1, generator=g) * 0.1; resid = torch.randn(M, N1, generator=g) * 2
gamma = 1 + 0.2 * torch.randn(N1, generator=g); beta = 0.1 * torch.randn(N1, generator=g)
W2 = (torch.randn(N2, N1, generator=g) * N1 ** -0.5).half().float(); b2 = torch.randn(N2, generator=g) * 0.1
A2 = E.op_cast_mixed(A.to(dev), torch.float16)
W1p, e1 = E.op_pack_weight_mixed(W1.to(dev), torch.float16); W2p, e2 = E.op_pack_weight_mixed(W2.to(dev), torch.float16)
W2_16 = W2p[:, :N1].contiguous()
out32, x16, part, nt = E.op_gemm_ln_producer(A2, W1p, b1.to(dev), resid.to(dev), gamma.to(dev), a_split=2, x16_split=2, ldb=W1p.shape[1], w8_exp=e1)
cs, bias2 = E.op_fold_vectors(W2_16, N1, gamma.to(dev), beta.to(dev), b2.to(dev))
epi = {"store": L_.EPI_STORE_SPLIT, "gelu": L_.EPI_GELU_SPLIT}[epi_name]


def run():
    r = E.op_gemm_folded(x16, W2p, cs, bias2, part, nt, epi=epi, a_split=2, ldb=W2p.shape[1], w8_exp=e2, out2=bool(want_u) and epi_name == "gelu")
    return r if isinstance(r, tuple) else (r,)


# ------------------------------------------------------------------------------------------------ partners
side = torch.cuda.Stream()
P = None
if partner not in ("none", "text"):
    P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libpartners.so"))
sink = torch.zeros(64, device=dev)
membuf = torch.zeros(64 << 20, device=dev) if partner == "mem" else None      # 256 MB
text = None
if partner == "text":
    from mvlpt_amd.class_prompts import load_class_prompts
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.model import CustomCLIP, FrozenCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    arch = ARCHS["ViT-B/16"]; cfg = get_cfg_default(); cfg.TRAINER.MVLPT.COOP.N_CTX = 16
    pre, Cn = load_class_prompts("caltech101", 16)
    model = CustomCLIP(cfg, ["c"] * Cn, FrozenCLIP(make_state_dict(arch, 3), "fp16", precision="split_grad"), pretokenized=pre).cuda()
    pl, eng = model.prompt_learner, model.engine; ctx = pl.ctx.detach()
    text = (eng, pl, ctx)
SCALE = int(os.environ.get("PARTNER_SCALE", "1"))


def launch_partner():
    s = C.c_void_p(side.cuda_stream)
    if partner == "none":
        return
    if partner == "text":
        eng, pl, ctx = text
        with torch.cuda.stream(side):
            eng.text_fwd(pl.token_prefix, pl.token_suffix, ctx, pl.layout, pl.eot, save_for_bwd=False)
        return
    if partner.startswith("vgpr"):
        n = int(partner[4:])
        P.partner_vgpr(s, 60000 * SCALE, n, 8)
    elif partner == "sgpr":
        P.partner_sgpr(s, 60000 * SCALE, 8)
    elif partner.startswith("ldsrw"):
        P.partner_ldsrw(s, 8192 * SCALE, 8192, 400, C.c_void_p(sink.data_ptr()))
    elif partner.startswith("lds"):
        P.partner_lds(s, 20000 * SCALE, int(partner[3:]), 8)
    elif partner == "mem":
        P.partner_mem(s, 2048, C.c_void_p(membuf.data_ptr()), C.c_long(membuf.numel() // 4), 2 * SCALE)
    elif partner == "valu":
        P.partner_valu(s, 16384 * SCALE, 400, C.c_void_p(sink.data_ptr()))
    elif partner == "mfma":
        P.partner_mfma(s, 16384 * SCALE, 300, C.c_void_p(sink.data_ptr()))
    else:
        raise SystemExit(f"unknown partner {partner}")


# ------------------------------------------------------------------------------------------------ forensics
def coefficients():
    p = part.float().cpu()[:, :nt]
    s1 = p[..., 0].sum(1).double(); s2 = p[..., 1].sum(1).double()
    mean = s1 / K1; var = (s2 / K1 - mean * mean).clamp_min(0)
    fa = (var + 1e-5).rsqrt(); fcc = -fa * mean
    return fa, fcc


def forensics(bad_u, ref_u, limit=24):
    """bad_u / ref_u: fp16 [M, N2] pre-activations of a differing launch and of the reference launch.  Per bad (row, element position):
    least-squares fit of the error over the row's bad columns against colsum[n] (a wrong -rstd*mean), the accumulator (a wrong rstd)
    and a constant; then the implied wrong coefficient is compared with the coefficients of the tile's other rows and with 0."""
    fa, fcc = coefficients()
    col, colb = cs.double().cpu(), bias2.double().cpu()
    ref = ref_u.double().cpu(); bad = bad_u.double().cpu()
    diff = bad != ref
    idx = diff.nonzero()
    lines = []
    groups = {}
    for m, n in idx.tolist():
        groups.setdefault((m, n % 4, n // 64), []).append(n)
    summary = {}
    for (m, e, blk), ns in list(groups.items())[:limit]:
        ns = torch.tensor(ns)
        err = (bad[m, ns] - ref[m, ns])
        t_m = fcc[m] * col[ns] + colb[ns]
        acc = (ref[m, ns] - t_m) / fa[m]
        res = {}
        for name, x in (("colsum (wrong -rstd*mean)", col[ns]), ("accumulator (wrong rstd)", acc), ("constant", torch.ones_like(err))):
            k = float((x * err).sum() / (x * x).sum())
            r = err - k * x
            res[name] = (k, float(r.abs().max()), float(err.abs().max()))
        best = min(res, key=lambda q: res[q][1])
        k, rmax, emax = res[best]
        tag = best if rmax < 0.15 * emax + 1.5e-3 else "no single-coefficient fit"
        ident = ""
        if tag.startswith("colsum"):
            wrong = float(fcc[m]) + k
            tile0 = (m // 128) * 128
            cand = {f"-rstd*mean of row {r - m:+d}": float(fcc[r]) for r in range(tile0, min(tile0 + 128, M)) if r != m}
            cand.update({f"rstd of row {r - m:+d}": float(fa[r]) for r in range(tile0, min(tile0 + 128, M))})
            cand["zero"] = 0.0
            b2 = min(cand, key=lambda q: abs(cand[q] - wrong))
            prev = {dm: (float(fcc[m + dm]) if 0 <= m + dm < M else float("nan")) for dm in (-4, -8, -12, -16, 4)}
            ident = (f"; implied -rstd*mean {wrong:+.5f} (right {float(fcc[m]):+.5f}); nearest: {b2} = {cand[b2]:+.5f}; rows -4/-8/-12/-16/+4 have "
                     + "/".join(f"{prev[d]:+.5f}" for d in (-4, -8, -12, -16, 4)))
            tag2 = b2 if abs(cand[b2] - wrong) < 4e-4 else "colsum fit, value unidentified"
            summary[tag2] = summary.get(tag2, 0) + 1
        elif tag.startswith("accumulator"):
            wrong = float(fa[m]) + k
            tile0 = (m // 128) * 128
            cand = {f"rstd of row {r - m:+d}": float(fa[r]) for r in range(tile0, min(tile0 + 128, M)) if r != m}
            cand.update({f"-rstd*mean of row {r - m:+d}": float(fcc[r]) for r in range(tile0, min(tile0 + 128, M))})
            cand["zero"] = 0.0
            b2 = min(cand, key=lambda q: abs(cand[q] - wrong))
            ident = f"; implied rstd {wrong:+.5f} (right {float(fa[m]):+.5f}); nearest: {b2} = {cand[b2]:+.5f}"
            tag2 = "rstd <- " + b2 if abs(cand[b2] - wrong) < 4e-4 else "accumulator fit, value unidentified"
            summary[tag2] = summary.get(tag2, 0) + 1
        else:
            summary[tag] = summary.get(tag, 0) + 1
        if len(lines) < 10:
            lines.append(f"      row {m} (row%16 = {m % 16}, tile row {m % 128}) element {e} of columns {int(ns.min())}..{int(ns.max())} ({len(ns)} bad): max|err| {emax:.4f}; "
                         f"fit {tag}: k = {k:+.5f}, residual {rmax:.1e}{ident}")
    return int(diff.sum()), summary, lines


with torch.no_grad():
    ref = [t.clone() for t in run()]; torch.cuda.synchronize()
    # is the partner really concurrent?  its duration alone and the six launches' duration alone
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(side); launch_partner(); ev[1].record(side)
    torch.cuda.synchronize()
    ev[2].record()
    for _ in range(6): run()
    ev[3].record(); torch.cuda.synchronize()
    print(f"# partner {partner}: alone {ev[0].elapsed_time(ev[1]) * 1e3:.0f} us; six launches alone {ev[2].elapsed_time(ev[3]) * 1e3:.0f} us", flush=True)
    bad = 0; nan_launches = 0; reports = []
    for it in range(ITERS):
        launch_partner()
        outs = [[t.clone() for t in run()] for _ in range(6)]
        torch.cuda.synchronize()
        for o in outs:
            if not all(torch.equal(a, b) for a, b in zip(o, ref)):
                bad += 1
                if any(bool(torch.isnan(t.float()).any()) for t in o): nan_launches += 1
                if len(reports) < 6 and len(o) > 1: reports.append(forensics(o[1], ref[1]))
    lib = os.path.basename(os.environ.get("MVLPT_HIP_LIB", "libmvlpt_hip.so"))
    nref = sum(int(torch.isnan(t.float()).sum()) for t in ref)
    print(f"{lib} M={M} N={N2} epi={epi_name} partner={partner}: mismatching launches {bad}/{6 * ITERS} (with NaN: {nan_launches}); NaN in the reference launch: {nref}", flush=True)
    for n_el, tally, worst in reports:
        print(f"   differing u elements {n_el}: " + "; ".join(f"{k} x{v}" for k, v in sorted(tally.items(), key=lambda kv: -kv[1])))
        for w in worst[:6]: print(w)
