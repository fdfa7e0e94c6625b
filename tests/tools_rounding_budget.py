"""Error-budget study (test tooling, CPU): which 16-bit roundings of the HIP data flow dominate the prompt-gradient
error?  Mirrors the text tower's HIP forward/backward on top of the oracle's primitives and rounds the listed sites
to fp16 (or to a hi+lo pair = ~22 bits), one class at a time.

    python tests/tools_rounding_budget.py [C] [L]
"""
import math
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import clip_oracle as O            # noqa: E402
from mvlpt_amd.weights import ARCHS, make_state_dict   # noqa: E402

SITES = ["h1", "qkv", "p", "o", "h2", "a", "u", "dx_pr", "du", "dx_o", "dO", "pb", "dS", "dqkv"]


LO_EXP = 10          # lo8 = e5m2(lo * 2^LO_EXP): |lo| <= 2^-11 |x| < 2^5 for every finite fp16 x, never saturates


def q_e5m2(t, exp):
    s = 2.0 ** exp
    return (t * s).clamp(-57344.0, 57344.0).to(torch.float8_e5m2).float() / s


def q_e4m3(t, exp=None):
    if exp is None:       # per-tensor power of two that puts max|t| just below 448
        exp = math.floor(math.log2(448.0 / float(t.abs().max())))
    s = 2.0 ** exp
    return (t * s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() / s


def make_round(enabled, split, mix=(), lo_fmt="e5m2"):
    """r(t, site): plain rounding of a site; r.mm(t, site, Wt): the GEMM  round(t) @ Wt  of a GEMM-A site, where a site in
    `mix` runs the mixed recipe  hi @ Wt + lo8 @ W8t  (lo in fp8 with a fixed exponent, the weight's fp8 copy per tensor)."""
    def r(t, site):
        if site not in enabled:
            return t
        hi = t.half().float()
        if site in split or site in mix:
            return hi + (t - hi).half().float()
        return hi

    def mm(t, site, wt):
        if site in enabled and site in mix:
            hi = t.half().float()
            lo = t - hi
            lo8 = q_e5m2(lo, LO_EXP) if lo_fmt == "e5m2" else q_e4m3(lo, LO_EXP + 2)
            return hi @ wt + lo8 @ q_e4m3(wt)
        return r(t, site) @ wt
    r.mm = mm
    return r


def block_fwd(x, sd, pre, heads, r):
    d = x.shape[-1]
    h1, ln1 = O.layernorm_fwd(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"])
    qkv = r(r.mm(h1, "h1", sd[pre + "attn.in_proj_weight"].t()) + sd[pre + "attn.in_proj_bias"], "qkv")
    q, k, v = (O._split_heads(t, heads) for t in qkv.split(d, dim=-1))
    L = q.shape[-2]
    s = torch.matmul(q, k.transpose(-1, -2)) / 8.0 + torch.full((L, L), float("-inf")).triu_(1)
    m = s.max(-1, keepdim=True).values
    e = torch.exp(s - m)
    den = e.sum(-1, keepdim=True)
    lse = m + den.log()
    p = e / den
    o_raw = torch.matmul(r(p, "p"), v)
    o = r(o_raw, "o")
    xm = x + r.mm(O._merge_heads(o_raw), "o", sd[pre + "attn.out_proj.weight"].t()) + sd[pre + "attn.out_proj.bias"]
    h2, ln2 = O.layernorm_fwd(xm, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"])
    u = r.mm(h2, "h2", sd[pre + "mlp.c_fc.weight"].t()) + sd[pre + "mlp.c_fc.bias"]
    xo = xm + r.mm(O.quick_gelu(u), "a", sd[pre + "mlp.c_proj.weight"].t()) + sd[pre + "mlp.c_proj.bias"]
    return xo, (ln1, q, k, v, lse, o, ln2, r(u, "u"))


def block_bwd(dx, saved, sd, pre, r):
    ln1, q, k, v, lse, o, ln2, u = saved
    da = r.mm(dx, "dx_pr", sd[pre + "mlp.c_proj.weight"])
    dh2 = r.mm(da * O.quick_gelu_grad(u), "du", sd[pre + "mlp.c_fc.weight"])
    dxm = dx + O.layernorm_bwd(dh2, ln2[0], ln2[1], sd[pre + "ln_2.weight"])
    do = O._split_heads(r(r.mm(dxm, "dx_o", sd[pre + "attn.out_proj.weight"]), "dO"), q.shape[1])
    L = q.shape[-2]
    s = torch.matmul(q, k.transpose(-1, -2)) / 8.0 + torch.full((L, L), float("-inf")).triu_(1)
    p = torch.exp(s - lse)
    dv = torch.matmul(r(p, "pb").transpose(-1, -2), do)
    dp = torch.matmul(do, v.transpose(-1, -2))
    delta = (do * o).sum(-1, keepdim=True)
    ds = r(p * (dp - delta) / 8.0, "dS")
    dq = torch.matmul(ds, k)
    dk = torch.matmul(ds.transpose(-1, -2), q)
    dh1 = r.mm(torch.cat([O._merge_heads(dq), O._merge_heads(dk), O._merge_heads(dv)], dim=-1), "dqkv", sd[pre + "attn.in_proj_weight"])
    return dxm + O.layernorm_bwd(dh1, ln1[0], ln1[1], sd[pre + "ln_1.weight"])


def run(sd, prompts, eot, dfeat_fn, heads, layers, enabled, split=(), mix=(), lo_fmt="e5m2"):
    r = make_round(set(enabled), set(split), set(mix), lo_fmt)
    C, L, dt = prompts.shape
    x = prompts + sd["positional_embedding"][:L]
    saved = []
    for l in range(layers):
        x, s = block_fwd(x, sd, f"transformer.resblocks.{l}.", heads, r)
        saved.append(s)
    rows = x[torch.arange(C), eot]
    y, lnf = O.layernorm_fwd(rows, sd["ln_final.weight"], sd["ln_final.bias"])
    feat = y @ sd["text_projection"]
    dfeat = dfeat_fn(feat)
    drows = O.layernorm_bwd(dfeat @ sd["text_projection"].t(), lnf[0], lnf[1], sd["ln_final.weight"])
    scale = 2.0 ** math.floor(math.log2(64.0 / float(drows.abs().max())))
    dx = torch.zeros_like(x)
    dx[torch.arange(C), eot] = drows * scale
    for l in reversed(range(layers)):
        dx = block_bwd(dx, saved[l], sd, f"transformer.resblocks.{l}.", r)
    return feat, dx / scale


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    torch.manual_seed(0)
    arch = ARCHS["ViT-B/16"]
    sd = make_state_dict(arch, seed=1)
    dt, n_ctx = arch.transformer_width, 16
    name_lens = [1 + (i % 3) for i in range(C)]
    layout = O.build_prompt_layout(name_lens, n_ctx, L, "middle")
    eot = torch.tensor([n_ctx + nl + 2 for nl in name_lens])
    prefix, suffix = torch.randn(C, 1, dt) * 0.02, torch.randn(C, L - 1 - n_ctx, dt) * 0.02
    ctx = torch.randn(n_ctx, dt) * 0.02
    prompts = O.assemble_prompts(ctx, prefix, suffix, layout)
    B = 64
    img = torch.randn(B, arch.embed_dim)
    label = torch.randint(0, C, (B,))
    scale = float(sd["logit_scale"].exp())

    def dfeat_fn(feat):
        logits, lctx = O.logits_fwd(img, feat, scale)
        _, dl = O.cross_entropy_fwd_bwd(logits, label)
        return O.logits_bwd(dl, lctx)[1]

    heads, layers = arch.transformer_heads, arch.transformer_layers
    with torch.no_grad():
        f0, dx0 = run(sd, prompts, eot, dfeat_fn, heads, layers, [])
        g0 = O.scatter_prompt_grad(dx0, layout, (n_ctx, dt))

        def report(tag, enabled, split=(), mix=(), lo_fmt="e5m2"):
            f, dx = run(sd, prompts, eot, dfeat_fn, heads, layers, enabled, split, mix, lo_fmt)
            g = O.scatter_prompt_grad(dx, layout, (n_ctx, dt))
            eg = float((g - g0).abs().max() / g0.abs().max())
            el2 = float((g - g0).norm() / g0.norm())
            ef = float((f - f0).abs().max() / f0.abs().max())
            print(f"{tag:34s} grad max-rel {eg:.2e}  L2-rel {el2:.2e}   feat {ef:.2e}")
            return eg

        GEMM_A = ["h1", "o", "h2", "a", "dx_pr", "du", "dx_o", "dqkv"]
        report("all sites fp16", SITES)
        report("all, split all", SITES, SITES)
        report("split all, GEMM-A lo in e5m2 x W e4m3", SITES, SITES, GEMM_A)
        report("split all, GEMM-A lo in e4m3 x W e4m3", SITES, SITES, GEMM_A, "e4m3")
        report("mixed fwd GEMMs only", SITES, SITES, GEMM_A[:4])
        report("mixed bwd GEMMs only", SITES, SITES, GEMM_A[4:])
        if len(sys.argv) > 3 and sys.argv[3] == "mix":
            return
        for s in SITES:
            report(f"only {s}", [s])
        report("forward sites only", SITES[:7])
        report("backward sites only", SITES[7:])
        report("all, split all", SITES, SITES)
        for s in SITES:
            report(f"all fp16 but split {s}", SITES, [s])
        report("split h1,h2,a,qkv,o", SITES, ["h1", "h2", "a", "qkv", "o"])
        report("split all fwd", SITES, SITES[:7])
        report("split all bwd", SITES, SITES[7:])
        report("split GEMM A operands (h1,o,h2,a,dx_pr,du,dx_o,dqkv)", SITES, ["h1", "o", "h2", "a", "dx_pr", "du", "dx_o", "dqkv"])


if __name__ == "__main__":
    main()
