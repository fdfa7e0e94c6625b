"""LayerNorm folding on / off against the full-size reference fixtures (forced on at B = 4 with min_rows = 1), and the towers'
stand-alone timings in the three modes.  GPU box.  `python tools/fold_check.py [parity|time]`"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import FULL_CASES, case_grads, full_case_inputs, load_npz, t
from tests.test_hip_model import build_model
from mvlpt_amd.model import FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict

what = sys.argv[1] if len(sys.argv) > 1 else "parity"
if what == "parity":
    groups = {}
    for name in FULL_CASES:
        arch_name = "ViT-B/32" if "vitb32" in name else ("ViT-L/14@336px" if "vitl14_336" in name else "ViT-B/16")
        groups.setdefault(arch_name, []).append(name)
    for arch_name, cases in groups.items():
        sd = make_state_dict(ARCHS[arch_name], 2, include_token_embedding=True)
        clip = FrozenCLIP(sd, compute_dtype="fp16")
        for name in cases:
            case = load_npz(name)
            image, pre, suf = full_case_inputs(case, sd, ARCHS[arch_name].image_resolution)
            for mode in (0, 2):
                clip.engine.set_ln_fold(mode, 1)
                model = build_model(case, clip, ARCHS[arch_name].image_resolution, pre, suf)
                dev = clip.device
                logits = model(image.to(dev), task=None)
                loss = model.cross_entropy(logits, t(case["label"]).to(dev))
                loss.backward()
                ref = t(case["out_logits"])
                le = float((logits.detach().cpu() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
                G = case_grads(case)
                ge = max(float((p.grad.cpu() - G[n]).abs().max()) / (float(G[n].abs().max()) + 1e-20) for n, p in model.prompt_learner.named_parameters())
                with torch.no_grad():
                    pl = model.prompt_learner
                    coop, vpt, deep = pl.forward_mvlpt_proj(torch.float32)
                    img = model.engine.image_fwd(image.to(dev), vpt, deep).cpu()
                fi = float((img - t(case["out_image_features"])).abs().max()) / float(t(case["out_image_features"]).abs().max())
                print(f"{name:26s} fold={mode}: logits {le:.2e} worst grad {ge:.2e} inference image features {fi:.2e}", flush=True)
        del clip
else:
    from mvlpt_amd.model import build_prompt_layout
    arch = ARCHS["ViT-B/16"]
    eng = FrozenCLIP(make_state_dict(arch, 1)).engine
    x = torch.randn(256, 3, 224, 224, device="cuda").half()
    C, L, n = 100, 77, 16
    nl = [1 + (i % 3) for i in range(C)]
    layout = build_prompt_layout(nl, n, L, "middle").cuda()
    eot = torch.tensor([n + v + 2 for v in nl], dtype=torch.int32).cuda()
    pre = torch.randn(C, 1, 512, device="cuda") * 0.02; suf = torch.randn(C, L - 1 - n, 512, device="cuda") * 0.02
    ctx = torch.randn(n, 512, device="cuda") * 0.02; dfeat = torch.randn(C, 512, device="cuda") * 1e-3

    def timed(fn, iters=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3

    def text():
        eng.text_fwd(pre, suf, ctx, layout, eot, save_for_bwd=True); eng.text_bwd(dfeat)
    for rep in range(2):
        for mode in (0, 1, 2):
            eng.set_ln_fold(mode, 4096)
            f0 = eng.image_fwd(x).float().cpu()
            print(f"fold mode {mode}: image tower fwd {timed(lambda: eng.image_fwd(x)):.3f} ms   text fwd+bwd {timed(text):.3f} ms   "
                  f"text fwd {timed(lambda: eng.text_fwd(pre, suf, ctx, layout, eot, save_for_bwd=True)):.3f} ms", flush=True)
