"""Inference-mode (model.eval() + torch.no_grad(): MVLPT.model_inference, trainers/mvlpt.py:986-987) logits and tower features of
the HIP path vs the reference goldens, every fixture.  GPU box.  `python tools/inference_parity.py [fp16]`"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import FULL_CASES, TINY_CASES, full_case_inputs, load_npz, t, tiny_state_dict
from tests.test_hip_model import build_model
from mvlpt_amd.model import FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict


def report(name, case, model, image):
    dev = model.clip_model.device
    model.eval()
    task = t(case["task"]) if "task" in case else None
    with torch.no_grad():
        logits = model(image.to(dev), task=task).cpu()
        pl = model.prompt_learner
        coop, vpt, deep = pl.forward_mvlpt_proj(torch.float32)
        img = model.engine.image_fwd(image.to(dev), vpt, deep).cpu()
        txt = model.engine.text_fwd(pl.token_prefix, pl.token_suffix, coop, pl.layout, pl.eot).cpu()
    ref = t(case["out_logits"])
    le = float((logits - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    fi = float((img - t(case["out_image_features"])).abs().max()) / float(t(case["out_image_features"]).abs().max())
    ft = float((txt - t(case["out_text_features"])).abs().max()) / float(t(case["out_text_features"]).abs().max())
    print(f"{name:26s} inference logits {le:.2e}   image features {fi:.2e}   text features {ft:.2e}", flush=True)


dt = sys.argv[1] if len(sys.argv) > 1 else "fp16"
clip = FrozenCLIP(tiny_state_dict(), compute_dtype=dt)
for name in TINY_CASES:
    case = load_npz(name)
    report(name, case, build_model(case, clip, 32, t(case["token_prefix"]), t(case["token_suffix"])), t(case["image"]))
groups = {}
for name in FULL_CASES:
    arch_name = "ViT-B/32" if "vitb32" in name else ("ViT-L/14@336px" if "vitl14_336" in name else "ViT-B/16")
    groups.setdefault(arch_name, []).append(name)
for arch_name, cases in groups.items():
    sd = make_state_dict(ARCHS[arch_name], 2, include_token_embedding=True)
    clip = FrozenCLIP(sd, compute_dtype=dt)
    for name in cases:
        case = load_npz(name)
        image, pre, suf = full_case_inputs(case, sd, ARCHS[arch_name].image_resolution)
        report(name, case, build_model(case, clip, ARCHS[arch_name].image_resolution, pre, suf), image)
    del clip
