"""Logits / worst prompt-gradient error of three CoOp fixtures under the three precision modes (GPU box):
   python tools/precision_modes_report.py          (MVLPT_SPLIT_LO8=0: 16-bit pairs instead of the mixed pair in split_grad)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import load_npz, t, tiny_state_dict, case_grads, full_case_inputs
from tests.test_hip_model import build_model
from mvlpt_amd.model import FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict

def run(clip, case, image, pre, suf, res):
    model = build_model(case, clip, res, pre, suf)
    dev = clip.device
    label = t(case["label"])
    logits = model(image.to(dev), task=None)
    loss = model.cross_entropy(logits, label.to(dev))
    loss.backward()
    ref = t(case["out_logits"])
    le = float((logits.detach().cpu() - ref).abs().max()) / float(ref.abs().max())
    G = case_grads(case)
    ge = max(float((p.grad.cpu() - G[n]).abs().max()) / (float(G[n].abs().max()) + 1e-20) for n, p in model.prompt_learner.named_parameters())
    return le, ge

for mode in ["split_grad", "split_all", "fast"]:
    clip = FrozenCLIP(tiny_state_dict(), compute_dtype="fp16", precision=mode)
    for name in ["tiny_coop_end", "tiny_coop_middle"]:
        case = load_npz(name)
        le, ge = run(clip, case, t(case["image"]), t(case["token_prefix"]), t(case["token_suffix"]), 32)
        print(f"{mode:10s} lo8={os.environ.get('MVLPT_SPLIT_LO8','1')} {name:24s} logits {le:.2e} grads {ge:.2e}", flush=True)
    sd = make_state_dict(ARCHS["ViT-B/32"], 2, include_token_embedding=True)
    clip = FrozenCLIP(sd, compute_dtype="fp16", precision=mode)
    case = load_npz("full_vitb32_coop_end")
    image, pre, suf = full_case_inputs(case, sd, 224)
    le, ge = run(clip, case, image, pre, suf, 224)
    print(f"{mode:10s} lo8={os.environ.get('MVLPT_SPLIT_LO8','1')} {'full_vitb32_coop_end':24s} logits {le:.2e} grads {ge:.2e}", flush=True)
