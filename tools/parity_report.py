"""Print logits / prompt-gradient errors of the HIP path vs the golden vectors (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import TINY_CASES, load_npz, t, tiny_state_dict, case_grads
from tests.test_hip_model import build_model
from mvlpt_amd.model import FrozenCLIP

for dt in sys.argv[1:] or ["fp16", "bf16"]:
    clip = FrozenCLIP(tiny_state_dict(), compute_dtype=dt)
    for name in TINY_CASES:
        case = load_npz(name)
        model = build_model(case, clip, 32, t(case["token_prefix"]), t(case["token_suffix"]))
        dev = clip.device
        label = t(case["label"])
        if label.dtype != torch.int64:
            label = label.float(); label = label / label.sum(-1, keepdim=True)
        task = t(case["task"]) if "task" in case else None
        logits = model(t(case["image"]).to(dev), task=task)
        loss = model.cross_entropy(logits, label.to(dev))
        loss.backward()
        ref = t(case["out_logits"])
        le = float((logits.detach().cpu() - ref).abs().max()) / float(ref.abs().max())
        G = case_grads(case)
        ge = {}
        for n, p in model.prompt_learner.named_parameters():
            g = G[n]
            ge[n] = float((p.grad.cpu() - g).abs().max()) / (float(g.abs().max()) + 1e-20)
        top = sorted(ge.items(), key=lambda kv: -kv[1])[:3]
        print(f"{dt} {name:18s} logits {le:.2e} loss {abs(float(loss.detach())-float(case['out_loss'])):.2e} grads " +
              " ".join(f"{k.split('.')[-2] if '.' in k else k}.{k.split('.')[-1]}={v:.2e}" for k, v in top))

# ---- full-size output-only fixtures (weights / inputs regenerated from seeds, as in oracle/make_golden.py)
from tests.golden_util import FULL_CASES, full_case_inputs
from mvlpt_amd.weights import ARCHS, make_state_dict
for dt in sys.argv[1:] or ["fp16"]:
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
            model = build_model(case, clip, ARCHS[arch_name].image_resolution, pre, suf)
            dev = clip.device
            logits = model(image.to(dev), task=None)
            loss = model.cross_entropy(logits, t(case["label"]).to(dev))
            loss.backward()
            ref = t(case["out_logits"])
            le = float((logits.detach().cpu() - ref).abs().max()) / float(ref.abs().max())
            l2 = float((logits.detach().cpu() - ref).norm() / ref.norm())
            G = case_grads(case)
            ge = {}
            for n, p in model.prompt_learner.named_parameters():
                g = G[n]
                ge[n] = (float((p.grad.cpu() - g).abs().max()) / (float(g.abs().max()) + 1e-20), float((p.grad.cpu() - g).norm() / (g.norm() + 1e-30)))
            top = sorted(ge.items(), key=lambda kv: -kv[1][0])[:3]
            print(f"{dt} {name:24s} logits max {le:.2e} l2 {l2:.2e} loss {abs(float(loss.detach())-float(case['out_loss'])):.2e} grads " +
                  " ".join(f"{k[-24:]}={v[0]:.2e}/{v[1]:.2e}" for k, v in top))
        del clip
