"""Image / text feature errors of the HIP towers vs the full-size reference fixtures (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import FULL_CASES, full_case_inputs, load_npz, t
from tests.test_hip_model import build_model
from mvlpt_amd.model import FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict
for arch_name, cases in (("ViT-B/32", ["full_vitb32_coop_end"]), ("ViT-B/16", ["full_vitb16_coop_middle", "full_vitb16_vpt_deep", "full_vitb16_upt_cut"])):
    sd = make_state_dict(ARCHS[arch_name], 2, include_token_embedding=True)
    clip = FrozenCLIP(sd, compute_dtype="fp16")
    for name in cases:
        case = load_npz(name)
        image, pre, suf = full_case_inputs(case, sd, ARCHS[arch_name].image_resolution)
        model = build_model(case, clip, ARCHS[arch_name].image_resolution, pre, suf)
        pl = model.prompt_learner
        with torch.no_grad():
            coop, vpt, deep = pl.forward_mvlpt_proj(torch.float32)
            img = model.engine.image_fwd(image.cuda(), vpt, deep)
            txt = model.engine.text_fwd(pl.token_prefix, pl.token_suffix, coop, pl.layout, pl.eot)
        for got, key in ((img, "out_image_features"), (txt, "out_text_features")):
            ref = t(case[key])
            print(f"{name:26s} {key:20s} max-rel {float((got.cpu() - ref).abs().max() / ref.abs().max()):.2e}")
