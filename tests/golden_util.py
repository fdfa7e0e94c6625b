"""Helpers shared by the oracle (CPU) and HIP (GPU) parity tests: load a golden case."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

TINY_CASES = ["tiny_coop_end", "tiny_coop_middle", "tiny_coop_front", "tiny_coop_csc", "tiny_coop_cut",
              "tiny_vpt_shallow", "tiny_vpt_deep", "tiny_upt", "tiny_upt_samedim", "tiny_upt_cut",
              "tiny_task_mask", "tiny_soft_labels",
              "tiny_upt_mask_soft"]        # UPT + per-task logit mask + soft multi-hot labels together (BASELINE cfg4 / cfg5 shape)
# TRAINER.MVLPT.VPT.PROJECT > -1 / VPT.DROPOUT > 0 (oracle/make_golden.py vptopt): carry `vpt_proj.*` parameters and the masks
VPT_OPTION_CASES = ["tiny_vpt_project", "tiny_vpt_project_dropout", "tiny_vpt_shallow_dropout"]
FULL_CASES = ["full_vitb32_coop_end", "full_vitb16_coop_middle", "full_vitb16_vpt_deep", "full_vitb16_upt_cut",
              "full_vitl14_336_upt_cut"]


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def tiny_state_dict():
    return {k: t(v) for k, v in load_npz("tiny_clip").items()}


def case_params(case):
    """prompt_learner tensors of a case, keyed by state_dict name."""
    return {k[len("param_"):]: t(v) for k, v in case.items() if k.startswith("param_")}


def case_grads(case):
    return {k[len("grad_"):]: t(v) for k, v in case.items() if k.startswith("grad_")}


def full_case_inputs(case, sd_with_tok, image_size=224):
    """Regenerate the inputs of an output-only fixture exactly as oracle/make_golden.py drew them:
    image from its own seeded stream; token_prefix/suffix from the synthetic token-embedding table."""
    g = torch.Generator().manual_seed(int(case["image_seed"]))
    B = case["out_logits"].shape[0]
    image = torch.randn(B, 3, image_size, image_size, generator=g)
    ids = t(case["tokenized_prompts"])
    emb = sd_with_tok["token_embedding.weight"][ids]
    n_ctx = int(case["meta_coop_n_ctx"])
    return image, emb[:, :1].contiguous(), emb[:, 1 + n_ctx:].contiguous()
