"""TRAINER.MVLPT.COOP.CTX_INIT — context vectors initialised from words (/root/reference/trainers/mvlpt.py:203-212; the reference's
*_ctxv1.yaml configs) — pinned by fixtures generated through the REAL reference (`python oracle/make_golden.py ctxinit`):
the words' token embeddings become `ctx`, their count overrides COOP.N_CTX, VPT.CTX_INIT raises.  CPU: host logic (initial ctx,
n_ctx, layout, prompt buffers bit-exact) and the oracle against the fixture; GPU: the HIP path at the north_star 1e-3."""
import numpy as np
import pytest
import torch

from tests.golden_util import case_grads, load_npz, t, tiny_state_dict

CTXINIT_CASES = ["tiny_coop_ctxinit", "tiny_upt_ctxinit"]


def _cfg(case):
    from tests.test_hip_model import cfg_for_case
    cfg = cfg_for_case(case, 32)
    cfg.TRAINER.MVLPT.COOP.CTX_INIT = str(case["meta_coop_ctx_init"])
    cfg.TRAINER.MVLPT.COOP.N_CTX = int(case["meta_coop_n_ctx_cfg"])      # what the config asked for; the words decide
    return cfg


def _sd_with_tokens():
    from mvlpt_amd.weights import ARCHS, make_state_dict
    sd = make_state_dict(ARCHS["tiny"], 1, include_token_embedding=True)      # oracle/make_golden.py TINY_SEED
    ref = tiny_state_dict()
    assert all(torch.equal(sd[k].reshape(-1), v.reshape(-1)) for k, v in ref.items())      # the same frozen weights as tiny_clip.npz
    return sd


@pytest.mark.parametrize("name", CTXINIT_CASES)
def test_ctx_init_host_logic_and_oracle(name):
    from mvlpt_amd.model import CustomCLIP, PretokenizedPrompts
    from mvlpt_amd.weights import ARCHS
    from oracle import clip_oracle as O
    from tests.fake_engine import OracleFrozenCLIP
    from tests.test_oracle_golden import run_oracle_on_case
    case = load_npz(name)
    sd = _sd_with_tokens()
    clip = OracleFrozenCLIP(sd, ARCHS["tiny"])
    clip._emb = sd["token_embedding.weight"].float()
    from mvlpt_amd.model import default_tokenizer
    clip.tokenizer = default_tokenizer()                                      # the words go through the real BPE table
    torch.manual_seed(int(case["case_seed"]))
    C = case["out_logits"].shape[1]
    pre = PretokenizedPrompts(t(case["tokenized_prompts"]), case["name_lens"].tolist())
    model = CustomCLIP(_cfg(case), [f"c{i}" for i in range(C)], clip, pretokenized=pre)
    pl = model.prompt_learner
    n_words = len(str(case["meta_coop_ctx_init"]).replace("_", " ").split(" "))
    assert pl.coop_n_ctx == n_words == int(case["meta_coop_n_ctx"]) and pl.ctx.shape == (n_words, ARCHS["tiny"].transformer_width)
    assert np.array_equal(pl.ctx.detach().numpy(), case["param_ctx"]), "ctx must start as the words' token embeddings, bit for bit"
    assert np.array_equal(pl.layout.numpy(), case["layout"]) and np.array_equal(pl.eot.numpy().astype(np.int64), case["eot"])
    assert np.array_equal(pl.token_prefix.numpy(), case["token_prefix"]) and np.array_equal(pl.token_suffix.numpy(), case["token_suffix"])
    # the oracle on the fixture's inputs
    res, _ = run_oracle_on_case(case, tiny_state_dict(), t(case["image"]), t(case["token_prefix"]), t(case["token_suffix"]), 2, 2)
    np.testing.assert_allclose(res.logits.numpy(), case["out_logits"], rtol=2e-4, atol=1e-5)
    for k, g in case_grads(case).items():
        assert float((res.grads[k] - g).abs().max()) / (float(g.abs().max()) + 1e-12) < 5e-5, k


def test_vpt_ctx_init_is_refused_as_in_the_reference():
    """trainers/mvlpt.py:180-182 raises ValueError("CTX initiation scheme is not supported") (checked against the real class by
    oracle/make_golden.py ctxinit)."""
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.model import CustomCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    from tests.fake_engine import OracleFrozenCLIP
    cfg = get_cfg_default()
    cfg.INPUT.SIZE = (32, 32)
    cfg.TRAINER.MVLPT.VPT.N_CTX, cfg.TRAINER.MVLPT.VPT.CTX_INIT = 2, "a photo"
    with pytest.raises(ValueError, match="not supported"):
        CustomCLIP(cfg, ["dog", "cat"], OracleFrozenCLIP(make_state_dict(ARCHS["tiny"], seed=5), ARCHS["tiny"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CTXINIT_CASES)
def test_ctx_init_fp16_on_the_hip_engine(name):
    from mvlpt_amd.model import CustomCLIP, FrozenCLIP, PretokenizedPrompts
    from tests.test_hip_model import GRAD_TOL_FP16, TOL_TINY_FP16, run_case
    case = load_npz(name)
    clip = FrozenCLIP(_sd_with_tokens(), compute_dtype="fp16")
    torch.manual_seed(int(case["case_seed"]))
    C = case["out_logits"].shape[1]
    pre = PretokenizedPrompts(t(case["tokenized_prompts"]), case["name_lens"].tolist())
    model = CustomCLIP(_cfg(case), [f"c{i}" for i in range(C)], clip, pretokenized=pre)
    pl = model.prompt_learner
    assert np.array_equal(pl.ctx.detach().cpu().numpy(), case["param_ctx"])
    # every other trainable tensor from the fixture (the random initialisers are drawn in the reference's order, but nn.Linear
    # defaults were overwritten by the generator); ctx keeps the value the words gave it
    sd = {k[len("param_"):]: t(v) for k, v in case.items() if k.startswith("param_")}
    sd["token_prefix"], sd["token_suffix"] = pl.token_prefix.clone(), pl.token_suffix.clone()
    pl.load_state_dict(sd, strict=True)
    model = model.to(clip.device)
    err, worst = run_case(case, model, t(case["image"]), TOL_TINY_FP16, GRAD_TOL_FP16)
    print(f"{name}: logits {err:.2e} grads {max(worst.values()):.2e}")
