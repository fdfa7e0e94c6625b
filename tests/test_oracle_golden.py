"""Pin the CPU oracle (oracle/clip_oracle.py) to golden vectors produced by the REAL reference
(oracle/make_golden.py). fp32 CPU on both sides: agreement ~1e-5; integer tables bit-exact."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as O
from tests.golden_util import (TINY_CASES, VPT_OPTION_CASES, case_grads, case_params, load_npz, t, tiny_state_dict)

RTOL = 2e-4   # fp32 reduction-order differences through 2-3 transformer layers
ATOL = 2e-6


def run_oracle_on_case(case, sd, image, token_prefix, token_suffix, vision_heads, text_heads):
    P = case_params(case)
    n_ctx, n_vpt = int(case["meta_coop_n_ctx"]), int(case["meta_vpt_n_ctx"])
    proj = {k: v for k, v in P.items() if k.startswith("mvlpt_proj")}
    label = t(case["label"])
    if label.dtype != torch.int64:
        label = label.float()
        label = label / label.sum(-1, keepdim=True)         # trainers/mvlpt.py:914-916
    mask = None
    if "task" in case:
        mask = O.task_mask(t(case["task"]), t(case["task_start"]), t(case["task_end"]), token_prefix.shape[0])
    L = case["tokenized_prompts"].shape[1]
    layout = O.build_prompt_layout(case["name_lens"].tolist(), n_ctx, L, str(case["meta_position"]))
    return O.forward_backward(
        sd, image=image, label=label, vision_heads=vision_heads, text_heads=text_heads,
        token_prefix=token_prefix, token_suffix=token_suffix, eot=t(case["eot"]), layout=layout,
        ctx=P.get("ctx"), vpt=P.get("vpt_embeddings"), vpt_deep=P.get("vpt_embeddings_deep"),
        proj_params=proj or None, n_ctx=n_ctx, n_vpt=n_vpt, mask=mask,
        vpt_proj=({k[len("vpt_proj."):]: v for k, v in P.items() if k.startswith("vpt_proj.")} or None),
        vpt_masks=(t(case["vpt_dropout_masks"]) if "vpt_dropout_masks" in case else None)), layout


@pytest.mark.parametrize("name", TINY_CASES + VPT_OPTION_CASES)
def test_tiny_case_matches_reference(name):
    case = load_npz(name)
    sd = tiny_state_dict()
    res, layout = run_oracle_on_case(case, sd, t(case["image"]), t(case["token_prefix"]), t(case["token_suffix"]), 2, 2)
    assert np.array_equal(layout.numpy(), case["layout"]), "prompt layout table must be bit-exact"
    np.testing.assert_allclose(res.logits.numpy(), case["out_logits"], rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(float(res.loss), float(case["out_loss"]), rtol=1e-5)
    G = case_grads(case)
    assert set(res.grads) == set(G), (sorted(res.grads), sorted(G))
    for k, g in G.items():
        scale = float(g.abs().max()) + 1e-12
        err = float((res.grads[k] - g).abs().max()) / scale
        assert err < 5e-5, f"{name}: grad {k} rel-to-max err {err:.3e}"      # measured <= 4.8e-6 (fp32 on both sides)


def test_eot_is_argmax_of_token_ids():
    z = load_npz("tokens")
    for n_ctx in (0, 4, 16):
        ids = z[f"ids_nctx{n_ctx}"]
        assert np.array_equal(ids.argmax(-1), z[f"eot_nctx{n_ctx}"])
        if n_ctx:
            # Appendix A.5: EOT index = n_ctx + name_len + 2, independent of class-token position
            assert np.array_equal(z[f"eot_nctx{n_ctx}"], n_ctx + z["name_lens"] + 2)
            assert int(z[f"cutlen_nctx{n_ctx}"]) == int((n_ctx + z["name_lens"] + 3).max())


def test_handwritten_backward_matches_autograd():
    """The dX-only backward restated by hand equals torch.autograd on the same forward."""
    torch.manual_seed(0)
    sd = tiny_state_dict()
    x = torch.randn(3, 7, 128, requires_grad=True)
    pre = "transformer.resblocks.1."
    y, saved = O.block_fwd(x, sd, pre, 2, causal=True)
    dy = torch.randn_like(y)
    y.backward(dy)
    with torch.no_grad():
        dx = O.block_bwd(dy, saved, sd, pre)
    np.testing.assert_allclose(dx.numpy(), x.grad.numpy(), rtol=1e-4, atol=1e-6)


def test_cross_entropy_matches_torch():
    torch.manual_seed(1)
    lg = torch.randn(6, 9, requires_grad=True)
    for lab in (torch.randint(0, 9, (6,)), torch.softmax(torch.randn(6, 9), -1)):
        lg.grad = None
        ref = torch.nn.functional.cross_entropy(lg, lab)
        ref.backward()
        loss, dl = O.cross_entropy_fwd_bwd(lg.detach(), lab)
        np.testing.assert_allclose(float(loss), float(ref), rtol=1e-6)
        np.testing.assert_allclose(dl.numpy(), lg.grad.numpy(), rtol=1e-5, atol=1e-7)
