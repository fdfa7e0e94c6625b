"""CPU tests of the host-side mirror of the reference interface: prompt layouts, tokenizer contract, config,
LR schedule, state_dict keys (no GPU needed: nothing here launches a kernel)."""
import numpy as np
import pytest
import torch

from tests.golden_util import TINY_CASES, load_npz


def test_layout_tables_bit_exact_vs_reference():
    from mvlpt_amd.model import build_prompt_layout
    for name in TINY_CASES:
        c = load_npz(name)
        L = c["tokenized_prompts"].shape[1]
        tab = build_prompt_layout(c["name_lens"].tolist(), int(c["meta_coop_n_ctx"]), L, str(c["meta_position"]))
        assert tab.dtype == torch.int32 and np.array_equal(tab.numpy(), c["layout"]), name


def test_layout_rejects_unknown_position():
    from mvlpt_amd.model import build_prompt_layout
    with pytest.raises(ValueError):
        build_prompt_layout([1, 2], 4, 20, "sideways")          # trainers/mvlpt.py:512-513


def test_synthetic_tokenizer_contract():
    from mvlpt_amd.model import EOT_TOKEN, SOT_TOKEN, SyntheticTokenizer
    tok = SyntheticTokenizer()
    ids = tok.tokenize(["X X X X grand piano.", "X X X X dog."], context_length=20)
    assert ids.shape == (2, 20) and ids.dtype == torch.long
    assert ids[0, 0] == SOT_TOKEN and ids[0].max() == EOT_TOKEN
    # EOT index = n_ctx + name_len + 2
    assert ids.argmax(-1).tolist() == [4 + 2 + 2, 4 + 1 + 2]
    with pytest.raises(RuntimeError):
        tok.tokenize("a b c d e f", context_length=4)             # clip/clip.py:218-219


def test_cfg_overrides_like_train_py():
    from mvlpt_amd.config import get_cfg_default
    cfg = get_cfg_default()
    assert cfg.TRAINER.MVLPT.COOP.CLASS_TOKEN_POSITION == "middle" and cfg.TRAINER.MVLPT.PROJECT_DIM == 128
    cfg.merge_from_list(["TRAINER.MVLPT.VPT.N_CTX", "4", "TRAINER.CUT_CONTEXTLEN", "True", "OPTIM.LR", "0.01"])
    assert cfg.TRAINER.MVLPT.VPT.N_CTX == 4 and cfg.TRAINER.CUT_CONTEXTLEN is True and cfg.OPTIM.LR == 0.01
    with pytest.raises(KeyError):
        cfg.merge_from_list(["TRAINER.NOPE", "1"])


def test_lr_schedule_constant_warmup_then_cosine():
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.trainer import build_lr_scheduler, build_optimizer
    cfg = get_cfg_default()
    cfg.OPTIM.MAX_EPOCH = 10
    m = torch.nn.Linear(2, 2)
    opt = build_optimizer(m, cfg.OPTIM)
    assert isinstance(opt, torch.optim.SGD) and opt.defaults["momentum"] == 0.9 and opt.defaults["weight_decay"] == 5e-4
    sch = build_lr_scheduler(opt, cfg.OPTIM)
    lrs = []
    for _ in range(10):
        lrs.append(opt.param_groups[0]["lr"])
        sch.step()
    assert lrs[0] == 1e-5                                          # WARMUP_CONS_LR for WARMUP_EPOCH=1
    assert abs(lrs[1] - 0.5 * 0.002 * (1 + np.cos(np.pi * 1 / 10))) < 1e-12
    assert all(a > b for a, b in zip(lrs[1:], lrs[2:]))
