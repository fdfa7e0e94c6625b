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
    # Dassl's ConstantWarmupScheduler steps its successor only after the warm-up: torch's own CosineAnnealingLR,
    # started one epoch late, is the comparator (first post-warm-up epoch at the full base LR)
    ref_opt = torch.optim.SGD(torch.nn.Linear(2, 2).parameters(), lr=0.002)
    ref = torch.optim.lr_scheduler.CosineAnnealingLR(ref_opt, T_max=10)
    for e in range(1, 10):
        assert abs(lrs[e] - ref_opt.param_groups[0]["lr"]) < 1e-12, e
        ref_opt.step()
        ref.step()
    assert lrs[1] == 0.002 and all(a > b for a, b in zip(lrs[1:], lrs[2:]))


def test_lr_schedule_linear_warmup_and_rejections():
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.trainer import build_lr_scheduler, build_optimizer
    cfg = get_cfg_default()
    cfg.OPTIM.MAX_EPOCH, cfg.OPTIM.WARMUP_EPOCH, cfg.OPTIM.WARMUP_TYPE = 10, 3, "linear"
    opt = build_optimizer(torch.nn.Linear(2, 2), cfg.OPTIM)
    sch = build_lr_scheduler(opt, cfg.OPTIM)
    lrs = []
    for _ in range(5):
        lrs.append(opt.param_groups[0]["lr"])
        sch.step()
    assert lrs[0] == 1e-5 and abs(lrs[1] - 0.002 / 3) < 1e-12 and abs(lrs[2] - 0.004 / 3) < 1e-12 and lrs[3] == 0.002
    cfg.OPTIM.WARMUP_TYPE = "exponential"
    with pytest.raises(ValueError):
        build_lr_scheduler(opt, cfg.OPTIM)
    cfg.OPTIM.WARMUP_TYPE, cfg.OPTIM.LR_SCHEDULER = "constant", "multi_step"
    with pytest.raises(ValueError):
        build_lr_scheduler(opt, cfg.OPTIM)


def test_class_shard_bounds_balanced():
    from mvlpt_amd.model import class_shard_bounds
    assert class_shard_bounds(9, 4) == [(0, 3), (3, 5), (5, 7), (7, 9)]          # ceil-split would leave rank 3 empty
    assert class_shard_bounds(2191, 8)[-1][1] == 2191
    for C, W in [(5, 2), (100, 8), (1151, 8), (8, 8)]:
        b = class_shard_bounds(C, W)
        assert b[0][0] == 0 and b[-1][1] == C and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1 and min(h - l for l, h in b) >= 1


def _tiny_oracle_model(n_ctx=4, names=("dog", "grand piano", "sea horse")):
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.model import CustomCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    from tests.fake_engine import OracleFrozenCLIP
    arch = ARCHS["tiny"]
    cfg = get_cfg_default()
    cfg.INPUT.SIZE = (32, 32)
    cfg.TRAINER.MVLPT.COOP.N_CTX = n_ctx
    torch.manual_seed(3)
    return CustomCLIP(cfg, list(names), OracleFrozenCLIP(make_state_dict(arch, seed=5), arch)), cfg


def test_stale_backward_is_refused():
    """the engine keeps one set of saved activations: backward of an older forward must raise, not use the wrong ones"""
    model, _ = _tiny_oracle_model()
    img = torch.randn(2, 3, 32, 32)
    loss_a = model.cross_entropy(model(img), torch.tensor([0, 1]))
    loss_b = model.cross_entropy(model(img + 1.0), torch.tensor([1, 2]))
    with pytest.raises(RuntimeError, match="stale forward"):
        loss_a.backward()
    loss_b.backward()
    assert model.prompt_learner.ctx.grad is not None


def test_init_weights_drop_class_buffers_and_mismatched_shapes(tmp_path):
    """MODEL.INIT_WEIGHTS from a checkpoint with ANOTHER class count (trainers/mvlpt.py:864-865 + Dassl
    load_pretrained_weights): ctx is taken, token_prefix / token_suffix are not."""
    from mvlpt_amd.trainer import load_pretrained_weights
    src, _ = _tiny_oracle_model(names=("a", "b", "c", "d", "e"))
    dst, _ = _tiny_oracle_model()
    sd = {k: v.clone() for k, v in src.prompt_learner.state_dict().items()}
    sd["ctx"] += 1.0
    path = tmp_path / "init.pth.tar"
    torch.save({"state_dict": sd, "epoch": 3}, path)
    before = dst.prompt_learner.token_suffix.clone()
    load_pretrained_weights(dst.prompt_learner, str(path))
    assert torch.equal(dst.prompt_learner.ctx.data, sd["ctx"])
    assert torch.equal(dst.prompt_learner.token_suffix, before) and dst.prompt_learner.token_suffix.shape[0] == 3


def test_bench_clock_sampler_reads_hwmon(tmp_path, monkeypatch):
    """bench.py's clock / power sampler: picks the busy board among the hwmon directories, averages MHz and W; no board -> None."""
    import glob as _glob
    import importlib.util
    import os
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dirs = []
    for i, (hz, uw) in enumerate([(158e6, 250e6), (2010e6, 1300e6)]):
        d = tmp_path / f"card{i}" / "device" / "hwmon" / f"hwmon{i}"
        d.mkdir(parents=True)
        (d / "freq1_input").write_text(f"{int(hz)}\n")
        (d / "power1_input").write_text(f"{int(uw)}\n")
        (d / "power1_cap").write_text("1400000000\n")
        dirs.append(str(d))
    monkeypatch.setattr(_glob, "glob", lambda pat: dirs if "hwmon" in pat else [])
    s = bench._ClockSampler()
    assert s.cards == sorted(dirs)
    s.start()
    time.sleep(0.25)
    out = s.stop()
    assert out["sclk_mhz_avg"] == 2010.0 and out["power_w_avg"] == 1300.0 and out["power_cap_w"] == 1400.0 and out["samples"] >= 2
    monkeypatch.setattr(_glob, "glob", lambda pat: [])
    s = bench._ClockSampler()
    s.start()
    assert s.stop() is None


def test_lookahead_loader_semantics():
    """LookAheadLoader (the place the one-batch look-ahead lives, so that an unmodified Dassl run_epoch gets it): while the
    consumer holds batch i, owner.next_batch is batch i+1; None on the last batch, after exhaustion and after an early exit."""
    from mvlpt_amd.trainer import LookAheadLoader

    class Owner:
        next_batch = "stale"

    class Loader(list):
        batch_size = 7

    o = Owner()
    la = LookAheadLoader(Loader(["a", "b", "c"]), o)
    assert len(la) == 3 and la.batch_size == 7
    seen = [(b, o.next_batch) for b in la]
    assert seen == [("a", "b"), ("b", "c"), ("c", None)] and o.next_batch is None and la.hits == 2
    assert [(b, o.next_batch) for b in la] == seen                       # re-iterable (one pass per epoch)
    it = iter(la)
    assert next(it) == "a" and o.next_batch == "b"
    it.close()                                                           # a hook broke out of the loop
    assert o.next_batch is None
    assert list(LookAheadLoader(Loader([]), o)) == [] and o.next_batch is None


@pytest.mark.parametrize("name", ["tiny_vpt_project", "tiny_vpt_project_dropout", "tiny_vpt_shallow_dropout"])
def test_vpt_project_and_dropout_host_logic_on_the_oracle_engine(name):
    """The host side of VPT.PROJECT / VPT.DROPOUT (model.py: `vpt_proj` through torch autograd, per-image masks handed to the
    engine) around the CPU-oracle engine, against the reference fixture: the same CustomCLIP code path the HIP engine sits under."""
    import numpy as np
    from tests.fake_engine import OracleFrozenCLIP
    from tests.golden_util import case_grads, load_npz, t, tiny_state_dict
    from tests.test_hip_model import build_model
    from mvlpt_amd.weights import ARCHS
    case = load_npz(name)
    clip = OracleFrozenCLIP(tiny_state_dict(), ARCHS["tiny"])
    model = build_model(case, clip, 32, t(case["token_prefix"]), t(case["token_suffix"]))
    logits = model(t(case["image"]))
    loss = model.cross_entropy(logits, t(case["label"]))
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), case["out_logits"], rtol=2e-4, atol=1e-5)
    G = case_grads(case)
    got = {n: p.grad for n, p in model.prompt_learner.named_parameters()}
    assert set(got) == set(G)
    for k, g in G.items():
        assert float((got[k] - g).abs().max()) / (float(g.abs().max()) + 1e-12) < 5e-4, k
