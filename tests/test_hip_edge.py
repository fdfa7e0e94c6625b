"""Edge cases of the HIP path against the CPU oracle (tiny architecture, fp16): ragged and minimal sizes, single class /
single image, many classes, maximum text length with long class names, prompt counts that are not tile multiples,
and loud failures for unsupported / inconsistent inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import clip_oracle as O  # noqa: E402


def _setup(n_ctx, n_vpt, names, B, position="middle", csc=False, deep=True, seed=0, cut=False):
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.model import CustomCLIP, FrozenCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    arch = ARCHS["tiny"]
    sd = make_state_dict(arch, seed=21)
    cfg = get_cfg_default()
    cfg.INPUT.SIZE = (32, 32)
    T = cfg.TRAINER.MVLPT
    T.COOP.N_CTX, T.COOP.CSC, T.COOP.CLASS_TOKEN_POSITION = n_ctx, csc, position
    T.VPT.N_CTX, T.VPT.DEEP, T.PROJECT_DIM = n_vpt, deep, 64
    cfg.TRAINER.CUT_CONTEXTLEN = cut
    torch.manual_seed(seed)
    model = CustomCLIP(cfg, names, FrozenCLIP(sd, "fp16")).cuda()
    g = torch.Generator().manual_seed(seed + 1)
    image = torch.randn(B, 3, 32, 32, generator=g)
    label = torch.randint(0, len(names), (B,), generator=g)
    return arch, sd, model, image, label


def _check(arch, sd, model, image, label, tol=1e-3, gtol=1e-3):       # north_star tolerances
    pl = model.prompt_learner
    logits = model(image.cuda())
    loss = model.cross_entropy(logits, label.cuda())
    loss.backward()
    P = {k: v.detach().cpu() for k, v in pl.named_parameters()}
    proj = {k: v for k, v in P.items() if k.startswith("mvlpt_proj")}
    ref = O.forward_backward(
        sd, image=image, label=label, vision_heads=arch.vision_heads, text_heads=arch.transformer_heads,
        token_prefix=pl.token_prefix.cpu(), token_suffix=pl.token_suffix.cpu(), eot=pl.eot.cpu().long(), layout=pl.layout.cpu(),
        ctx=P.get("ctx"), vpt=P.get("vpt_embeddings"), vpt_deep=P.get("vpt_embeddings_deep"), proj_params=proj or None,
        n_ctx=pl.coop_n_ctx, n_vpt=pl.vpt_n_ctx)
    err = float((logits.detach().cpu() - ref.logits).abs().max()) / max(1.0, float(ref.logits.abs().max()))
    assert err < tol, f"logits {err}"
    assert abs(float(loss.detach()) - float(ref.loss)) < tol
    for k, g in ref.grads.items():
        got = dict(pl.named_parameters())[k].grad.cpu()
        e = float((got - g).abs().max()) / (float(g.abs().max()) + 1e-20)
        assert e < gtol, f"grad {k}: {e}"


@pytest.mark.parametrize("B,names", [(1, ["dog"]), (1, ["dog", "cat"]), (3, ["a b c d e", "x"]), (17, [f"n{i}" for i in range(7)])])
def test_minimal_and_ragged_batches_coop(B, names):
    _check(*_setup(4, 0, names, B))


def test_many_classes_text_tower():
    names = [f"class number {i}" for i in range(300)]           # 300 x 77 = 23k text tokens: multi-tile GEMMs, ragged M
    _check(*_setup(2, 0, names, 5, position="front"))


def test_long_class_names_fill_the_context():
    long_name = " ".join(["w"] * 58)                            # 58 words + 16 ctx + SOT/'.'/EOT = 77 = context length
    arch, sd, model, image, label = _setup(16, 0, [long_name, "dog"], 2, position="end")
    assert int(model.prompt_learner.eot.max()) == 76
    _check(arch, sd, model, image, label)
    with pytest.raises(RuntimeError):                           # one more word does not fit (clip/clip.py:218-219)
        _setup(16, 0, [long_name + " w"], 2)


@pytest.mark.parametrize("n_vpt,deep", [(1, True), (3, False), (5, True)])
def test_odd_visual_prompt_counts(n_vpt, deep):
    _check(*_setup(0, n_vpt, ["dog", "cat", "bird"], 6, deep=deep))


def test_upt_csc_cut_combo():
    _check(*_setup(3, 2, ["dog", "grand piano", "cat"], 4, csc=True, cut=True))


def test_loud_failures():
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.model import CustomCLIP, FrozenCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    arch, sd, model, image, label = _setup(4, 0, ["dog", "cat"], 2)
    with pytest.raises(RuntimeError):
        model(image)                                            # CPU tensor: refused, never computed on the host
    with pytest.raises((RuntimeError, ValueError)):
        model.cross_entropy(model(image.cuda()), torch.zeros(3, dtype=torch.long).cuda())   # wrong label count
    eng = model.engine
    with pytest.raises(RuntimeError):
        eng.image_bwd(torch.zeros(2, arch.embed_dim, device="cuda"))   # no saved forward
    with torch.no_grad():
        model(image.cuda())                                     # inference forward: nothing is saved for a backward
    with pytest.raises(RuntimeError):
        eng.text_bwd(torch.zeros(2, arch.embed_dim, device="cuda"))
    bad = dict(sd)
    bad["visual.ln_pre.weight"] = torch.ones(7)
    with pytest.raises(RuntimeError):
        FrozenCLIP(bad, "fp16")                                 # shape mismatch names the tensor
    cfg = get_cfg_default()
    cfg.INPUT.SIZE = (32, 32)
    cfg.TRAINER.MVLPT.COCOOP.N_CTX = 4
    with pytest.raises(NotImplementedError):
        CustomCLIP(cfg, ["dog"], FrozenCLIP(sd, "fp16"))        # CoCoOp is out of scope: refused, not emulated


def test_trim_releases_outgrown_workspaces_and_the_engine_keeps_working():
    """mvlpt_trim: workspaces only grow and retire what they outgrow; trim frees the retired blocks at an epoch
    boundary.  The engine must give the same answers before and after."""
    from mvlpt_amd.model import FrozenCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    arch = ARCHS["tiny"]
    eng = FrozenCLIP(make_state_dict(arch, seed=5)).engine
    g = torch.Generator().manual_seed(0)
    small = torch.randn(2, 3, arch.image_resolution, arch.image_resolution, generator=g).cuda()
    big = torch.randn(64, 3, arch.image_resolution, arch.image_resolution, generator=g).cuda()
    f_small = eng.image_fwd(small).clone()
    f_big = eng.image_fwd(big).clone()                       # the workspace grows: the small block is retired, not freed
    eng.trim()
    assert torch.equal(eng.image_fwd(small), f_small) and torch.equal(eng.image_fwd(big), f_big)
    eng.trim()                                                # nothing retired: a no-op
    assert torch.equal(eng.image_fwd(small), f_small)


@pytest.mark.parametrize("arch_name,B", [("tiny", 5), ("ViT-B/32", 3), ("ViT-B/16", 2)])
def test_image_dtypes_give_identical_features(arch_name, B):
    """mvlpt_image_fwd takes fp32, fp16 or bf16 images (the 16-bit ones of the compute type go through the 16-byte patchify
    fast path when the patch size allows): for pixel values both types hold exactly, the features are bit-identical."""
    from mvlpt_amd.model import FrozenCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    arch = ARCHS[arch_name]
    eng = FrozenCLIP(make_state_dict(arch, seed=3)).engine
    g = torch.Generator().manual_seed(11)
    x16 = torch.randn(B, 3, arch.image_resolution, arch.image_resolution, generator=g).half().cuda()
    f32 = eng.image_fwd(x16.float()).clone()
    f16 = eng.image_fwd(x16).clone()
    assert torch.isfinite(f32).all() and float(f32.abs().max()) > 0
    assert torch.equal(f16, f32), "fp16 and fp32 inputs holding the same pixel values must give the same features"
    # non-contiguous views are made contiguous by the binding; a different batch offset must not matter
    assert torch.equal(eng.image_fwd(x16[1:].contiguous()), f32[1:])
