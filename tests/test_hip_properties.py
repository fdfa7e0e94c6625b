"""Size-independent properties of the HIP path at BASELINE.json's FULL headline size (ViT-B/16, batch 256, 100 classes,
CoOp-16 `middle`, text L = 77; plus VPT-deep / UPT at batch 64), where the CPU oracle would take minutes per step:

  * rows are independent: permuting the images permutes the logits rows, permuting the classes permutes the columns,
    a batch evaluated in two halves equals the whole batch — BIT-EXACT (same kernels, same K order per output row);
  * the backward is linear in d(logits): 2x the loss gives exactly 2x every prompt gradient (device-side power-of-two
    gradient scaling must be transparent), and a zero upstream gradient gives exactly zero;
  * cosine logits are bounded by exp(logit_scale); the multiplicative task mask zeroes out-of-task columns exactly and
    leaves in-task columns untouched (trainers/mvlpt.py:575-581);
  * mean cross-entropy of B copies of one image equals the single-image loss and gradient (tolerance: fp32 reduction
    order over the batch only);
  * one plain-SGD step along the HIP gradient lowers the HIP loss (the gradient is a descent direction)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(method="coop", C=100, n_ctx=16, n_vpt=8, arch_name="ViT-B/16", tasks=None, seed=0, class_list=None):
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.model import CustomCLIP, FrozenCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    arch = ARCHS[arch_name]
    sd = make_state_dict(arch, seed=3)
    cfg = get_cfg_default()
    cfg.INPUT.SIZE = (arch.image_resolution, arch.image_resolution)
    T = cfg.TRAINER.MVLPT
    T.COOP.N_CTX = n_ctx if method in ("coop", "upt") else 0
    T.COOP.CLASS_TOKEN_POSITION = "middle"
    T.VPT.N_CTX = n_vpt if method in ("vpt", "upt") else 0
    T.VPT.DEEP = True
    T.PROJECT_DIM = 128 if method == "upt" else -1
    T.PROJECT_METHOD = "transformer" if method == "upt" else "identity"
    dm = None
    if tasks:
        cfg.DATASET.MULTITASK_LABEL_PERTASK = True

        class DM:
            _num_classes = C
            _task_names = [f"t{i}" for i in range(len(tasks))]
            _labelmap = {f"t{i}": list(range(c)) for i, c in enumerate(tasks)}
        dm = DM()
    torch.manual_seed(seed)
    names = [f"class number {i}" if i % 3 else f"c{i}" for i in range(C)]
    pre = None
    if class_list is not None:       # the reference tokenizer's own ids for a BASELINE class list (mvlpt_amd/data/class_prompts.npz)
        from mvlpt_amd.class_prompts import load_class_prompts
        pre, n = load_class_prompts(class_list, T.COOP.N_CTX)
        assert n == C
    # synthetic class names get the hash tokenizer (ids independent of the BPE table: the step sizes / tolerances of the tests
    # below were calibrated on them); BASELINE class lists come pre-tokenised by the reference tokenizer
    from mvlpt_amd.model import SyntheticTokenizer
    model = CustomCLIP(cfg, names, FrozenCLIP(sd, "fp16", tokenizer=SyntheticTokenizer()), dm=dm, pretokenized=pre).cuda()
    return arch, model


def _images(arch, B, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, arch.image_resolution, arch.image_resolution, generator=g).cuda()


def _grads(model):
    return {k: p.grad.detach().clone() for k, p in model.prompt_learner.named_parameters() if p.grad is not None}


@pytest.fixture(scope="module")
def headline():
    return _model("coop", C=100, n_ctx=16)


def test_image_rows_are_independent_at_headline_size(headline):
    arch, model = headline
    B = 256
    x = _images(arch, B)
    with torch.no_grad():
        full = model(x)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).cuda()
        assert torch.equal(model(x[perm]), full[perm]), "permuting the images must permute the logits rows bit-exactly"
        halves = torch.cat([model(x[:128].contiguous()), model(x[128:].contiguous())])
        assert torch.equal(halves, full), "a batch evaluated in two halves must equal the whole batch bit-exactly"
    assert full.shape == (B, 100) and torch.isfinite(full).all()
    bound = math.exp(math.log(1 / 0.07)) * (1 + 1e-5)
    assert float(full.abs().max()) <= bound, "cosine logits are bounded by exp(logit_scale)"


def test_class_rows_are_independent_at_headline_size():
    arch, m1 = _model("coop", C=100)
    x = _images(arch, 32)
    perm = torch.randperm(100, generator=torch.Generator().manual_seed(7))
    pl = m1.prompt_learner
    # second model: same prompts, classes in permuted order
    _, m2 = _model("coop", C=100)
    pl2 = m2.prompt_learner
    with torch.no_grad():
        pl2.ctx.copy_(pl.ctx)
        pl2.token_prefix.copy_(pl.token_prefix[perm.to(pl.token_prefix.device)])
        pl2.token_suffix.copy_(pl.token_suffix[perm.to(pl.token_suffix.device)])
        pl2.layout.copy_(pl.layout[perm.to(pl.layout.device)])
        pl2.eot.copy_(pl.eot[perm.to(pl.eot.device)])
        pl2.tokenized_prompts.copy_(pl.tokenized_prompts[perm.to(pl.tokenized_prompts.device)])
        a, b = m1(x), m2(x)
    assert torch.equal(a[:, perm.cuda()], b), "permuting the classes must permute the logits columns bit-exactly"


@pytest.mark.parametrize("method,B", [("coop", 256), ("vpt", 64), ("upt", 64)])
def test_backward_is_linear_and_scale_transparent(method, B):
    arch, model = _model(method, C=100, n_ctx=16 if method == "coop" else 4, n_vpt=8 if method == "vpt" else 4)
    x = _images(arch, B)
    y = torch.randint(0, 100, (B,), generator=torch.Generator().manual_seed(2)).cuda()

    def grads(scale):
        model.zero_grad(set_to_none=True)
        loss = model.cross_entropy(model(x), y) * scale
        loss.backward()
        return float(loss.detach()), _grads(model)

    l1, g1 = grads(1.0)
    l4, g4 = grads(4.0)
    lz, gz = grads(0.0)
    assert g1 and set(g1) == set(g4)
    direct = [k for k in g1 if not k.startswith("mvlpt_proj")] if method != "upt" else []
    for k in g1:
        assert torch.isfinite(g1[k]).all() and float(g1[k].abs().max()) > 0, k
        if k in direct:      # straight out of the HIP backward: power-of-two scaling must be exact
            assert torch.equal(g4[k], 4.0 * g1[k]), f"{k}: 4x loss must give exactly 4x gradient"
        else:                # UPT: passes through torch autograd of the projection (fp32 GEMM reassociation)
            assert float((g4[k] - 4.0 * g1[k]).abs().max()) <= 1e-5 * float(g4[k].abs().max()), k
        assert float(gz[k].abs().max()) == 0.0, f"{k}: zero upstream gradient must give zero"
    assert abs(l4 - 4 * l1) <= 1e-5 * abs(l4)


def test_task_mask_zeroes_out_of_task_columns_exactly():
    tasks = [40, 35, 25]
    arch, model = _model("coop", C=100, tasks=tasks)
    _, unmasked = _model("coop", C=100)        # same seed -> same context vectors; no per-task label space
    assert torch.equal(unmasked.prompt_learner.ctx, model.prompt_learner.ctx)
    B = 96
    x = _images(arch, B)
    task = torch.randint(0, 3, (B,), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        masked = model(x, task=task)
        plain = unmasked(x)
    lo = torch.tensor([0, 40, 75])[task]
    hi = torch.tensor([40, 75, 100])[task]
    cols = torch.arange(100).view(1, -1)
    inside = ((cols >= lo.view(-1, 1)) & (cols < hi.view(-1, 1))).cuda()
    assert torch.equal(masked[inside], plain[inside])
    assert float(masked[~inside].abs().max()) == 0.0, "mask is multiplicative 0/1: out-of-task logits are exactly 0"


def test_mean_loss_over_copies_equals_single_image():
    arch, model = _model("vpt", C=100, n_vpt=8)
    x1 = _images(arch, 1)
    y1 = torch.tensor([17]).cuda()

    def run(x, y):
        model.zero_grad(set_to_none=True)
        loss = model.cross_entropy(model(x), y)
        loss.backward()
        return float(loss.detach()), _grads(model)

    l1, g1 = run(x1, y1)
    l8, g8 = run(x1.expand(64, -1, -1, -1).contiguous(), y1.expand(64).contiguous())
    assert abs(l1 - l8) <= 1e-6 * max(1.0, abs(l1))
    for k in g1:
        # 64 identical per-image gradients of weight 1/64 each: only the fp32 summation order and the 16-bit
        # rounding of the (64x smaller) upstream gradient differ
        assert float((g1[k] - g8[k]).abs().max()) <= 2e-3 * float(g1[k].abs().max()), k


@pytest.mark.parametrize("method", ["coop", "vpt"])
def test_gradient_is_a_descent_direction_at_full_size(method):
    arch, model = _model(method, C=100)
    B = 128
    x = _images(arch, B)
    y = torch.randint(0, 100, (B,), generator=torch.Generator().manual_seed(4)).cuda()
    params = [p for p in model.prompt_learner.parameters() if p.requires_grad]
    loss0 = model.cross_entropy(model(x), y)
    loss0.backward()
    gnorm2 = sum(float((p.grad ** 2).sum()) for p in params)
    assert gnorm2 > 0
    step = 0.05 / math.sqrt(gnorm2)          # small normalised step
    with torch.no_grad():
        for p in params:
            p.add_(p.grad, alpha=-step)
        loss1 = model.cross_entropy(model(x), y)
    predicted = step * gnorm2
    drop = float(loss0.detach()) - float(loss1)
    assert drop > 0, f"loss did not go down: {float(loss0):.6f} -> {float(loss1):.6f}"
    assert 0.5 * predicted < drop < 1.5 * predicted, f"first-order prediction {predicted:.3e} vs actual drop {drop:.3e}"


@pytest.mark.parametrize("method,B", [("coop", 256), ("upt", 64)])
def test_step_is_bitwise_deterministic(method, B):
    """No atomics with order-dependent results, no run-to-run variation from the multi-stream overlap: two identical
    steps give identical logits, loss and prompt gradients, bit for bit."""
    arch, model = _model(method, C=100, n_ctx=16 if method == "coop" else 4, n_vpt=4)
    x = _images(arch, B)
    y = torch.randint(0, 100, (B,), generator=torch.Generator().manual_seed(9)).cuda()

    def run():
        model.zero_grad(set_to_none=True)
        logits = model(x)
        loss = model.cross_entropy(logits, y)
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), loss.detach().clone(), _grads(model)

    l0, s0, g0 = run()
    for _ in range(3):
        l1, s1, g1 = run()
        assert torch.equal(l0, l1) and torch.equal(s0, s1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k


# ---------------------------------------------------------------------------------------------- many-class configurations
# BASELINE configs[2] (per-GPU shape: VPT-deep 8 tokens/layer, ImageNet-1k = 1000 classes, batch 256) and configs[3]
# (UPT 4+4, the 11-dataset CoOp multitask set = 2191 classes, per-task logit mask AND soft labels together, batch 256),
# with the reference tokenizer's ids for those class lists.  The oracle would need tens of minutes per step here.

def test_cfg3_vpt_deep_1000_classes_batch_256():
    arch, model = _model("vpt", C=1000, n_vpt=8, class_list="imagenet1k")
    B = 256
    x = _images(arch, B)
    y = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(12)).cuda()

    def run(scale=1.0):
        model.zero_grad(set_to_none=True)
        logits = model(x)
        loss = model.cross_entropy(logits, y) * scale
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), float(loss.detach()), _grads(model)

    l0, s0, g0 = run()
    txt = model._const_text_features
    assert txt is not None and txt.shape == (1000, 512), "no text context: the 1000 text features are constants and must be cached"
    l1, s1, g1 = run()
    assert model._const_text_features is txt, "cached text features must be reused, not recomputed every step"
    assert torch.equal(l0, l1) and s0 == s1 and all(torch.equal(g0[k], g1[k]) for k in g0), "bitwise deterministic"
    assert set(g0) == {"vpt_embeddings", "vpt_embeddings_deep"} and g0["vpt_embeddings_deep"].shape == (11, 8, 768)
    _, s4, g4 = run(4.0)
    for k in g0:
        assert torch.isfinite(g0[k]).all() and float(g0[k].abs().max()) > 0
        assert torch.equal(g4[k], 4.0 * g0[k]), f"{k}: 4x loss must give exactly 4x gradient"
    assert l0.shape == (B, 1000) and float(l0.abs().max()) <= math.exp(math.log(1 / 0.07)) * (1 + 1e-5)
    with torch.no_grad():      # (inference forwards run the single-operand mode: compare them with each other)
        full = model(x)
        halves = torch.cat([model(x[:128].contiguous()), model(x[128:].contiguous())])
    assert torch.equal(halves, full)
    assert float((full - l0).abs().max()) <= 1e-3 * float(l0.abs().max()), "training and inference forwards agree to 1e-3"
    # cross-entropy at C = 1000 against torch on the same logits
    ref = torch.nn.functional.cross_entropy(l0, y)
    assert abs(float(ref) - s0) < 1e-5 * max(1.0, s0)


def test_cfg4_upt_2191_classes_task_mask_and_soft_labels_batch_256():
    from mvlpt_amd.class_prompts import task_class_counts
    tasks = task_class_counts("coop11")
    assert len(tasks) == 11 and sum(tasks) == 2191
    arch, model = _model("upt", C=2191, n_ctx=4, n_vpt=4, tasks=tasks, class_list="coop11")
    B, C = 256, 2191
    x = _images(arch, B)
    g = torch.Generator().manual_seed(13)
    task = torch.randint(0, len(tasks), (B,), generator=g)
    starts = torch.tensor([0] + list(torch.tensor(tasks).cumsum(0)[:-1]))
    lo, hi = starts[task], starts[task] + torch.tensor(tasks)[task]
    cols = torch.arange(C).view(1, -1)
    inside = (cols >= lo.view(-1, 1)) & (cols < hi.view(-1, 1))
    soft = ((torch.rand(B, C, generator=g) > 0.7) & inside).float()
    soft[torch.arange(B), lo + (torch.rand(B, generator=g) * (hi - lo)).long()] = 1.0      # at least one positive, in range
    soft = (soft / soft.sum(-1, keepdim=True)).cuda()                                      # trainers/mvlpt.py:914-916

    def run(scale=1.0):
        model.zero_grad(set_to_none=True)
        logits = model(x, task=task)
        loss = model.cross_entropy(logits, soft) * scale
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), float(loss.detach()), _grads(model)

    l0, s0, g0 = run()
    assert l0.shape == (B, C) and torch.isfinite(l0).all()
    assert float(l0[~inside.cuda()].abs().max()) == 0.0, "out-of-task logits are exactly 0 (multiplicative mask, :578-581)"
    assert float(l0[inside.cuda()].abs().max()) > 0
    # soft-label cross-entropy (probability targets) against torch on the same logits; masked zeros stay in the softmax
    ref = torch.nn.functional.cross_entropy(l0, soft)
    assert abs(float(ref) - s0) < 1e-5 * max(1.0, s0)
    assert {"ctx", "vpt_embeddings", "vpt_embeddings_deep"} <= set(g0) and len(g0) == 23      # + 20 projection tensors (PROJECT_DIM 128: pre/post Linears on both sides)
    _, _, g4 = run(4.0)
    _, _, gz = run(0.0)
    for k in g0:
        assert torch.isfinite(g0[k]).all() and float(g0[k].abs().max()) > 0, k
        assert float((g4[k] - 4.0 * g0[k]).abs().max()) <= 1e-5 * float(g4[k].abs().max()), k
        assert float(gz[k].abs().max()) == 0.0, k
    # descent direction: one small normalised SGD step lowers the (masked, soft-label) loss by the first-order amount
    params = [p for p in model.prompt_learner.parameters() if p.requires_grad]
    _, s_before, _ = run()
    gnorm2 = sum(float((p.grad ** 2).sum()) for p in params)
    step = 0.02 / math.sqrt(gnorm2)
    with torch.no_grad():
        for p in params:
            p.add_(p.grad, alpha=-step)
        s_after = float(model.cross_entropy(model(x, task=task), soft))
    drop, predicted = s_before - s_after, step * gnorm2
    assert drop > 0 and 0.5 * predicted < drop < 1.5 * predicted, (drop, predicted)


def test_cfg5_vitl14_336_upt_1151_classes_batch_128():
    """BASELINE configs[4], full per-GPU size: ViT-L/14@336 (577 + 4 tokens per image: the streamed pair attention inside a
    24-layer tower), UPT 4 + 4, the ELEVATER-20 class list (1151 classes, reference tokenizer ids), batch 128.  The oracle
    cannot run this in test time; what the size cannot hide is checked instead: bitwise determinism, exact 4x linearity in the
    loss scale (power-of-two scaling commutes with every rounding of the backward), zero gradient for a zero loss, image rows
    independent of their batch, cross-entropy against torch on the same logits."""
    arch, model = _model("upt", C=1151, n_ctx=4, n_vpt=4, arch_name="ViT-L/14@336px", class_list="elevater20")
    B, C = 128, 1151
    x = _images(arch, B)
    y = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(14)).cuda()

    def run(scale=1.0):
        model.zero_grad(set_to_none=True)
        logits = model(x)
        loss = model.cross_entropy(logits, y) * scale
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), float(loss.detach()), _grads(model)

    l0, s0, g0 = run()
    assert l0.shape == (B, C) and torch.isfinite(l0).all()
    assert float(l0.abs().max()) <= math.exp(math.log(1 / 0.07)) * (1 + 1e-5)
    assert abs(float(torch.nn.functional.cross_entropy(l0, y)) - s0) < 1e-5 * max(1.0, s0)
    l1, s1, g1 = run()
    assert torch.equal(l0, l1) and s0 == s1 and all(torch.equal(g0[k], g1[k]) for k in g0), "bitwise deterministic"
    assert {"ctx", "vpt_embeddings", "vpt_embeddings_deep"} <= set(g0) and g0["vpt_embeddings_deep"].shape == (23, 4, 1024)
    _, _, g4 = run(4.0)
    _, _, gz = run(0.0)
    for k in g0:
        assert torch.isfinite(g0[k]).all() and float(g0[k].abs().max()) > 0, k
        assert float((g4[k] - 4.0 * g0[k]).abs().max()) <= 1e-5 * float(g4[k].abs().max()), k
        assert float(gz[k].abs().max()) == 0.0, k
    with torch.no_grad():      # inference forwards (single operands): a batch in two halves equals the whole batch bit-exactly
        full = model(x)
        halves = torch.cat([model(x[:64].contiguous()), model(x[64:].contiguous())])
    assert torch.equal(halves, full)
    # (the inference forward keeps single 16-bit operands in the image tower — 24 layers of 581 tokens —, the text tower runs with
    # split operands in both; the training forward sits 3e-5 from the reference (full_vitl14_336 fixture))
    assert float((full - l0).abs().max()) <= 1e-3 * float(l0.abs().max()), "training and inference forwards agree to 1e-3"
