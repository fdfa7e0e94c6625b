"""TEST INFRASTRUCTURE — generates tests/golden/*.npz from the REAL reference.

Run in the build container only (needs /root/reference; see oracle/ref_shim.py):

    python oracle/make_golden.py            # all fixtures
    python oracle/make_golden.py tiny       # only the tiny-arch ones
    python oracle/make_golden.py vptopt     # VPT.PROJECT / VPT.DROPOUT cases (tiny arch)
    python oracle/make_golden.py ctxinit    # COOP.CTX_INIT (context initialised from words) cases (tiny arch)

The reference's `trainers.mvlpt.CustomCLIP` (trainers/mvlpt.py:517-583) is instantiated on a
`clip.model.CLIP` (clip/model.py:239-322) whose weights come from OUR deterministic generator
(`mvlpt_amd.weights.make_state_dict`), run forward + `F.cross_entropy` + `backward()` on CPU fp32
(the reference's own CPU path, trainers/mvlpt.py:910-932), and inputs/outputs are stored as data.
No reference source, bytecode or pickled module is written anywhere.

Fixture kinds:
  tiny_clip.npz            frozen weights of the tiny arch (shared by all tiny_* cases)
  tiny_<case>.npz          inputs, prompt-learner state, token ids, layout -> logits/loss/grads
  full_<case>.npz          ViT-B/32 / ViT-B/16 output-only (weights regenerated from the seed)
  tokens.npz               clip.tokenize ids / name_lens / EOT positions (bit-exact integers)
  full_vitb32_coop_trainer.npz   (`coop`) the same ViT-B/32 CoOp case through trainers/coop.py's own classes
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from mvlpt_amd.weights import ARCHS, make_state_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

CLASSNAMES = ["dog", "grand piano", "sea horse", "airplane", "great white shark", "cat",
              "golden retriever", "mountain bike", "hot air balloon", "tree frog", "bus", "water lily"]

TINY_SEED, FULL_SEED = 1, 2


def build_ref_clip(cm, arch, seed):
    sd = make_state_dict(arch, seed, include_token_embedding=True)
    model = cm.CLIP(*arch.ctor_args())
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.float().eval(), sd


def layout_from_reference(pl):
    """Recover forward_coop's row layout by pushing index markers through the reference."""
    C = pl.n_cls
    pre, suf = pl.token_prefix.clone(), pl.token_suffix.clone()
    dt = pre.shape[-1]
    try:
        pl.token_prefix.copy_(torch.zeros_like(pre))
        pl.token_suffix.copy_((torch.arange(suf.shape[1]).float() + 1).view(1, -1, 1).expand_as(suf))
        if pl.ctx is None:
            marker = None
        else:
            n = pl.ctx.shape[-2]
            marker = (-(torch.arange(n).float() + 1)).view(n, 1).expand(n, dt)
            if pl.ctx.dim() == 3:
                marker = marker.unsqueeze(0).expand(C, n, dt)
        with torch.no_grad():
            out = pl.forward_coop(marker)
    finally:
        pl.token_prefix.copy_(pre)
        pl.token_suffix.copy_(suf)
    return out[..., 0].round().to(torch.int32)


def run_case(mv, clip_model, *, name, image_size, classnames, B, case_seed, soft_labels=False,
             task_counts=None, store_inputs=True, **cfgkw):
    cfg = ref_shim.make_cfg(input_size=image_size, label_pertask=task_counts is not None, **cfgkw)
    dm = ref_shim.make_dm(task_counts) if task_counts is not None else None
    torch.manual_seed(case_seed)
    cc = mv.CustomCLIP(cfg, classnames, clip_model, dm=dm)
    for n_, p in cc.named_parameters():
        p.requires_grad_("prompt_learner" in n_)           # trainers/mvlpt.py:855-858
    pl = cc.prompt_learner
    # nn.Linear / MHA defaults leave some projection biases at 0: make every trainable tensor non-trivial
    g = torch.Generator().manual_seed(case_seed + 77)
    with torch.no_grad():
        for n_, p in pl.named_parameters():
            if n_.startswith("mvlpt_proj") and p.dim() == 1 and float(p.abs().sum()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    # VPT.DROPOUT > 0: record what `vpt_dropout` (trainers/mvlpt.py:165) does to the prompt rows of every prompted layer — the
    # masks are part of the fixture (the HIP path is given the same ones; a mask is data, 0 or 1 / (1 - p))
    drop_masks = []
    if float(cfgkw.get("vpt_dropout", 0.0)) > 0.0:
        p_drop = float(cfgkw["vpt_dropout"])

        class _RecordingDropout(torch.nn.Module):
            def forward(self, x):
                if not self.training:
                    return x
                m = F.dropout(torch.ones_like(x), p_drop, training=True)
                drop_masks.append(m.detach().clone())
                return x * m
        pl.vpt_dropout = _RecordingDropout()
    C = len(classnames)
    g = torch.Generator().manual_seed(case_seed + 1000)    # inputs have their own stream
    image = torch.randn(B, 3, image_size, image_size, generator=g)
    task = None
    if task_counts is not None:
        task = torch.randint(0, len(task_counts), (B,), generator=g)
        starts = np.concatenate([[0], np.cumsum(task_counts)[:-1]])
        label = torch.tensor([int(starts[t] + torch.randint(0, task_counts[t], (1,), generator=g)) for t in task.tolist()])
        if soft_labels:
            # ELEVATER multitask batches: multi-hot float targets inside the sample's own task range
            # (trainers/mvlpt.py:914-916 normalises them; :573-581 masks the logits to the same range)
            hot = torch.zeros(B, C)
            for b, t in enumerate(task.tolist()):
                lo, n = int(starts[t]), int(task_counts[t])
                hot[b, lo:lo + n] = (torch.rand(n, generator=g) > 0.5).float()
                hot[b, label[b]] = 1.0
            label = hot
    elif soft_labels:
        label = (torch.rand(B, C, generator=g) > 0.6).float()
        label[torch.arange(B), torch.randint(0, C, (B,), generator=g)] = 1.0
    else:
        label = torch.randint(0, C, (B,), generator=g)
    lab = label
    if soft_labels:                                        # trainers/mvlpt.py:914-916
        lab = label.float()
        lab = lab / lab.sum(dim=-1, keepdim=True)
    logits = cc(image, task=task)
    loss = F.cross_entropy(logits, lab)
    loss.backward()

    if drop_masks:
        cc.eval()                                          # the stored tower features are the evaluation-mode ones
    with torch.no_grad():
        coop_emb, vpt_emb, vpt_deep_emb = pl.forward_mvlpt_proj(cc.dtype)
        img_feat = cc.image_encoder(image, vpt_emb, vpt_deep_emb)
        txt_feat = cc.text_encoder(pl.forward_coop(coop_emb), cc.tokenized_prompts)

    d = {
        "meta_coop_n_ctx": np.int64(pl.coop_n_ctx), "meta_vpt_n_ctx": np.int64(pl.vpt_n_ctx),
        "meta_vpt_deep": np.int64(bool(pl.vpt_deep) and pl.vpt_embeddings_deep is not None),
        "meta_position": np.array(pl.class_token_position),
        "meta_cut": np.int64(bool(cfgkw.get("cut_contextlen", False))),
        "meta_csc": np.int64(bool(cfgkw.get("csc", False))),
        "tokenized_prompts": pl.tokenized_prompts.numpy().astype(np.int64),
        "name_lens": np.array(pl.name_lens, dtype=np.int64),
        "eot": pl.tokenized_prompts.argmax(dim=-1).numpy().astype(np.int64),
        "layout": layout_from_reference(pl).numpy(),
        "label": label.numpy(),
        "out_logits": logits.detach().numpy(), "out_loss": loss.detach().numpy(),
        "out_image_features": img_feat.numpy(), "out_text_features": txt_feat.numpy(),
        "case_seed": np.int64(case_seed),
    }
    if drop_masks:
        d["vpt_dropout_masks"] = torch.stack(drop_masks).numpy()      # [n_layers, B, n_vpt, width]
        d["meta_vpt_dropout"] = np.float64(cfgkw["vpt_dropout"])
    if int(cfgkw.get("vpt_project", -1)) > -1:
        d["meta_vpt_project"] = np.int64(cfgkw["vpt_project"])
    if cfgkw.get("coop_ctx_init"):
        d["meta_coop_ctx_init"] = np.array(cfgkw["coop_ctx_init"])
        d["meta_coop_n_ctx_cfg"] = np.int64(cfgkw["coop_n_ctx"])      # what the config asked for (the words decide: trainers/mvlpt.py:207)
    if task is not None:
        d["task"] = task.numpy().astype(np.int64)
        d["task_start"] = cc.class_index_pertask_start.numpy().astype(np.int64)
        d["task_end"] = cc.class_index_pertask_end.numpy().astype(np.int64)
    for n_, p in pl.named_parameters():
        d["param_" + n_] = p.detach().numpy()
        d["grad_" + n_] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    if store_inputs:
        d["image"] = image.numpy()
        d["token_prefix"] = pl.token_prefix.numpy()
        d["token_suffix"] = pl.token_suffix.numpy()
    else:
        d["image_seed"] = np.int64(case_seed + 1000)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(f"[golden] {name}: logits {tuple(logits.shape)} loss {float(loss):.6f} "
          f"params {[n_ for n_, _ in pl.named_parameters()][:4]}…")


def make_tiny(mv, cm):
    arch = ARCHS["tiny"]
    clip_model, sd = build_ref_clip(cm, arch, TINY_SEED)
    np.savez_compressed(os.path.join(OUT, "tiny_clip.npz"),
                        **{k: v.numpy() for k, v in sd.items() if k != "token_embedding.weight"})
    names5 = CLASSNAMES[:5]
    common = dict(image_size=arch.image_resolution, classnames=names5, B=4)
    run_case(mv, clip_model, name="tiny_coop_end", case_seed=11, coop_n_ctx=4, class_token_position="end", **common)
    run_case(mv, clip_model, name="tiny_coop_middle", case_seed=12, coop_n_ctx=4, class_token_position="middle", **common)
    run_case(mv, clip_model, name="tiny_coop_front", case_seed=13, coop_n_ctx=4, class_token_position="front", **common)
    run_case(mv, clip_model, name="tiny_coop_csc", case_seed=14, coop_n_ctx=3, csc=True, class_token_position="middle", **common)
    run_case(mv, clip_model, name="tiny_vpt_shallow", case_seed=16, vpt_n_ctx=2, vpt_deep=False, **common)
    run_case(mv, clip_model, name="tiny_vpt_deep", case_seed=17, vpt_n_ctx=2, vpt_deep=True, **common)
    run_case(mv, clip_model, name="tiny_upt", case_seed=18, coop_n_ctx=4, vpt_n_ctx=2, vpt_deep=True, project_dim=64, **common)
    run_case(mv, clip_model, name="tiny_upt_samedim", case_seed=19, coop_n_ctx=2, vpt_n_ctx=3, vpt_deep=True, project_dim=128, **common)
    run_case(mv, clip_model, name="tiny_task_mask", case_seed=20, coop_n_ctx=4, class_token_position="middle",
             task_counts=[2, 1, 2], **common)
    run_case(mv, clip_model, name="tiny_soft_labels", case_seed=21, coop_n_ctx=4, class_token_position="end",
             soft_labels=True, **common)
    # the ELEVATER-shaped combination (BASELINE cfg4/cfg5): UPT + per-task logit mask + soft multi-hot labels together
    run_case(mv, clip_model, name="tiny_upt_mask_soft", case_seed=23, coop_n_ctx=4, vpt_n_ctx=2, vpt_deep=True, project_dim=64,
             task_counts=[2, 1, 2], soft_labels=True, **common)
    make_train_steps(mv, clip_model, arch)
    # CUT_CONTEXTLEN slices the shared causal masks in place (trainers/mvlpt.py:115-117): run LAST
    run_case(mv, clip_model, name="tiny_coop_cut", case_seed=15, coop_n_ctx=4, class_token_position="middle",
             cut_contextlen=True, **common)
    run_case(mv, clip_model, name="tiny_upt_cut", case_seed=22, coop_n_ctx=4, vpt_n_ctx=2, vpt_deep=True,
             project_dim=64, cut_contextlen=True, **common)


def make_vpt_options(mv, cm):
    """`vptopt`: TRAINER.MVLPT.VPT.PROJECT > -1 (a trainable Linear in front of the visual prompts, trainers/mvlpt.py:170-175) and
    VPT.DROPOUT > 0 (:165, per-image masks).  Separate target: the other tiny fixtures are not regenerated."""
    arch = ARCHS["tiny"]
    clip_model, _ = build_ref_clip(cm, arch, TINY_SEED)
    common = dict(image_size=arch.image_resolution, classnames=CLASSNAMES[:5], B=4)
    run_case(mv, clip_model, name="tiny_vpt_project", case_seed=41, vpt_n_ctx=2, vpt_deep=True, vpt_project=24, **common)
    run_case(mv, clip_model, name="tiny_vpt_project_dropout", case_seed=42, vpt_n_ctx=3, vpt_deep=True, vpt_project=24,
             vpt_dropout=0.25, **common)
    run_case(mv, clip_model, name="tiny_vpt_shallow_dropout", case_seed=43, vpt_n_ctx=2, vpt_deep=False, vpt_dropout=0.5, **common)


def make_ctx_init(mv, cm):
    """`ctxinit`: TRAINER.MVLPT.COOP.CTX_INIT (trainers/mvlpt.py:203-212; the reference's *_ctxv1.yaml configs): the context vectors
    start as the token embeddings of the given words and their count overrides COOP.N_CTX.  The fixture stores the initial `ctx`
    (param_ctx: nothing touches it before the forward) next to the usual outputs.  VPT.CTX_INIT raises in the reference (:180-182)."""
    arch = ARCHS["tiny"]
    clip_model, _ = build_ref_clip(cm, arch, TINY_SEED)
    common = dict(image_size=arch.image_resolution, classnames=CLASSNAMES[:5], B=4)
    run_case(mv, clip_model, name="tiny_coop_ctxinit", case_seed=44, coop_n_ctx=16, coop_ctx_init="a_photo_of_a", class_token_position="end", **common)
    run_case(mv, clip_model, name="tiny_upt_ctxinit", case_seed=45, coop_n_ctx=2, coop_ctx_init="a photo of", vpt_n_ctx=2, vpt_deep=True,
             project_dim=64, **common)
    try:
        mv.CustomCLIP(ref_shim.make_cfg(input_size=arch.image_resolution, vpt_n_ctx=2, vpt_ctx_init="a photo"), CLASSNAMES[:5], clip_model, dm=None)
        raise AssertionError("the reference accepted VPT.CTX_INIT")
    except ValueError as e:
        print(f"[golden] VPT.CTX_INIT -> ValueError({e})")


def make_train_steps(mv, clip_model, arch):
    """Train-step fixture: three `forward_backward` steps of the REFERENCE model — forward,
    F.cross_entropy, backward (trainers/mvlpt.py:927-932) — with the optimizer Dassl's build_optimizer("sgd") makes
    (torch.optim.SGD, momentum 0.9, weight_decay 5e-4, dampening 0, no nesterov — recalled Dassl defaults, SURVEY Appendix
    B) over prompt_learner.parameters() (:869), one step per "epoch" so that the learning rate goes through the constant
    warm-up -> base LR -> first cosine step (recalled Dassl ConstantWarmupScheduler + CosineAnnealingLR; configs/trainers/
    MVLPT/vit_b16.yaml:15-22 with MAX_EPOCH = 3).  Stores inputs, initial and final parameters, losses."""
    import math
    cfg = ref_shim.make_cfg(input_size=arch.image_resolution, coop_n_ctx=4, vpt_n_ctx=2, vpt_deep=True, project_dim=64)
    torch.manual_seed(51)
    cc = mv.CustomCLIP(cfg, CLASSNAMES[:5], clip_model, dm=None)
    for n_, p in cc.named_parameters():
        p.requires_grad_("prompt_learner" in n_)
    pl = cc.prompt_learner
    base_lr, max_epoch, cons = 0.002, 3, 1e-5
    lrs = [cons, base_lr, 0.5 * base_lr * (1 + math.cos(math.pi * 1 / max_epoch))]
    opt = torch.optim.SGD(pl.parameters(), lr=base_lr, momentum=0.9, weight_decay=5e-4, dampening=0, nesterov=False)
    g = torch.Generator().manual_seed(5100)
    B = 4
    d = {"lrs": np.array(lrs), "base_lr": np.float64(base_lr), "max_epoch": np.int64(max_epoch),
         "tokenized_prompts": pl.tokenized_prompts.numpy().astype(np.int64), "name_lens": np.array(pl.name_lens, dtype=np.int64),
         "token_prefix": pl.token_prefix.numpy(), "token_suffix": pl.token_suffix.numpy()}
    for n_, p in pl.named_parameters():
        d["init_" + n_] = p.detach().numpy().copy()
    images, labels, losses = [], [], []
    for step in range(3):
        image = torch.randn(B, 3, arch.image_resolution, arch.image_resolution, generator=g)
        label = torch.randint(0, 5, (B,), generator=g)
        for grp in opt.param_groups:
            grp["lr"] = lrs[step]
        loss = F.cross_entropy(cc(image, task=None), label)
        opt.zero_grad()
        loss.backward()
        opt.step()
        images.append(image.numpy()); labels.append(label.numpy()); losses.append(float(loss))
    d["images"], d["labels"], d["losses"] = np.stack(images), np.stack(labels), np.array(losses)
    for n_, p in pl.named_parameters():
        d["final_" + n_] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "tiny_train_steps.npz"), **d)
    print(f"[golden] tiny_train_steps: losses {losses} lrs {lrs}")


def make_full(mv, cm):
    for arch_name, tag in (("ViT-B/32", "vitb32"), ("ViT-B/16", "vitb16")):
        arch = ARCHS[arch_name]
        clip_model, _ = build_ref_clip(cm, arch, FULL_SEED)
        common = dict(image_size=224, classnames=CLASSNAMES, B=4, store_inputs=False)
        if tag == "vitb32":   # BASELINE cfg1 shape family: CoOp-16 `end`, L=77
            run_case(mv, clip_model, name="full_vitb32_coop_end", case_seed=31, coop_n_ctx=16,
                     class_token_position="end", **common)
        else:
            run_case(mv, clip_model, name="full_vitb16_vpt_deep", case_seed=33, vpt_n_ctx=8, vpt_deep=True, **common)
            run_case(mv, clip_model, name="full_vitb16_coop_middle", case_seed=32, coop_n_ctx=16,
                     class_token_position="middle", **common)
            run_case(mv, clip_model, name="full_vitb16_upt_cut", case_seed=34, coop_n_ctx=4, vpt_n_ctx=4,
                     vpt_deep=True, cut_contextlen=True, **common)


def run_coop_trainer_case(cm, golden_name="full_vitb32_coop_end", arch_name="ViT-B/32"):
    """BASELINE configs[0] is `--trainer CoOp`, i.e. trainers/coop.py — its OWN PromptLearner / TextEncoder / CustomCLIP
    (trainers/coop.py:45-80, 83-212, 215-260), not the MVLPT trainer with VPT.N_CTX = 0 that `full_vitb32_coop_end.npz` was
    generated through.  Instantiate that class on the same frozen weights, give it the fixture's context vectors (the two
    constructors draw their random initialisations in a different order), run forward + F.cross_entropy + backward on the
    fixture's inputs and return what it computes."""
    import importlib
    ref_shim.install()
    coop = importlib.import_module("trainers.coop")
    arch = ARCHS[arch_name]
    clip_model, _ = build_ref_clip(cm, arch, FULL_SEED)
    z = np.load(os.path.join(OUT, golden_name + ".npz"))
    cfg = ref_shim.make_cfg(input_size=arch.image_resolution, coop_n_ctx=int(z["meta_coop_n_ctx"]),
                            class_token_position=str(z["meta_position"]))
    cfg.TRAINER.COOP = cfg.TRAINER.MVLPT.COOP           # trainers/coop.py reads TRAINER.COOP.* (train.py:105-112)
    cc = coop.CustomCLIP(cfg, CLASSNAMES, clip_model)
    for n_, p in cc.named_parameters():
        p.requires_grad_("prompt_learner" in n_)           # trainers/coop.py:300-302
    pl = cc.prompt_learner
    with torch.no_grad():
        pl.ctx.copy_(torch.from_numpy(z["param_ctx"]))
    assert np.array_equal(pl.tokenized_prompts.numpy(), z["tokenized_prompts"])
    g = torch.Generator().manual_seed(int(z["image_seed"]))        # run_case's input stream
    image = torch.randn(len(z["label"]), 3, arch.image_resolution, arch.image_resolution, generator=g)
    label = torch.from_numpy(z["label"])
    logits = cc(image)
    loss = F.cross_entropy(logits, label)
    loss.backward()
    return {"out_logits": logits.detach().numpy(), "out_loss": loss.detach().numpy(), "grad_ctx": pl.ctx.grad.numpy(),
            "param_ctx": z["param_ctx"], "label": z["label"], "image_seed": z["image_seed"]}


def make_coop_trainer(cm):
    d = run_coop_trainer_case(cm)
    np.savez_compressed(os.path.join(OUT, "full_vitb32_coop_trainer.npz"), **d)
    print(f"[golden] full_vitb32_coop_trainer (trainers/coop.py): loss {float(d['out_loss']):.6f}")


def make_large(mv, cm):
    """BASELINE cfg5 family: ViT-L/14@336px (581 vision tokens with 4 prompts, 24 layers, width 1024; text width 768)."""
    arch = ARCHS["ViT-L/14@336px"]
    clip_model, _ = build_ref_clip(cm, arch, FULL_SEED)
    run_case(mv, clip_model, name="full_vitl14_336_upt_cut", case_seed=41, coop_n_ctx=4, vpt_n_ctx=4, vpt_deep=True,
             cut_contextlen=True, image_size=336, classnames=CLASSNAMES[:6], B=2, store_inputs=False)


def make_tokens(mv):
    from clip import clip as refclip  # noqa
    tok = mv._tokenizer
    names = CLASSNAMES
    out = {"names": np.array(names)}
    for n_ctx in (0, 4, 16):
        prefix = " ".join(["X"] * n_ctx) if n_ctx else "a photo of a "      # trainers/mvlpt.py:201,227
        prompts = [prefix + " " + n + "." for n in names]                   # :295
        ids = torch.cat([refclip.tokenize(p) for p in prompts]).numpy().astype(np.int64)
        out[f"ids_nctx{n_ctx}"] = ids
        out[f"eot_nctx{n_ctx}"] = ids.argmax(-1).astype(np.int64)
        out[f"cutlen_nctx{n_ctx}"] = np.int64(max(len(tok.encode(p)) + 2 for p in prompts))  # :297-300
    out["name_lens"] = np.array([len(tok.encode(n)) for n in names], dtype=np.int64)      # :293
    np.savez_compressed(os.path.join(OUT, "tokens.npz"), **out)
    print("[golden] tokens: name_lens", out["name_lens"].tolist())


def make_metrics():
    """ELEVATER metrics of the reference (trainers/vision_benchmark/datasets/metrics.py:1254-1294; needs sklearn) on
    seeded score matrices: continuous scores, heavily tied scores, a class that never occurs, multi-label targets."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_elevater_metrics", os.path.join(ref_shim.REFERENCE_ROOT, "trainers", "vision_benchmark", "datasets", "metrics.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(11)
    out = {}
    cases = []
    for i, (N, C, tied, skip, multilabel) in enumerate([(64, 7, False, False, False), (200, 10, True, False, False),
                                                        (90, 12, False, True, False), (150, 5, False, False, True),
                                                        (33, 2, True, False, False), (120, 20, True, True, True)]):
        score = rng.normal(size=(N, C))
        if tied:
            score = np.round(score * 2) / 2
        classes = np.arange(C) if not skip else np.delete(np.arange(C), [1, C - 1])
        y = rng.choice(classes, size=N)
        onehot = np.eye(C, dtype=np.int64)[y]
        if multilabel:
            extra = (rng.random(size=(N, C)) < 0.15).astype(np.int64)
            if skip:
                extra[:, [1, C - 1]] = 0
            onehot = np.maximum(onehot, extra)
        score = score + 1.5 * onehot * rng.random(size=(N, 1))          # informative scores
        out[f"c{i}_score"], out[f"c{i}_y"], out[f"c{i}_onehot"] = score, y.astype(np.int64), onehot
        out[f"c{i}_accuracy"] = np.float64(m.accuracy(y, score))                           # trainers/mvlpt.py:1062-1064
        out[f"c{i}_mean_per_class"] = np.float64(m.balanced_accuracy_score(onehot, score))
        out[f"c{i}_map11"] = np.float64(m.map_11_points(onehot, score))
        if not skip:                                                    # sklearn raises on a single-valued column
            out[f"c{i}_roc_auc"] = np.float64(m.roc_auc(onehot, score))
        cases.append(i)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("[golden] metrics:", {k: float(v) for k, v in out.items() if k.endswith(("accuracy", "class", "map11", "auc"))})


def make_preprocess():
    """Input pipeline fixtures: Pillow's own `crop` + `resize(BICUBIC)` (the arithmetic behind torchvision's
    RandomResizedCrop / Resize on PIL images: configs/trainers/MVLPT/vit_b16.yaml:8-13, feature.py:538-553) followed by
    ToTensor (u8 -> fp32 / 255) and Normalize with the CLIP mean/std, computed with torch CPU ops.  Data only."""
    from PIL import Image
    rng = np.random.default_rng(5)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073])           # vit_b16.yaml:11-12
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711])
    out = {"mean": mean.numpy(), "std": std.numpy()}
    # (H, W, crop(top,left,h,w), resized (rh,rw), window (top,left,h,w), flip, content)
    specs = [
        (60, 80, (7, 5, 40, 50), (32, 32), (0, 0, 32, 32), 0, "noise"),        # downscale both ways
        (60, 80, (0, 0, 60, 80), (32, 32), (0, 0, 32, 32), 1, "smooth"),       # whole image, flipped
        (20, 24, (2, 3, 11, 13), (32, 32), (0, 0, 32, 32), 0, "noise"),        # upscale
        (48, 64, (8, 16, 32, 40), (32, 32), (0, 0, 32, 32), 1, "binary"),      # height unchanged: vertical pass skipped
        (64, 48, (10, 4, 50, 32), (32, 32), (0, 0, 32, 32), 0, "smooth"),      # width unchanged: horizontal pass skipped
        (40, 40, (4, 4, 32, 32), (32, 32), (0, 0, 32, 32), 1, "noise"),        # no resize at all
        (96, 130, (0, 0, 96, 130), (36, 48), (2, 8, 32, 32), 0, "smooth"),     # Resize(36) + CenterCrop(32) (eval path)
        (33, 200, (0, 0, 33, 200), (32, 193), (0, 80, 32, 32), 0, "noise"),    # extreme aspect, centre window
        (7, 9, (1, 1, 5, 7), (32, 32), (0, 0, 32, 32), 0, "binary"),           # tiny source, clamped taps
        (300, 400, (37, 91, 213, 251), (224, 224), (0, 0, 224, 224), 1, "smooth"),   # full-size output
        (1, 50, (0, 3, 1, 40), (32, 32), (0, 0, 32, 32), 0, "noise"),          # one-row source
    ]
    for i, (H, W, (ct, cl, ch, cw), (rh, rw), (ot, ol, oh, ow), flip, kind) in enumerate(specs):
        if kind == "noise":
            a = rng.integers(0, 256, (H, W, 3))
        elif kind == "binary":
            a = rng.integers(0, 2, (H, W, 3)) * 255
        else:
            yy, xx = np.mgrid[0:H, 0:W]
            a = np.stack([127 + 120 * np.sin(yy / 7.0 + xx / 11.0), (xx * 255) // max(W - 1, 1),
                          127 + 100 * np.cos(yy / 3.0) * np.sin(xx / 5.0)], -1) + rng.integers(-6, 7, (H, W, 3))
        a = np.clip(a, 0, 255).astype(np.uint8)
        img = Image.fromarray(a).crop((cl, ct, cl + cw, ct + ch)).resize((rw, rh), Image.BICUBIC)     # PIL: (width, height)
        img = img.crop((ol, ot, ol + ow, ot + oh))
        if flip:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        u8 = np.asarray(img).copy()
        tens = torch.from_numpy(u8).permute(2, 0, 1).contiguous().to(torch.float32).div(255)           # ToTensor
        tens = (tens - mean[:, None, None]) / std[:, None, None]                                        # Normalize
        out[f"c{i}_src"] = a
        out[f"c{i}_desc"] = np.array([ct, cl, ch, cw, rh, rw, ot, ol, oh, ow, flip], dtype=np.int64)
        out[f"c{i}_u8"] = u8
        out[f"c{i}_f32"] = tens.numpy()
    out["n"] = np.int64(len(specs))
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **out)
    print("[golden] preprocess:", len(specs), "cases,", os.path.getsize(os.path.join(OUT, "preprocess.npz")) // 1024, "KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    if set(sys.argv[1:]) == {"preprocess"}:
        return make_preprocess()
    which = set(sys.argv[1:]) or {"tiny", "full", "tokens", "metrics"}
    if which == {"metrics"}:
        return make_metrics()
    if "metrics" in which:
        make_metrics()
    if "preprocess" in which or not sys.argv[1:]:
        make_preprocess()
    torch.set_num_threads(8)
    mv, cm = ref_shim.load_reference()
    if "tokens" in which:
        make_tokens(mv)
    if "tiny" in which:
        make_tiny(mv, cm)
    if "full" in which:
        make_full(mv, cm)
    if "large" in which:
        make_large(mv, cm)
    if "coop" in which:
        make_coop_trainer(cm)
    if "vptopt" in which:
        make_vpt_options(mv, cm)
    if "ctxinit" in which:
        make_ctx_init(mv, cm)


if __name__ == "__main__":
    main()
