"""Trainer-level GPU tests: the MVLPT trainer surface (forward_backward / test / save / load_model) on the HIP
engine, a short training run that must reduce the loss, and the inference-time text-feature cache."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def make_trainer(tmp_path, method="coop", classes=6, tasks=None, steps=4, B=8, elevater=False, metric_names=None, pipelining=True):
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.trainer import MVLPT, SyntheticDataManager
    from mvlpt_amd.weights import ARCHS, make_state_dict
    cfg = get_cfg_default()
    cfg.MODEL.BACKBONE.NAME = "tiny"
    cfg.INPUT.SIZE = (32, 32)
    cfg.DATALOADER.TRAIN_X.BATCH_SIZE = B
    cfg.OUTPUT_DIR = str(tmp_path)
    cfg.OPTIM.MAX_EPOCH = 3
    cfg.OPTIM.LR = 0.05
    cfg.OPTIM.WARMUP_EPOCH = 0
    cfg.TRAIN.PRINT_FREQ = 1000
    if method in ("coop", "upt"):
        cfg.TRAINER.MVLPT.COOP.N_CTX = 4
    if method in ("vpt", "upt"):
        cfg.TRAINER.MVLPT.VPT.N_CTX = 2
    cfg.TRAINER.MVLPT.PROJECT_DIM = 64
    if tasks:
        cfg.DATASET.MULTITASK = True
        cfg.DATASET.MULTITASK_LABEL_PERTASK = True
    cfg.DATASET.COOP = not elevater
    cfg.TRAINER.MVLPT.STEP_PIPELINING = pipelining
    dm = SyntheticDataManager(cfg, classes, steps, task_class_counts=tasks, device="cuda", seed=3, elevater=elevater,
                              metric_names=metric_names)
    return MVLPT(cfg, dm=dm, clip_state_dict=make_state_dict(ARCHS["tiny"], seed=9))


@pytest.mark.parametrize("method", ["coop", "vpt", "upt"])
def test_training_reduces_loss_and_checkpoints_roundtrip(tmp_path, method):
    tr = make_trainer(tmp_path, method)
    losses = []
    for ep in range(3):
        tr.epoch = ep
        tr.set_model_mode("train")
        tr.num_batches = len(tr.train_loader_x)
        for tr.batch_idx, batch in enumerate(tr.train_loader_x):
            losses.append(float(tr.forward_backward(batch)["loss"]))
    assert all(l == l for l in losses)
    assert sum(losses[-4:]) < sum(losses[:4]), losses          # same 4 batches revisited: the prompts must fit them
    tr.save_model(2, str(tmp_path))
    ck = os.path.join(str(tmp_path), "prompt_learner", "model.pth.tar-3")
    state = torch.load(ck, map_location="cpu")
    assert set(state) == {"state_dict", "epoch", "optimizer", "scheduler", "val_result"} and state["epoch"] == 3
    assert "token_prefix" in state["state_dict"] and "token_suffix" in state["state_dict"]
    before = {k: v.clone() for k, v in tr.model.prompt_learner.state_dict().items()}
    with torch.no_grad():
        for p in tr.model.prompt_learner.parameters():
            p.add_(1.0)
    tr.load_model(str(tmp_path), epoch=3)                       # drops token_prefix/suffix, strict=False (:1112-1125)
    for k, v in tr.model.prompt_learner.state_dict().items():
        assert torch.equal(v.cpu(), before[k].cpu()), k


def test_eval_accuracy_and_text_cache(tmp_path):
    tr = make_trainer(tmp_path, "coop", classes=6, tasks=[2, 1, 3])
    acc = tr.test()
    assert 0.0 <= acc <= 100.0 and set(tr.last_task_results) <= {"task0", "task1", "task2"}
    model = tr.model
    assert model._eval_text_cache is not None
    ver0 = model._eval_text_cache[0]
    tr.test()
    assert model._eval_text_cache[0] == ver0                    # prompts unchanged -> text tower not re-run
    tr.set_model_mode("train")
    tr.num_batches, tr.batch_idx = 10, 0
    tr.forward_backward(tr.train_loader_x[0])                    # SGD step bumps the parameter versions
    tr.test()
    assert model._eval_text_cache[0] != ver0


def test_elevater_eval_branch_uses_the_task_metrics(tmp_path):
    """ELEVATER data (tuple batches, one-hot targets): per-task metric on the task's column slice, then the average
    (trainers/mvlpt.py:1048-1075), recomputed here from the model's own logits."""
    import numpy as np
    from mvlpt_amd import metrics as M
    names = ["accuracy", "mean-per-class", "11point_mAP"]
    tr = make_trainer(tmp_path, "coop", classes=9, tasks=[3, 2, 4], B=32, elevater=True, metric_names=names)
    tr.test_loader = tr.train_loader_x                         # 4 batches
    got = tr.test()
    assert set(tr.last_task_results) == {"task0", "task1", "task2"}
    preds, trues, tasks = [], [], []
    with torch.no_grad():
        for img, lab, _idx, task in tr.test_loader:
            preds.append(tr.model(img, task=task).float().cpu().numpy())
            trues.append(lab.cpu().numpy())
            tasks.append(task.numpy())
    pred, true, task = np.concatenate(preds), np.concatenate(trues), np.concatenate(tasks)
    want = {}
    for t, (lo, hi) in enumerate([(0, 3), (3, 5), (5, 9)]):
        yt, yp = true[task == t][:, lo:hi], pred[task == t][:, lo:hi]
        if names[t] == "accuracy":
            yt = yt.argmax(-1)
        want[f"task{t}"] = M.get_metric(names[t])(yt, yp)
    for k in want:
        assert abs(want[k] - tr.last_task_results[k]) < 1e-12, k
    assert abs(got - sum(want.values()) / 3) < 1e-12
    # single-task ELEVATER data: the dataset's own metric over all classes (:1076-1080)
    tr1 = make_trainer(tmp_path, "vpt", classes=5, B=16, elevater=True, metric_names="roc_auc")
    auc = tr1.test()
    assert 0.0 <= auc <= 1.0 and list(tr1.last_results) == ["roc_auc"]


def test_step_pipelining_is_transparent(tmp_path):
    """Prefetching the next batch's image features under the backward (TrainerX.run_epoch reads the loader one batch ahead)
    must not change any result, and it must actually happen under the plain Dassl-style loop."""
    runs = []
    for pipe in (False, True):
        torch.manual_seed(0)                      # same prompt initialisation in both runs
        tr = make_trainer(tmp_path, "coop")
        tr.cfg.TRAINER.MVLPT.STEP_PIPELINING = pipe
        tr.cfg.TRAIN.PRINT_FREQ = 10 ** 9
        hits = []
        orig = tr.model.engine.image_fwd
        tr.model.engine.image_fwd = lambda *a, **k: (hits.append(torch.cuda.current_stream().cuda_stream), orig(*a, **k))[1]
        losses = []
        for tr.epoch in range(2):                 # two epochs of 4 batches through the real loop
            out = tr.run_epoch()
            losses.append(float(out["loss"]))
        main = torch.cuda.current_stream().cuda_stream
        if pipe:      # per epoch: the first batch on the main stream, the other three prefetched on the side stream
            assert sum(h != main for h in hits) == 6 and len(hits) == 8, hits
        else:
            assert all(h == main for h in hits) and len(hits) == 8
        runs.append((losses, tr.model.prompt_learner.ctx.detach().cpu().clone()))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])


def test_lookahead_reaches_a_plain_dassl_loop(tmp_path):
    """the look-ahead must not depend on this repo's run_epoch.  A PLAIN loop over the trainer's loader
    — `for batch in loader: forward_backward(batch)`, what Dassl's own run_epoch does — gets the prefetch, because
    MVLPT.build_data_loader wraps train_loader_x in a LookAheadLoader."""
    from mvlpt_amd.trainer import LookAheadLoader
    torch.manual_seed(0)
    tr = make_trainer(tmp_path, "coop")
    assert isinstance(tr.train_loader_x, LookAheadLoader) and len(tr.train_loader_x) == 4
    hits = []
    orig = tr.model.engine.image_fwd
    tr.model.engine.image_fwd = lambda *a, **k: (hits.append(torch.cuda.current_stream().cuda_stream), orig(*a, **k))[1]
    tr.set_model_mode("train")
    tr.num_batches = len(tr.train_loader_x)
    losses = []
    for tr.batch_idx, batch in enumerate(tr.train_loader_x):          # no run_epoch, no next_batch bookkeeping here
        losses.append(float(tr.forward_backward(batch)["loss"]))
    main = torch.cuda.current_stream().cuda_stream
    assert len(hits) == 4 and [h != main for h in hits] == [False, True, True, True], hits
    assert tr.train_loader_x.hits == 3 and tr.next_batch is None
    # same numbers as the un-pipelined trainer
    torch.manual_seed(0)
    tr2 = make_trainer(tmp_path, "coop", pipelining=False)
    assert not isinstance(tr2.train_loader_x, LookAheadLoader)
    tr2.set_model_mode("train")
    tr2.num_batches = len(tr2.train_loader_x)
    l2 = []
    for tr2.batch_idx, batch in enumerate(tr2.train_loader_x):
        l2.append(float(tr2.forward_backward(batch)["loss"]))
    assert losses == l2
    # leaving the loop early drops the pending prefetch instead of leaving it for an unrelated forward
    it = iter(tr.train_loader_x)
    tr.forward_backward(next(it))
    assert len(tr.model._prefetched) == 1
    it.close()
    tr.end_of_epoch_loop()
    assert not tr.model._prefetched and tr._parsed_ahead is None and tr.next_batch is None


@pytest.mark.parametrize("prec", ["fp16", "amp", "fp32"])
def test_three_reference_train_steps_on_the_hip_engine(tmp_path, prec):
    """TRAINER.MVLPT.PREC (trainers/mvlpt.py:835-836, 919-926): `amp` — autocast + GradScaler in the reference — is this engine's default
    mode (fp32 master prompts, 16-bit MFMA inputs, gradient scaling inside the backward; no GradScaler object: `trainer.scaler is
    None`), `fp32` runs every tower on split operands; all three reproduce the reference's fp32 train fixture.
    f2 pin on the GPU: MVLPT (HIP forward/backward + torch SGD + warm-up/cosine schedule, driven through run_epoch)
    reproduces the reference's three-step train fixture: losses within 1e-3, every parameter's UPDATE within 2e-3 of its
    max (the gradients themselves are within 1e-3; momentum carries three of them)."""
    from mvlpt_amd.model import PretokenizedPrompts
    from mvlpt_amd.trainer import MVLPT, SyntheticDataManager
    from tests.golden_util import load_npz, t, tiny_state_dict
    from tests.train_step_util import check_against_fixture, fixture_cfg, run_three_steps
    z = load_npz("tiny_train_steps")
    cfg = fixture_cfg(z)
    cfg.OUTPUT_DIR = str(tmp_path)
    cfg.TRAINER.MVLPT.PREC = prec
    dm = SyntheticDataManager(cfg, 5, 1, device="cuda", seed=3)
    dm.pretokenized = PretokenizedPrompts(t(z["tokenized_prompts"]), z["name_lens"].tolist())
    tr = MVLPT(cfg, dm=dm, clip_state_dict=tiny_state_dict())
    assert tr.scaler is None
    losses, lrs, params = run_three_steps(tr, z, "cuda")
    check_against_fixture(z, losses, lrs, params, loss_tol=1e-3, delta_tol=2e-3)
