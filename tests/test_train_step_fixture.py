"""f2 pins on the CPU (no GPU): the trainer stand-in (SGD + warm-up/cosine schedule + the reference's step order) around
the ORACLE engine reproduces the reference's three-step train fixture; checkpoints written with the reference's key set
load through MVLPT.load_model; ours load into the real reference prompt learner (build container only)."""
import os

import pytest
import torch

from tests.golden_util import load_npz, t
from tests.train_step_util import check_against_fixture, fixture_cfg, run_three_steps

NAMES5 = ["c0", "c1", "c2", "c3", "c4"]


def _oracle_trainer(z):
    """MVLPT with the CPU-oracle engine behind the same host code (tests/fake_engine.py), built without a GPU."""
    from mvlpt_amd.model import CustomCLIP, PretokenizedPrompts
    from mvlpt_amd.trainer import MVLPT, build_lr_scheduler, build_optimizer
    from mvlpt_amd.weights import ARCHS
    from tests.fake_engine import OracleFrozenCLIP
    from tests.golden_util import tiny_state_dict
    cfg = fixture_cfg(z)
    tr = MVLPT.__new__(MVLPT)
    from collections import OrderedDict
    tr._models, tr._optims, tr._scheds = OrderedDict(), OrderedDict(), OrderedDict()
    tr.cfg, tr.rank, tr.world_size, tr.local_rank, tr.device = cfg, 0, 1, 0, torch.device("cpu")
    tr.epoch, tr.max_epoch, tr.batch_idx, tr.num_batches, tr.next_batch, tr.batch_hook = 0, 3, 0, 0, None, None
    tr.multi_task = False
    pre = PretokenizedPrompts(t(z["tokenized_prompts"]), z["name_lens"].tolist())
    tr.model = CustomCLIP(cfg, NAMES5, OracleFrozenCLIP(tiny_state_dict(), ARCHS["tiny"]), pretokenized=pre)
    tr.optim = build_optimizer(tr.model.prompt_learner, cfg.OPTIM)
    tr.sched = build_lr_scheduler(tr.optim, cfg.OPTIM)
    tr.register_model("prompt_learner", tr.model.prompt_learner, tr.optim, tr.sched)
    return tr


def test_three_reference_train_steps_on_the_oracle_engine():
    z = load_npz("tiny_train_steps")
    tr = _oracle_trainer(z)
    losses, lrs, params = run_three_steps(tr, z, "cpu")
    check_against_fixture(z, losses, lrs, params, loss_tol=2e-5, delta_tol=5e-4)     # fp32 on both sides (the bound of tests/test_oracle_golden.py: reduction order varies with the thread count)


def _reference_style_checkpoint(tmp_path, z, rename_to_upt=False):
    sd = {k[len("final_"):]: t(v) for k, v in z.items() if k.startswith("final_")}
    sd["token_prefix"], sd["token_suffix"] = t(z["token_prefix"]), t(z["token_suffix"])     # Dassl saves the buffers too
    if rename_to_upt:
        sd = {k.replace("mvlpt_proj", "upt_proj"): v for k, v in sd.items()}                # checkpoints of the older code base
    d = tmp_path / "prompt_learner"
    d.mkdir(parents=True, exist_ok=True)
    torch.save({"state_dict": sd, "epoch": 3, "optimizer": None, "scheduler": None, "val_result": 12.5}, d / "model-best.pth.tar")
    return sd


@pytest.mark.parametrize("rename", [False, True])
def test_reference_checkpoint_loads_through_load_model(tmp_path, rename):
    """trainers/mvlpt.py:1090-1125: model-best.pth.tar, `upt_proj` -> `mvlpt_proj`, token_prefix / token_suffix dropped
    (they belong to THIS model's class list), strict=False."""
    z = load_npz("tiny_train_steps")
    tr = _oracle_trainer(z)
    pl = tr.model.prompt_learner
    with torch.no_grad():
        pl.token_suffix.add_(1.0)                                  # this model's own class tokens differ from the checkpoint's
    keep = pl.token_suffix.clone()
    sd = _reference_style_checkpoint(tmp_path, z, rename_to_upt=rename)
    tr.load_model(str(tmp_path))
    for n, p in pl.named_parameters():
        assert torch.equal(p.detach(), t(z["final_" + n])), n
    assert torch.equal(pl.token_suffix, keep)
    with pytest.raises(FileNotFoundError):
        tr.load_model(str(tmp_path), epoch=7)                      # :1106-1107


def test_checkpoint_key_set_equals_the_references():
    """The Dassl checkpoint dict our save_model writes carries exactly the prompt-learner keys the reference's does."""
    z = load_npz("tiny_train_steps")
    tr = _oracle_trainer(z)
    ours = set(tr.model.prompt_learner.state_dict())
    ref = {k[len("final_"):] for k in z if k.startswith("final_")} | {"token_prefix", "token_suffix"}
    assert ours == ref, ours ^ ref


@pytest.mark.skipif(not os.path.isfile("/root/reference/trainers/mvlpt.py"), reason="needs the reference tree (build container)")
def test_our_checkpoint_loads_into_the_real_reference(tmp_path):
    """The other direction: a checkpoint written by OUR save_model loads into the reference's prompt learner with its own
    load_model body (drop token_prefix / token_suffix, strict=False: trainers/mvlpt.py:1112-1125)."""
    from oracle import ref_shim
    from mvlpt_amd.weights import ARCHS, make_state_dict
    mv, cm = ref_shim.load_reference()
    z = load_npz("tiny_train_steps")
    tr = _oracle_trainer(z)
    run_three_steps(tr, z, "cpu")
    tr.output_dir = str(tmp_path)
    tr.save_model(2, str(tmp_path), is_best=True)
    ck = torch.load(tmp_path / "prompt_learner" / "model-best.pth.tar", map_location="cpu")
    assert set(ck) == {"state_dict", "epoch", "optimizer", "scheduler", "val_result"}                  # scripts/avg_ckpt.py:21-66
    arch = ARCHS["tiny"]
    ref_clip = cm.CLIP(*arch.ctor_args())
    ref_clip.load_state_dict(make_state_dict(arch, 1, include_token_embedding=True))
    cfg = ref_shim.make_cfg(input_size=32, coop_n_ctx=4, vpt_n_ctx=2, vpt_deep=True, project_dim=64)
    ref_pl = mv.CustomCLIP(cfg, ["dog", "grand piano", "sea horse", "airplane", "great white shark"], ref_clip.float(), dm=None).prompt_learner
    sd = dict(ck["state_dict"])
    sd = {k.replace("upt_proj", "mvlpt_proj"): v for k, v in sd.items()}        # :1112
    sd.pop("token_prefix"), sd.pop("token_suffix")                              # :1115-1121
    res = ref_pl.load_state_dict(sd, strict=False)                              # :1125
    assert not res.unexpected_keys and set(res.missing_keys) <= {"token_prefix", "token_suffix"}
    for n, p in ref_pl.named_parameters():
        assert torch.equal(p.detach(), tr.model.prompt_learner.state_dict()[n]), n


def test_best_val_checkpoint_flow(tmp_path):
    """Dassl's after_epoch / after_train with TEST.FINAL_MODEL = "best_val" (SURVEY Appendix B): validate every epoch, keep the
    best as prompt_learner/model-best.pth.tar — the file `load_model(directory)` opens (trainers/mvlpt.py:1098-1104) —, write
    model.pth.tar-<last epoch>, and run the final test on the best checkpoint."""
    from tests.train_step_util import EpochLoader
    z = load_npz("tiny_train_steps")
    tr = _oracle_trainer(z)
    tr.cfg.TEST.FINAL_MODEL, tr.cfg.TEST.NO_TEST, tr.cfg.DATASET.COOP = "best_val", False, True
    tr.output_dir, tr.start_epoch, tr.dm = str(tmp_path), 0, None
    batches = [{"img": t(z["images"][i]), "label": t(z["labels"][i]), "domain": torch.zeros(4, dtype=torch.long)} for i in range(3)]
    tr.train_loader_x = EpochLoader(batches)
    tr.val_loader, tr.test_loader = batches[:2], batches[2:]
    seen = []
    real_test = tr.test

    def scripted_test(split=None):             # validation results 30, 50, 40: epoch 2 is the best
        acc = real_test(split)
        assert 0.0 <= acc <= 100.0
        if split == "val":
            seen.append([30.0, 50.0, 40.0][len(seen)])
            return seen[-1]
        return acc

    tr.test = scripted_test
    after = []
    real_save = tr.save_model

    def spy_save(epoch, directory, **kw):
        after.append((epoch, kw.get("model_name", ""), {n: p.detach().clone() for n, p in tr.model.prompt_learner.named_parameters()}))
        return real_save(epoch, directory, **kw)

    tr.save_model = spy_save
    tr.train()
    d = tmp_path / "prompt_learner"
    assert (d / "model-best.pth.tar").is_file() and (d / "model.pth.tar-3").is_file() and not (d / "model.pth.tar-1").exists()
    best = torch.load(d / "model-best.pth.tar", map_location="cpu")
    assert best["epoch"] == 2 and best["val_result"] == 50.0 and set(best) == {"state_dict", "epoch", "optimizer", "scheduler", "val_result"}
    assert [(e, n) for e, n, _ in after] == [(0, "model-best.pth.tar"), (1, "model-best.pth.tar"), (2, "")]
    # after_train loaded the best checkpoint (epoch 2's parameters, not the last epoch's) before the final test
    for n, p in tr.model.prompt_learner.named_parameters():
        assert torch.equal(p.detach(), after[1][2][n]), n


def _best_val_rank(rank, world, port, out_dir, ret):
    """One rank of the two-rank best-val flow: the LAST epoch is the best one and rank 0's torch.save is slow, so a rank that
    ran ahead of the checkpoint (no barrier behind save_model) would open a missing or truncated model-best.pth.tar."""
    import time

    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mvlpt_amd import distributed as D
    from tests.train_step_util import EpochLoader
    D.init_process_group("gloo")
    z = load_npz("tiny_train_steps")
    tr = _oracle_trainer(z)
    tr.rank, tr.world_size = rank, world
    tr.cfg.TEST.FINAL_MODEL, tr.cfg.TEST.NO_TEST, tr.cfg.DATASET.COOP = "best_val", False, True
    tr.output_dir, tr.start_epoch, tr.dm = out_dir, 0, None
    batches = [{"img": t(z["images"][i]), "label": t(z["labels"][i]), "domain": torch.zeros(4, dtype=torch.long)} for i in range(3)]
    tr.train_loader_x = EpochLoader(batches)
    tr.val_loader, tr.test_loader = batches[:2], batches[2:]
    seen = []
    real_test, real_save = tr.test, tr.save_model

    def scripted_test(split=None):
        acc = real_test(split)
        if split == "val":
            seen.append([30.0, 40.0, 50.0][len(seen)])
            return seen[-1]
        return acc

    def slow_save(epoch, directory, **kw):
        if rank == 0:
            time.sleep(0.5)
        return real_save(epoch, directory, **kw)

    tr.test, tr.save_model = scripted_test, slow_save
    tr.train()
    mine = torch.cat([p.detach().flatten() for p in tr.model.prompt_learner.parameters()])
    best = torch.load(os.path.join(out_dir, "prompt_learner", "model-best.pth.tar"), map_location="cpu")
    want = torch.cat([best["state_dict"][n].flatten() for n, _ in tr.model.prompt_learner.named_parameters()])
    ret[rank] = bool(best["epoch"] == 3 and torch.equal(mine, want))
    dist.destroy_process_group()


def test_best_val_checkpoint_flow_two_ranks(tmp_path):
    """every rank loads the best checkpoint only after rank 0 has finished writing it."""
    import socket

    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_best_val_rank, args=(2, port, str(tmp_path), ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}
