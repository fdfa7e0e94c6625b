"""Shared by the CPU (oracle engine) and GPU (HIP engine) train-step tests: drive the trainer stand-in through the
three steps of tests/golden/tiny_train_steps.npz (reference forward/backward + torch SGD, one step per epoch so the
LR goes warm-up -> base -> first cosine step) and return what the fixture pins."""
import torch

from tests.golden_util import load_npz, t


class EpochLoader:
    """train_loader_x of ONE batch per epoch: batch e in epoch e (the trainer calls update_lr at the last batch)."""

    def __init__(self, batches):
        self.batches, self.epoch = batches, 0

    def __len__(self):
        return 1

    def __iter__(self):
        yield self.batches[self.epoch]
        self.epoch += 1


def fixture_cfg(z):
    from mvlpt_amd.config import get_cfg_default
    cfg = get_cfg_default()
    cfg.MODEL.BACKBONE.NAME = "tiny"
    cfg.INPUT.SIZE = (32, 32)
    cfg.DATALOADER.TRAIN_X.BATCH_SIZE = 4
    T = cfg.TRAINER.MVLPT
    T.COOP.N_CTX, T.VPT.N_CTX, T.VPT.DEEP, T.PROJECT_DIM = 4, 2, True, 64
    cfg.OPTIM.LR, cfg.OPTIM.MAX_EPOCH = float(z["base_lr"]), int(z["max_epoch"])      # vit_b16.yaml:15-22 otherwise
    cfg.TRAIN.PRINT_FREQ = 10 ** 9
    return cfg


def run_three_steps(trainer, z, device):
    """`trainer`: a constructed MVLPT whose prompt_learner has the fixture's class tokens; returns (losses, lrs, params)."""
    pl = trainer.model.prompt_learner
    sd = {k[len("init_"):]: t(v) for k, v in z.items() if k.startswith("init_")}
    sd["token_prefix"], sd["token_suffix"] = t(z["token_prefix"]), t(z["token_suffix"])
    pl.load_state_dict(sd, strict=True)
    batches = [{"img": t(z["images"][i]).to(device), "label": t(z["labels"][i]).to(device), "domain": torch.zeros(4, dtype=torch.long)}
               for i in range(3)]
    trainer.train_loader_x = EpochLoader(batches)
    losses, lrs = [], []
    for trainer.epoch in range(3):
        lrs.append(trainer.optim.param_groups[0]["lr"])
        losses.append(float(trainer.run_epoch()["loss"]))
    return losses, lrs, {n: p.detach().cpu() for n, p in pl.named_parameters()}


def check_against_fixture(z, losses, lrs, params, loss_tol, delta_tol):
    import numpy as np
    assert np.allclose(lrs, z["lrs"], rtol=1e-12, atol=0), (lrs, z["lrs"].tolist())
    assert np.allclose(losses, z["losses"], atol=loss_tol), (losses, z["losses"].tolist())
    for n, p in params.items():
        init, final = t(z["init_" + n]), t(z["final_" + n])
        want = final - init                                   # what three SGD steps did to this tensor
        got = p - init
        scale = float(want.abs().max()) + 1e-20
        err = float((got - want).abs().max()) / scale
        assert err < delta_tol, f"{n}: parameter update differs from the reference by {err:.2e} of its max"
