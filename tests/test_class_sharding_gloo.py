"""world_size-2 gloo tests (CPU) of the N>1 path of the prompted-CLIP step: data-parallel image slices, the
class-sharded text tower (all-gather of features / reduce-scatter of their gradients) and the flat prompt-gradient
all-reduce must reproduce the single-process gradients on the concatenated batch.  The towers are the CPU oracle
behind the same host code (tests/fake_engine.py); only the distributed logic is under test here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

NAMES = ["dog", "grand piano", "sea horse", "airplane", "great white shark"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(csc, vpt):
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.model import CustomCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    from tests.fake_engine import OracleFrozenCLIP
    arch = ARCHS["tiny"]
    sd = make_state_dict(arch, seed=5)
    cfg = get_cfg_default()
    cfg.INPUT.SIZE = (32, 32)
    cfg.TRAINER.MVLPT.COOP.N_CTX = 4
    cfg.TRAINER.MVLPT.COOP.CSC = csc
    cfg.TRAINER.MVLPT.VPT.N_CTX = 2 if vpt else 0
    cfg.TRAINER.MVLPT.PROJECT_DIM = 64
    torch.manual_seed(7)
    model = CustomCLIP(cfg, NAMES, OracleFrozenCLIP(sd, arch))
    g = torch.Generator().manual_seed(11)
    image = torch.randn(4, 3, 32, 32, generator=g)
    label = torch.randint(0, len(NAMES), (4,), generator=g)
    return model, image, label


def _step(model, image, label):
    for p in model.parameters():
        p.grad = None
    loss = model.cross_entropy(model(image), label)
    loss.backward()
    return {n: p.grad.clone() for n, p in model.prompt_learner.named_parameters()}, float(loss.detach())


def _worker(rank, world, port, csc, vpt, shard, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from mvlpt_amd import distributed as D
    D.init_process_group("gloo")
    model, image, label = _build(csc, vpt)
    if shard:
        model.enable_class_sharding(rank, world)
    per = image.shape[0] // world
    grads, loss = _step(model, image[rank * per:(rank + 1) * per], label[rank * per:(rank + 1) * per])
    for n, p in model.prompt_learner.named_parameters():
        p.grad = grads[n]
    D.all_reduce_gradients(model.prompt_learner.parameters(), world)
    ret[rank] = ({n: p.grad.clone() for n, p in model.prompt_learner.named_parameters()}, loss)
    dist.destroy_process_group()


@pytest.mark.parametrize("csc,vpt,shard", [(False, False, True), (True, False, True), (False, True, True), (False, True, False)])
def test_two_ranks_match_single_process(csc, vpt, shard):
    torch.set_num_threads(4)
    model, image, label = _build(csc, vpt)
    ref, ref_loss = _step(model, image, label)
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), csc, vpt, shard, ret), nprocs=world, join=True)
    assert abs(sum(ret[r][1] for r in range(world)) / world - ref_loss) < 1e-5
    for r in range(world):
        for n, g in ref.items():
            got = ret[r][0][n]
            err = float((got - g).abs().max()) / (float(g.abs().max()) + 1e-20)
            assert err < 2e-4, f"rank {r} {n}: {err}"
