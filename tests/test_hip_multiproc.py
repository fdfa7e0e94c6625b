"""N > 1 path on the REAL HIP engine (-m gpu) without an N-GPU node: two processes share cuda:0
(MVLPT_DEBUG_SHARE_GPU=1: collectives through gloo, every rank its own engine handle / workspaces / streams).
The data-parallel step (image slices + one flat prompt-gradient all-reduce) and the class-sharded text tower
(all-gather of features / reduce-scatter of their gradients) must reproduce the single-process gradients on the
concatenated batch — the GPU twin of tests/test_class_sharding_gloo.py (which runs the same host code on the CPU oracle).
Replaces nn.DataParallel (trainers/mvlpt.py:877-880)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ["dog", "grand piano", "sea horse", "airplane", "great white shark", "cat", "tree frog"]     # 7 classes: uneven shards


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(method, csc, device="cuda:0", names=NAMES):
    from mvlpt_amd.config import get_cfg_default
    from mvlpt_amd.model import CustomCLIP, FrozenCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    arch = ARCHS["tiny"]
    cfg = get_cfg_default()
    cfg.INPUT.SIZE = (32, 32)
    if method in ("coop", "upt"):
        cfg.TRAINER.MVLPT.COOP.N_CTX, cfg.TRAINER.MVLPT.COOP.CSC = 4, csc
    if method in ("vpt", "upt"):
        cfg.TRAINER.MVLPT.VPT.N_CTX = 2
    cfg.TRAINER.MVLPT.PROJECT_DIM = 64
    torch.manual_seed(7)
    model = CustomCLIP(cfg, names, FrozenCLIP(make_state_dict(arch, seed=5), "fp16", device=device)).to(device)
    g = torch.Generator().manual_seed(11)
    image = torch.randn(8, 3, 32, 32, generator=g)
    label = torch.randint(0, len(names), (8,), generator=g)
    return model, image, label


def _step(model, image, label):
    for p in model.parameters():
        p.grad = None
    dev = next(model.parameters()).device
    loss = model.cross_entropy(model(image.to(dev)), label.to(dev))
    loss.backward()
    torch.cuda.synchronize(dev)
    return {n: p.grad.detach().cpu().clone() for n, p in model.prompt_learner.named_parameters()}, float(loss.detach())


def _worker(rank, world, port, method, csc, shard, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MVLPT_DEBUG_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from mvlpt_amd import distributed as D
    r, w, local = D.init_process_group()
    assert (r, w, local) == (rank, world, 0) and dist.get_backend() == "gloo"
    model, image, label = _build(method, csc)
    D.broadcast_parameters(model.prompt_learner)
    if shard:
        model.enable_class_sharding(rank, world)
    per = image.shape[0] // world
    grads, loss = _step(model, image[rank * per:(rank + 1) * per], label[rank * per:(rank + 1) * per])
    # the trainer's gradient exchange: ONE flat all-reduce (TrainerX.sync_gradients)
    D.all_reduce_gradients(model.prompt_learner.parameters(), world)
    ret[rank] = ({n: p.grad.detach().cpu().clone() for n, p in model.prompt_learner.named_parameters()}, loss)
    dist.destroy_process_group()


@pytest.mark.parametrize("method,csc,shard", [("coop", False, False), ("coop", False, True), ("coop", True, True),
                                              ("upt", False, True), ("vpt", False, False)])
def test_two_processes_match_single_process(method, csc, shard):
    import torch.multiprocessing as mp
    model, image, label = _build(method, csc)
    ref, ref_loss = _step(model, image, label)
    del model
    torch.cuda.empty_cache()
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), method, csc, shard, ret), nprocs=world, join=True)
    assert abs(sum(ret[r][1] for r in range(world)) / world - ref_loss) < 2e-4
    for r in range(world):
        for n, g in ref.items():
            got = ret[r][0][n]
            err = float((got - g).abs().max()) / (float(g.abs().max()) + 1e-20)
            # same kernels, different batch split: only the summation order of the per-image contributions differs
            assert err < 3e-4, f"rank {r} {n}: {err}"
