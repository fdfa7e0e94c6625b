"""RCCL over xGMI with N > 1 (-m gpu, skipped on a one-GPU box): two processes, one GPU each, torch.distributed backend
"nccl" (= RCCL on ROCm).  Runs by itself the moment a box has two devices — the first hardware evidence of the N > 1 path that
does not go through gloo / a shared GPU (tests/test_hip_multiproc.py, test_hip_bench_contract.py):

* the data-parallel step (image slices, ONE in-place all_reduce(AVG) of the flat prompt-gradient buffer) and the class-sharded text
  tower (all_gather_into_tensor of the features, reduce_scatter_tensor of their gradients; even and ragged shards, class-specific
  contexts, UPT) reproduce the single-process gradients on the concatenated batch to 3e-4;
* `python bench.py --gpus 2` with no debug switch: self-spawned ranks on distinct devices, one JSON line with n_gpus = 2.
Replaces nn.DataParallel (trainers/mvlpt.py:877-880)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from tests.test_hip_multiproc import NAMES, _build, _free_port, _step

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, method, csc, shard, n_names, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("MVLPT_DEBUG_SHARE_GPU", None)
    import torch.distributed as dist
    from mvlpt_amd import distributed as D
    r, w, local = D.init_process_group()
    assert (r, w, local) == (rank, world, rank) and dist.get_backend() == "nccl" and not D._host_staged()
    torch.cuda.set_device(local)
    model, image, label = _build(method, csc, device=f"cuda:{local}", names=NAMES[:n_names])
    D.broadcast_parameters(model.prompt_learner)
    if shard:
        model.enable_class_sharding(rank, world)
    fg = D.FlatGradients(list(model.prompt_learner.parameters()))
    per = image.shape[0] // world
    _, loss = _step(model, image[rank * per:(rank + 1) * per], label[rank * per:(rank + 1) * per])
    # the trainer's gradient exchange: ONE in-place all_reduce(AVG) on the flat buffer (TrainerX.sync_gradients)
    fg.all_reduce_mean_(world)
    assert fg.intact()
    torch.cuda.synchronize()
    ret[rank] = ({n: p.grad.detach().cpu().clone() for n, p in model.prompt_learner.named_parameters()}, loss)
    dist.destroy_process_group()


@pytest.mark.parametrize("method,csc,shard,n_names", [("coop", False, False, 7), ("coop", False, True, 6), ("coop", False, True, 7),
                                                      ("coop", True, True, 7), ("upt", False, True, 7), ("vpt", False, False, 7)])
def test_two_ranks_on_two_gpus_match_single_process(method, csc, shard, n_names):
    import torch.multiprocessing as mp
    model, image, label = _build(method, csc, names=NAMES[:n_names])
    ref, ref_loss = _step(model, image, label)
    del model
    torch.cuda.empty_cache()
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), method, csc, shard, n_names, ret), nprocs=world, join=True)
    assert abs(sum(ret[r][1] for r in range(world)) / world - ref_loss) < 2e-4
    for r in range(world):
        for n, g in ref.items():
            err = float((ret[r][0][n] - g).abs().max()) / (float(g.abs().max()) + 1e-20)
            assert err < 3e-4, f"rank {r} {n}: {err}"


def test_bench_two_gpus_over_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MVLPT_DEBUG_SHARE_GPU", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "32",
                          "--no-cpu-baseline", "--no-trim-extra"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak" and line["value"] > 0
