"""One process per GPU; RCCL (torch.distributed backend "nccl" on ROCm) over xGMI, gloo on CPU for tests.

The prompted-CLIP step shards over images with NO data-path collective in the towers: every rank holds the
frozen CLIP and its slice of the batch.  The single exchange per step is one all-reduce (sum) of ONE flat fp32
buffer holding every prompt-learner gradient — 8 192 elements for CoOp-16, 73 728 for VPT-deep-8, 566 400 for
UPT-4 (32 KB … 2.3 MB): latency-bound, so it is a single collective instead of one per parameter
(SURVEY.md §8e).  It replaces nn.DataParallel's per-step replicate/scatter/gather (trainers/mvlpt.py:877-880).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def _share_one_gpu() -> bool:
    """MVLPT_DEBUG_SHARE_GPU=1: every rank uses cuda:0 and the collectives go through gloo — lets the N > 1 code path
    (rendezvous, broadcast, gradient all-reduce, class sharding, bench timing protocol) be exercised with the real HIP
    engine on a ONE-GPU box.  Debug aid only: RCCL cannot place two ranks on one device."""
    return os.environ.get("MVLPT_DEBUG_SHARE_GPU", "0") == "1"


def rank_info() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1 process when absent)."""
    local = 0 if _share_one_gpu() else int(os.environ.get("LOCAL_RANK", 0))
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), local


def init_process_group(backend: str | None = None) -> Tuple[int, int, int]:
    rank, world, local = rank_info()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if (torch.cuda.is_available() and not _share_one_gpu()) else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _host_staged() -> bool:
    """gloo moves host memory: device tensors are staged through the host (debug mode MVLPT_DEBUG_SHARE_GPU and CPU tests
    only; RCCL works on device memory directly)."""
    return dist.is_initialized() and dist.get_backend() == "gloo"


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if _host_staged() and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class FlatGradients:
    """ONE persistent fp32 buffer that the parameters' `.grad` are views of (what DDP calls gradient_as_bucket_view).
    Autograd accumulates into an existing `.grad` in place and zeroing is one launch on the buffer, so the views survive the
    training loop; the per-step exchange is then a single in-place collective on `flat` — no torch.cat, no copy back, no
    scaling launch (RCCL averages itself: ReduceOp.AVG).  N > 1 adds exactly ONE launch per step.

    A parameter's `.grad` is adopted (copied into its slot and replaced by the view) the first time it HAS one — at the first
    exchange or `zero_()` after its first backward.  A parameter that never receives a gradient keeps `.grad is None`, exactly as
    under Dassl's `optimizer.zero_grad()`, so SGD skips it (no weight decay / momentum drift on an unused prompt tensor,
    checkpoints equal to the reference's); its slot in the buffer just stays zero."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("FlatGradients: fp32 master parameters expected")
            v = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self.views.append(v)
        self.attach()

    def attach(self) -> None:
        """Adopt every `.grad` that exists and is not yet its view."""
        for p, v in zip(self.params, self.views):
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v

    def intact(self) -> bool:
        """False when a `.grad` exists that is not its view (a first backward, zero_grad(set_to_none=True) + backward, a
        tensor assigned by hand): the next zero_() / all_reduce_mean_() adopts it."""
        return all(p.grad is None or p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))

    def adopted(self) -> int:
        return sum(1 for p, v in zip(self.params, self.views) if p.grad is not None and p.grad.data_ptr() == v.data_ptr())

    def zero_(self) -> None:
        if not self.intact():
            self.attach()
        self.flat.zero_()

    def all_reduce_mean_(self, world_size: int) -> None:
        if world_size <= 1 or self.flat.numel() == 0:
            return
        if not self.intact():
            self.attach()
        if _host_staged() or not self.flat.is_cuda:
            all_reduce_sum_(self.flat)                      # gloo: no AVG; host-staged in the one-GPU debug mode
            self.flat.mul_(1.0 / world_size)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], world_size: int) -> None:
    """grad <- mean over ranks, through one flat buffer (each rank's loss is the mean over its own slice, all
    slices have the same size, so the mean of rank gradients is the gradient of the global-batch mean loss)."""
    grads: List[torch.Tensor] = [p.grad for p in params if p.requires_grad and p.grad is not None]
    if world_size <= 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    all_reduce_sum_(flat)
    flat.mul_(1.0 / world_size)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    if not dist.is_initialized():
        return
    tensors = [p.data for p in module.parameters()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    if _host_staged() and flat.is_cuda:
        h = flat.cpu()
        dist.broadcast(h, src=src)
        flat.copy_(h)
    else:
        dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def all_reduce_max(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if _host_staged() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
