"""world_size-2 gloo test (CPU) of the only collective on the path: ONE flat all-reduce of the prompt gradients.
Checks that rank-averaged gradients equal the gradient of the global-batch mean loss."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mvlpt_amd import distributed as D
    assert D.init_process_group("gloo") == (rank, world, rank)
    torch.manual_seed(0)
    # a stand-in "prompt learner": several parameters of different shapes (ctx, vpt, deep, projection)
    params = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s)) for s in [(16, 8), (1, 4, 8), (3, 4, 8), (5,)]])
    if rank == 1:
        with torch.no_grad():
            for p in params:
                p.add_(1.0)                      # ranks start different ...
    D.broadcast_parameters(params)               # ... and must end identical to rank 0
    g = torch.Generator().manual_seed(123)
    X = torch.randn(8, 8, generator=g)           # global batch of 8, 4 per rank

    def loss_fn(x):
        h = x @ params[0].t()
        return (h.sum(-1) * params[3].sum() + (params[1] * params[2][:1]).sum()).pow(2).mean()

    loss_fn(X[rank * 4:(rank + 1) * 4]).backward()
    D.all_reduce_gradients(params, world)
    got = [p.grad.clone() for p in params]
    # the same through the persistent flat buffer the trainer uses (FlatGradients): .grad are views of ONE tensor, autograd
    # accumulates into them in place across steps, the exchange is one in-place collective
    for p in params:
        p.grad = None
    fg = D.FlatGradients(params)
    assert all(p.grad is None for p in params) and fg.intact() and fg.adopted() == 0    # nothing is adopted before it exists
    ptrs = [v.data_ptr() for v in fg.views]
    for step in range(3):                        # the first exchange adopts the fresh grads; the views then survive zero + backward
        fg.zero_()
        loss_fn(X[rank * 4:(rank + 1) * 4]).backward()
        assert fg.intact() == (step > 0)
        fg.all_reduce_mean_(world)
        assert fg.intact() and fg.adopted() == len(params) and [p.grad.data_ptr() for p in params] == ptrs
    flat_ok = all(torch.allclose(a, p.grad, rtol=1e-6, atol=1e-7) for a, p in zip(got, params))
    flat_ok = flat_ok and fg.flat.numel() == sum(p.numel() for p in params)
    params[0].grad = torch.ones_like(params[0])  # somebody replaced a grad: adopted again at the next zero_()
    fg.zero_()
    flat_ok = flat_ok and fg.intact() and params[0].grad.data_ptr() == ptrs[0] and float(params[0].grad.abs().sum()) == 0.0
    # a parameter that never receives a gradient keeps .grad None (SGD skips it, as under Dassl's zero_grad): ADVICE r3
    unused = torch.nn.Parameter(torch.ones(3))
    fg2 = D.FlatGradients([params[3], unused])
    fg2.zero_()
    (params[3].sum() * 2).backward()
    fg2.all_reduce_mean_(world)
    flat_ok = flat_ok and unused.grad is None and fg2.adopted() == 1 and fg2.intact()
    for p in params:
        p.grad = None
    loss_fn(X).backward()                        # single-process reference on the concatenated batch
    ok = all(torch.allclose(a, p.grad, rtol=1e-5, atol=1e-6) for a, p in zip(got, params))
    mx = D.all_reduce_max(float(rank + 1), torch.device("cpu"))
    D.barrier()
    ret[rank] = bool(ok and flat_ok and mx == float(world))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_matches_global_batch():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
