"""The RCCL branches of the N > 1 code (torch.distributed backend "nccl" = RCCL on ROCm) on the ONE GPU a test box has: a process
group of world size 1.  It cannot show scaling, but it runs exactly the calls an 8-GPU job makes — `all_reduce(AVG)` in place on the
flat gradient buffer, `all_gather_into_tensor` / `reduce_scatter_tensor` on the persistent class-shard buffers (issued on the side
stream, as CustomCLIP.forward does), broadcast, barrier, the float64 MAX all-reduce of the timing protocol — so an API or dtype
mismatch of those branches (only the gloo branches run in the 2-rank tests) shows up here instead of on the 8-GPU node."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.pop("MVLPT_DEBUG_SHARE_GPU", None)
    import torch.distributed as dist
    from mvlpt_amd import distributed as D
    from mvlpt_amd.model import ClassShard, class_shard_bounds
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl" and not D._host_staged()
        dev = torch.device("cuda:0")
        # gradient exchange: in-place AVG on the flat buffer the .grad views alias
        params = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s, device=dev)) for s in [(16, 512), (1, 8, 768), (11, 8, 768), (5,)]])
        fg = D.FlatGradients(params)
        (sum((p * p).sum() for p in params)).backward()
        want = [2 * p.detach() for p in params]
        fg.all_reduce_mean_(2)                         # "world 2" arithmetic on a group of one: AVG over one rank = identity
        assert fg.intact() and all(torch.equal(p.grad, w) for p, w in zip(params, want))
        D.broadcast_parameters(params)
        assert D.all_reduce_max(3.5, dev) == 3.5
        D.barrier()
        # class shard collectives on persistent buffers, issued on a side stream like the text tower
        sh = ClassShard(0, class_shard_bounds(1151, 1), dev)
        loc = torch.randn(1151, 768, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            txt = sh.gather(loc)
        torch.cuda.current_stream().wait_stream(side)
        assert txt.shape == (1151, 768) and torch.equal(txt, loc)
        d = torch.randn(1151, 768, device=dev)
        own = sh.scatter_grads(d)
        torch.cuda.synchronize()
        assert own.shape == (1151, 768) and torch.equal(own, d)
        ret["ok"] = True
    finally:
        dist.destroy_process_group()


def test_rccl_branches_on_a_group_of_one():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    p = mp.get_context("spawn").Process(target=_worker, args=(ret,))
    p.start()
    p.join(300)
    assert p.exitcode == 0 and ret.get("ok") is True
