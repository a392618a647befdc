"""RCCL smoke on one GPU: the process group the multi-GPU bench uses (backend "nccl" == RCCL) initialises in this
image and the two collectives of the data-parallel path (packed loss sums, flat gradient bucket) run through it.
With one rank the values are trivially unchanged; what is checked is that the RCCL code path itself works."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_collectives_single_rank():
    from neuraludf_amd import dist as nd
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        x = torch.arange(5, dtype=torch.float32, device=dev).requires_grad_(True)
        y = nd._AllReduceSum.apply(x * 2.0)            # the autograd rule of the packed loss sums
        y.sum().backward()
        assert torch.equal(y.detach(), x.detach() * 2.0)
        assert torch.equal(x.grad, torch.full_like(x, 2.0))
        ps = [torch.randn(7, 3, device=dev, requires_grad=True), torch.randn(11, device=dev, requires_grad=True),
              torch.randn(2, device=dev, requires_grad=True)]
        ps[0].grad, ps[1].grad = torch.ones_like(ps[0]), torch.full_like(ps[1], 3.0)      # ps[2].grad stays None
        b = nd.GradBucket(ps)
        flat = torch.cat([g.reshape(-1) for g in (ps[0].grad, ps[1].grad, torch.zeros_like(ps[2]))])
        dist.all_reduce(flat)
        torch.cuda.synchronize()
        assert float(flat.sum()) == 21.0 + 33.0
        # the bucket path itself (world_size() == 1 short-circuits it in production; call the body directly)
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b.params]
        fl = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(fl, op=dist.ReduceOp.SUM)
        torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(fl.split(b.sizes), grads)])
        assert torch.equal(grads[1], torch.full_like(ps[1], 3.0))
    finally:
        dist.destroy_process_group()
