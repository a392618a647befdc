"""Ray-sharded data parallelism (new functionality: the reference is single-GPU).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in
the CPU tests).  Rays are independent units, so rank r renders rays [r*N/G, (r+1)*N/G) of each
batch and there is NO data-path collective.  The only exchange steps are
  (1) one all-reduce(SUM) of the packed batch-global loss partial sums (eikonal numerators /
      denominators, sparsity sum, L1 numerators, mask counts: <= 16 floats, latency-bound), and
  (2) one all-reduce(SUM) of the flat 1.29 M-float gradient bucket after backward.
Every rank then holds the same global loss value L and d L / d theta = sum over ranks of the
local backward, which is exactly the single-process gradient on the full batch
(tests/test_dist_gloo.py checks the algebra with world_size 2)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class _AllReduceSum(torch.autograd.Function):
    """y = sum_r x_r ; every rank differentiates the SAME global loss, so dL/dx_r = dL/dy."""

    @staticmethod
    def forward(ctx, x):
        y = x.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        return y

    @staticmethod
    def backward(ctx, g):
        return g


def all_reduce_sum(x: torch.Tensor) -> torch.Tensor:
    if world_size() == 1:
        return x
    return _AllReduceSum.apply(x)


def shard(t: torch.Tensor, r: int = None, w: int = None) -> torch.Tensor:
    """contiguous block of rows for this rank (rays / per-ray targets)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    n = t.shape[0]
    per = (n + w - 1) // w
    return t[r * per: min((r + 1) * per, n)]


class GradBucket:
    """one flat fp32 bucket for all parameter gradients -> a single all-reduce(SUM) per step.
    On the 8-GPU xGMI mesh a 5.17 MB message is latency/per-link bound either way; one bucket keeps it to a
    single collective launch.  Packing is one `cat` kernel and unpacking one multi-tensor copy (not 2 x 62
    tiny launches)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]

    def all_reduce(self):
        if world_size() == 1 or not self.params:
            return
        # parameters that took no part in this step (e.g. the background NeRF when n_outside = 0: 0.6 M of the
        # 1.29 M floats) have grad None on EVERY rank -- the ranks run the same graph on their ray shards -- and
        # stay out of the message; Adam skips them exactly as in the single-process step
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        grads = [p.grad for p in live]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])
