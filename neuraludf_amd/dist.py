"""Ray-sharded data parallelism (new functionality: the reference is single-GPU).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in
the CPU tests).  Rays are independent units, so rank r renders rays [r*N/G, (r+1)*N/G) of each
batch and there is NO data-path collective.  The exchange steps of one training step are exactly two:
  (1) one all-reduce(SUM) of the PACKED batch-global partial sums -- the renderer's five (eikonal numerators /
      denominators, sparsity sum), the colour loss's three (both L1 numerators, the mask count) and, if the mask
      term is on, its two (<= 16 floats, latency-bound) -- with the autograd rule dL/dx_r = dL/dy;
  (2) one all-reduce(SUM) of the flat gradient bucket after backward.  The bucket is allocated once; the
      weight-gradient unpack kernels of the three networks write dv / dg / db straight into their segments of it
      (mlp.unpack_group(slot=...)), so a step issues no `cat` and no copy back -- `p.grad` ARE views of the bucket.
(The globally trimmed SSIM patch loss of the *_ft confs adds one all-gather of the per-ray errors + masks.)
Every rank then holds the same global loss value L and d L / d theta = sum over ranks of the local backward, which
is exactly the single-process gradient on the full batch (tests/test_dist_gloo.py, world_size 2)."""
from __future__ import annotations

import torch
import torch.distributed as dist

# collectives issued by this module since the last reset (bench.py / the tests report them per step)
_counts = {"all_reduce": 0, "all_gather": 0}


def collective_counts(reset=False):
    c = dict(_counts)
    if reset:
        for k in _counts:
            _counts[k] = 0
    return c


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# With ONE rank every collective is the identity and is skipped.  FORCE_COLLECTIVES = True issues them anyway (the process
# group must be initialised): what lets a one-GPU box run -- and CAPTURE in a HIP graph -- the exact launch sequence of a
# ray-sharded step, RCCL calls included (tests/test_gpu_dist.py).
FORCE_COLLECTIVES = False


def exchanging() -> bool:
    """do the collectives of a ray-sharded step run in this process?"""
    return world_size() > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class _AllReduceSum(torch.autograd.Function):
    """y = sum_r x_r ; every rank differentiates the SAME global loss, so dL/dx_r = dL/dy."""

    @staticmethod
    def forward(ctx, x):
        y = x.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        _counts["all_reduce"] += 1
        return y

    @staticmethod
    def backward(ctx, g):
        return g


def all_reduce_sum(x: torch.Tensor) -> torch.Tensor:
    if not exchanging():
        return x
    return _AllReduceSum.apply(x)


def all_gather_rows(t: torch.Tensor) -> torch.Tensor:
    """[n, ...] per rank (equal n) -> [world * n, ...] in rank order, no autograd (callers re-insert their own rows)."""
    w = world_size()
    out = torch.empty((w * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather(list(out.chunk(w, dim=0)), t.detach().contiguous())
    _counts["all_gather"] += 1
    return out


def shard(t: torch.Tensor, r: int = None, w: int = None) -> torch.Tensor:
    """contiguous block of rows for this rank (rays / per-ray targets).  Equal shards only: the renderer normalises
    its batch-global means with N_local * world_size and the patch loss gathers equally sized blocks."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    n = t.shape[0]
    if n % w != 0:
        raise ValueError("ray batch of %d rows does not split evenly over %d ranks (pad or drop the remainder)" % (n, w))
    per = n // w
    return t[r * per:(r + 1) * per]


class GradBucket:
    """One persistent flat fp32 buffer for all parameter gradients -> a single all-reduce(SUM) per step.

    `segments` = [(params, engine | None), ...] in the order the buffer is laid out; put the networks that can be
    unused in a step LAST (the background NeRF with n_outside = 0: 0.6 M of the 1.29 M floats) -- only the prefix up to
    the last live gradient is sent.  An `engine` (mlp.UDFEngine / ColorEngine / NerfEngine) gets `engine.grad_slot`
    = its segment: its unpack kernel then writes there and autograd installs the bucket views as `p.grad`.
    Gradients that arrive any other way (scalar networks, accumulated `.grad`s, engines on the per-layer path) are
    copied in before the collective and back after it."""

    def __init__(self, segments, device=None):
        if segments and not isinstance(segments[0], (tuple, list)):
            segments = [(list(segments), None)]           # plain parameter list
        self.params, self.views = [], []
        total = sum(p.numel() for ps, _ in segments for p in ps)
        dev = device
        if dev is None:
            dev = next((p.device for ps, _ in segments for p in ps), torch.device("cpu"))
        dtype = next((p.dtype for ps, _ in segments for p in ps), torch.float32)
        self.flat = torch.zeros(total, device=dev, dtype=dtype)
        off = 0
        for ps, eng in segments:
            n = sum(p.numel() for p in ps)
            if eng is not None and self.flat.dtype == torch.float32:
                eng.grad_slot = self.flat[off:off + n]
            for p in ps:
                self.params.append(p)
                self.views.append(self.flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
        self.sizes = [p.numel() for p in self.params]
        self.last_message_floats = 0

    def all_reduce(self):
        if not exchanging() or not self.params:
            return
        # parameters that took no part in this step have grad None on EVERY rank -- the ranks run the same graph on
        # their ray shards -- and Adam skips them exactly as in the single-process step
        end, off, copied_g, copied_v = 0, 0, [], []
        for p, v, n in zip(self.params, self.views, self.sizes):
            off += n
            g = p.grad
            if g is None:
                continue
            end = off
            if g.data_ptr() != v.data_ptr():
                copied_g.append(g)
                copied_v.append(v)
        if end == 0:
            return
        if copied_g:
            torch._foreach_copy_(copied_v, copied_g)
        dist.all_reduce(self.flat[:end], op=dist.ReduceOp.SUM)
        _counts["all_reduce"] += 1
        self.last_message_floats = end
        if copied_g:
            torch._foreach_copy_(copied_g, copied_v)
