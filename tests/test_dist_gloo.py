"""Ray-sharded data parallelism, world_size 2 on CPU (gloo): the sharded loss / gradient algebra of
neuraludf_amd/dist.py + the data-parallel ColorLoss equals the single-process result on the full batch.

The HIP kernels cannot run here (no GPU), so the per-ray renderer is replaced by the CPU oracle's
differentiable render (test infrastructure); what is under test is everything that is NOT per-ray:
the packed partial-sum all-reduce with its autograd rule, the globally-trimmed patch loss, `shard`,
and the flat gradient bucket."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_problem(n_rays=24, seed=0):
    """a small differentiable per-ray 'renderer' with the same batch-global reductions as render_core:
    colour [N,3], two masked eikonal sums, a sparsity sum, per-ray patch errors."""
    g = torch.Generator().manual_seed(seed)
    theta = torch.randn(7, generator=g, dtype=torch.float64)
    x = torch.randn(n_rays, 7, generator=g, dtype=torch.float64)
    gt = torch.rand(n_rays, 3, generator=g, dtype=torch.float64)
    pmask = torch.rand(n_rays, generator=g) > 0.3
    return theta, x, gt, pmask


def _local_terms(theta, x):
    h = torch.tanh(x * theta)                               # [n,7]
    color = torch.sigmoid(h[:, :3] + h[:, 3:6])
    gm = (h ** 2).sum(-1).sqrt()
    m1 = (x[:, 0] < 0.8).double()
    m2 = (x[:, 1] < 0.1).double()
    e = (gm - 1.0) ** 2
    sums = torch.stack([(m1 * e).sum(), m1.sum(), (m2 * e).sum(), m2.sum(), torch.exp(-h[:, 6].abs()).sum()])
    patch_err = (h[:, 2] - h[:, 5]).abs()
    return color, sums, patch_err


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neuraludf_amd import dist as nd
        from neuraludf_amd.loss.loss import ColorPixelLoss, _global_trimmed_mean
        theta, x, gt, pmask = _toy_problem()
        n = x.shape[0]
        theta = theta.clone().requires_grad_(True)
        xs, gts, pms = nd.shard(x), nd.shard(gt), nd.shard(pmask)
        color, sums, perr = _local_terms(theta, xs)
        sums = nd.all_reduce_sum(sums)
        pixel = ColorPixelLoss()
        pixel.data_parallel = True
        ge = sums[0] / (sums[1] + 1e-5)
        gens = sums[2] / (sums[3] + 1e-5)
        sp = sums[4] / float(n)
        l1 = pixel(color, gts, torch.ones(color.shape[0], 1, dtype=torch.float64))
        lp = _global_trimmed_mean(perr * pms.double(), pms, 0.3)
        loss = l1 + 0.1 * ge + 0.05 * gens + 0.01 * sp + 0.5 * lp
        loss.backward()
        bucket = nd.GradBucket([theta])
        bucket.all_reduce()
        ret[rank] = (float(loss), theta.grad.clone())
    finally:
        dist.destroy_process_group()


def test_sharded_loss_and_gradients_equal_single_process():
    sys.path.insert(0, ROOT)
    from neuraludf_amd.loss.loss import ColorPixelLoss
    theta, x, gt, pmask = _toy_problem()
    theta = theta.clone().requires_grad_(True)
    color, sums, perr = _local_terms(theta, x)
    pixel = ColorPixelLoss()
    ge = sums[0] / (sums[1] + 1e-5)
    gens = sums[2] / (sums[3] + 1e-5)
    sp = sums[4] / float(x.shape[0])
    l1 = pixel(color, gt, torch.ones(color.shape[0], 1, dtype=torch.float64))
    # single-process trimmed mean (loss/loss.py:79-84)
    err = perr * pmask.double()
    es, idx = torch.sort(err, descending=True)
    mk = pmask[idx].clone()
    mk[:int(0.3 * mk.sum())] = False
    lp = es[mk].mean()
    loss = l1 + 0.1 * ge + 0.05 * gens + 0.01 * sp + 0.5 * lp
    loss.backward()

    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        l, g = ret[r]
        assert abs(l - float(loss)) < 1e-6, (l, float(loss))   # ColorPixelLoss forms its denominator in fp32
        assert float((g - theta.grad).abs().max()) < 1e-6


def test_shard_covers_batch_without_overlap():
    sys.path.insert(0, ROOT)
    from neuraludf_amd import dist as nd
    t = torch.arange(103)
    for w in (1, 2, 4, 8):
        parts = [nd.shard(t, r, w) for r in range(w)]
        assert torch.equal(torch.cat(parts), t)
