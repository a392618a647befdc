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
        from neuraludf_amd.loss import loss as L
        from neuraludf_amd.loss.loss import ColorPixelLoss, _global_trimmed_mean
        L._l1_sum = lambda pred, gt: (pred - gt).abs().sum()      # host stand-in for the HIP reduction (test only)
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
    from neuraludf_amd.loss import loss as L
    from neuraludf_amd.loss.loss import ColorPixelLoss
    L._l1_sum = lambda pred, gt: (pred - gt).abs().sum()          # host stand-in for the HIP reduction (test only)
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
    t = torch.arange(104)
    for w in (1, 2, 4, 8):
        parts = [nd.shard(t, r, w) for r in range(w)]
        assert torch.equal(torch.cat(parts), t)
        assert len({p.shape[0] for p in parts}) == 1
    with pytest.raises(ValueError):          # unequal shards would break the renderer's batch-global normalisation
        nd.shard(torch.arange(103), 0, 2)


def _bucket_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neuraludf_amd import dist as nd
        g = torch.Generator().manual_seed(5)
        shapes = [(3, 4), (5,), (2, 2), (7,), (1,)]
        ps = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]

        class Eng:            # stands for an mlp engine: writes its gradients straight into its bucket segment
            grad_slot = None
        eng = Eng()
        b = nd.GradBucket([([ps[4]], None), (ps[0:2], eng), ([ps[2]], None), ([ps[3]], None)])
        assert eng.grad_slot is not None and eng.grad_slot.numel() == 17
        nd.collective_counts(reset=True)
        for step in range(2):
            for p in ps:
                p.grad = None
            # engine segment: gradients ARE bucket views (what autograd installs after unpack_group(slot=...))
            v0 = eng.grad_slot[:12].view(3, 4)
            v1 = eng.grad_slot[12:17]
            v0.copy_(torch.full((3, 4), float(rank + 1 + step)))
            v1.copy_(torch.arange(5.0) * (rank + 1))
            ps[0].grad, ps[1].grad = v0, v1
            ps[4].grad = torch.tensor([10.0 * (rank + 1)])        # a gradient that lives outside the bucket: copied in / out
            ps[2].grad = torch.ones(2, 2) * (rank + 2)
            # ps[3] (the trailing segment) took no part in the step: grad None, not sent
            b.all_reduce()
            assert b.last_message_floats == 1 + 17 + 4
            assert ps[0].grad.data_ptr() == v0.data_ptr()
        ret[rank] = ([p.grad.clone() if p.grad is not None else None for p in ps], nd.collective_counts())
    finally:
        dist.destroy_process_group()


def test_grad_bucket_views_copies_and_unused_tail():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bucket_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        grads, counts = ret[r]
        assert torch.equal(grads[0], torch.full((3, 4), 2.0 + 3.0))          # step 1: (1+1) + (2+1)
        assert torch.equal(grads[1], torch.arange(5.0) * 3)
        assert torch.equal(grads[4], torch.tensor([30.0]))
        assert torch.equal(grads[2], torch.ones(2, 2) * 5)
        assert grads[3] is None
        assert counts["all_reduce"] == 2 and counts["all_gather"] == 0            # one collective per step
