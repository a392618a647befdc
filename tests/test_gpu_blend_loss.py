"""The blending step's fused loss (loss._BlendStepLossFn: nudf_blend_loss_prepare / _fwd / _bwd around one torch.sort) against the
generic path -- ColorLoss's torch expressions (loss/loss.py:105-133 of the reference), the runner's patch-mask algebra
(exp_runner_blending.py:313-315), the trimmed SSIM patch loss (:66-84), the regularisers and the weighted total (:330-371) --
on the SAME render: the total, every logged term, and every parameter gradient of the step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, n_rays):
    from neuraludf_amd import synth
    from neuraludf_amd.train import Trainer
    rconf = dict(n_samples=32, n_importance=16, n_outside=0, up_sample_steps=2, perturb=1.0, upsampling_type="mix",
                 use_norm_grad_for_cosine=True, h_patch_size=3)
    tr = Trainer(dev, rconf, color_loss_conf=dict(color_pixel_weight=0.5, color_patch_weight=0.1), seed=0,
                 train_conf=dict(igr_ns_weight=0.05, sparse_weight=0.01))
    scene = synth.make_scene("tiny")
    rays = synth.make_rays(scene, 0, n_rays, seed=7, margin=8)
    batch = {k: v.to(dev) for k, v in rays.items()}
    g = torch.Generator().manual_seed(3)
    batch["gt_patch_colors"] = torch.rand(n_rays, 49, 3, generator=g).to(dev)
    blend = {k: v.to(dev) for k, v in synth.make_source_views(scene, 0, 8, hwc=True).items()}
    return tr, batch, blend


@pytest.mark.parametrize("n_rays", [96, 1000])
def test_fused_blend_step_loss_equals_the_generic_expressions(n_rays):
    dev = torch.device("cuda:0")
    tr, batch, blend = _setup(dev, n_rays)
    res = {}
    for fused in (True, False):
        tr.fuse_blend_loss = fused
        for m in tr.modules().values():
            m.zero_grad()
        loss, out = tr.loss(batch, cos_anneal_ratio=0.8, flip_saturation=0.9, blend=blend, perturb_overwrite=0)
        loss.backward()
        torch.cuda.synchronize()
        grads = {f"{k}.{n}": p.grad.detach().clone() for k, m in tr.modules().items() for n, p in m.named_parameters()
                 if p.grad is not None}
        res[fused] = (float(loss), {k: float(out[k]) for k in ("gradient_error", "gradient_error_near_surface", "sparse_error")},
                      grads)
    lf, tf, gf = res[True]
    lg, tg, gg = res[False]
    assert abs(lf - lg) <= 2e-6 * max(1.0, abs(lg)), (lf, lg)
    for k in tf:
        assert abs(tf[k] - tg[k]) <= 1e-6 * max(1.0, abs(tg[k])), (k, tf[k], tg[k])
    assert set(gf) == set(gg) and len(gf) > 50
    worst = ("", 0.0)
    for k in gg:
        den = float(gg[k].abs().max())
        if den == 0.0:
            assert float(gf[k].abs().max()) == 0.0, k
            continue
        r = float((gf[k] - gg[k]).abs().max()) / den
        if r > worst[1]:
            worst = (k, r)
        assert r < 2e-5, (k, r)        # different association of the N-term sums only
    print(f"fused blend step loss vs generic ({n_rays} rays): loss {lf:.7f} / {lg:.7f}, worst parameter gradient {worst[0]} {worst[1]:.1e}")


def test_fused_blend_step_is_few_launches():
    """the point of the fusion: between the render and the backward of the render the step's loss is a handful of launches"""
    dev = torch.device("cuda:0")
    tr, batch, blend = _setup(dev, 96)
    counts = {}
    for fused in (True, False):
        tr.fuse_blend_loss = fused
        tr.loss(batch, blend=blend)            # warm the caches
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            loss, _ = tr.loss(batch, blend=blend)
            loss.backward()
            torch.cuda.synchronize()
        counts[fused] = sum(e.count for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA)
    print("device launches of loss + backward:", counts)
    assert counts[True] <= counts[False] - 40, counts
