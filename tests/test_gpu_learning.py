"""End-to-end learning check on the GPU: a student network trained with the full loop body (schedules -> generated ray
batches -> render -> loss -> backward -> fused Adam) on views rendered by a teacher network reduces its colour error
on the training rays and on a held-out view.  Everything on the HIP path; no oracle involved."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_student_learns_teacher_views():
    from neuraludf_amd import synth
    from neuraludf_amd.dataset import RayBatchSource
    from neuraludf_amd.schedules import Schedules
    from neuraludf_amd.train import Trainer
    dev = torch.device("cuda:0")
    rconf = dict(n_samples=32, n_importance=16, n_outside=0, up_sample_steps=2, perturb=1.0)
    scene = synth.make_scene("tiny")
    n_views = 7
    dummy = torch.zeros(n_views, scene.H, scene.W, 3)
    src = RayBatchSource(dummy, torch.ones_like(dummy), scene.intrinsics[:n_views], scene.c2w[:n_views])
    teacher = Trainer(dev, rconf, seed=1)
    with torch.no_grad():      # make the teacher's colours view-dependent and non-trivial
        for p in teacher.color.parameters():
            p.mul_(1.5)
    src.images = torch.stack([teacher.render_image(src, i, resolution_level=1)["color"].clamp(0, 1) for i in range(n_views)])
    assert torch.isfinite(src.images).all() and float(src.images.std()) > 1e-3
    held_out = n_views - 1

    student = Trainer(dev, rconf, seed=0, fused_adam=True)
    sched = Schedules(end_iter=400, learning_rate=2e-3, learning_rate_geo=2e-4, learning_rate_alpha=0.05, warm_up_end=20.0,
                      anneal_end=0.0, fix_geo_end=0, color_base_weight=0.01, color_weight=1.0)

    def psnr_on(view):
        img = student.render_image(src, view, resolution_level=2)["color"]
        # gen_rays_at samples the pixel grid linspace(0, W-1, W // 2): the matching ground truth is the align_corners
        # bilinear resize of the full-resolution teacher view
        gt = torch.nn.functional.interpolate(src.images[view].permute(2, 0, 1)[None], size=img.shape[:2], mode="bilinear",
                                             align_corners=True)[0].permute(1, 2, 0)
        mse = float(((img - gt) ** 2).mean())
        return 20.0 * math.log10(1.0 / math.sqrt(mse + 1e-12))

    p0 = psnr_on(held_out)
    losses = []
    perm = torch.arange(n_views - 1)
    for it in range(300):
        loss, _, _ = student.iteration(src, it, sched, image_perm=perm, batch_size=512)
        losses.append(float(loss))
    p1 = psnr_on(held_out)
    first, last = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
    print(f"loss {first:.4f} -> {last:.4f}; held-out PSNR {p0:.2f} -> {p1:.2f} dB")
    assert all(math.isfinite(x) for x in losses)
    assert last < 0.6 * first, (first, last)
    assert p1 > p0 + 3.0, (p0, p1)
