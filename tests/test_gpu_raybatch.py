"""GPU: nudf_gen_ray_batch (through RayBatchSource -> C ABI) against the reference's own output (golden fixture)
and against the oracle at the full DTU image size."""
import numpy as np
import pytest
import torch

from neuraludf_amd.dataset import RayBatchSource
from oracle import rays_oracle as ro
from test_rays_oracle import CASES, check_sample, load_gold

pytestmark = pytest.mark.gpu


def to_np(s):
    return {k: (None if v is None else v.cpu().numpy()) for k, v in s.items()}


def source(g):
    return RayBatchSource(g["images"], g["masks"], g["intrinsics_all"], g["pose_all"])


@pytest.mark.parametrize("name", CASES)
def test_matches_reference_fixture(name):
    g = load_gold()
    src = source(g)
    crop = f"{name}.rays_patch_color" in g
    s = to_np(src.rays_at_pixels(int(g[f"{name}.img_idx"]), torch.from_numpy(g[f"{name}.px"]),
                                 torch.from_numpy(g[f"{name}.py"]), int(g[f"{name}.h"]), crop, with_near_far=True))
    check_sample(g, name, s, s["near"], s["far"], atol=5e-6)
    n2, f2 = src.near_far_from_sphere(torch.from_numpy(s["rays"][:, :3]).cuda(), torch.from_numpy(s["rays"][:, 3:6]).cuda())
    np.testing.assert_allclose(n2.cpu().numpy(), s["near"], atol=1e-5)


def test_importance_draws_follow_reference_order():
    """same generator state -> same pixels as the reference's draw sequence (randint W, randint H, randint n_valid)."""
    g = load_gold()
    src = source(g)
    i = int(g["importance.img_idx"])
    B = g["importance.px"].shape[0]
    gen = torch.Generator(device="cuda").manual_seed(5)
    px, py = src.draw_pixels(i, B, True, gen)
    gen.manual_seed(5)
    kw = dict(device="cuda", generator=gen)
    x1 = torch.randint(0, src.W, [B // 4], **kw); y1 = torch.randint(0, src.H, [B // 4], **kw)
    valid = torch.nonzero(src.masks[i][:, :, 0] > 0)
    sel = valid[torch.randint(0, valid.shape[0], [B // 4 * 3], **kw)]
    assert torch.equal(px, torch.cat([x1, sel[:, 1]])) and torch.equal(py, torch.cat([y1, sel[:, 0]]))
    assert (src.masks[i][py[B // 4:], px[B // 4:], 0] > 0).all()
    s = src.gen_random_rays_patches_at(i, B, importance_sample=True, crop_patch=True)
    assert s["rays"].shape == (B, 10) and s["rays_patch_color"].shape == (B, 49, 3) and s["rays_patch_mask"].dtype == torch.bool


def test_ref_src_pairs_match_reference():
    g = load_gold()
    src = source(g)
    pairs = src.prepare_ref_src_pairs()
    got = np.stack([pairs[i].cpu().numpy() for i in range(src.n_images)])
    # the ring scene has equidistant neighbours, so tied entries may come in either order: compare the distances
    loc = g["pose_all"][:, :3, 3]
    dist = np.linalg.norm(loc[:, None] - loc[None], axis=-1)
    np.testing.assert_allclose(np.take_along_axis(dist, got, 1), np.take_along_axis(dist, g["ref_src_pairs"], 1), atol=1e-5)
    assert all(i not in got[i] for i in range(src.n_images))
    ref_c2w, c2ws, intr, imgs, wh = src.get_ref_src_info(1, num=2)
    assert c2ws.shape == (2, 4, 4) and intr.shape == (2, 4, 4) and imgs.shape == (2, 3, src.H, src.W) and wh == [src.W, src.H]
    assert torch.equal(imgs[0].permute(1, 2, 0), src.images[pairs[1][0]])
    # the cached world-to-camera matrices replace the runner's per-iteration torch.inverse(src_c2ws) (:283-285)
    w2c = src.src_w2cs(1, num=2)
    assert w2c.shape == (2, 4, 4) and float((w2c - torch.inverse(c2ws)).abs().max()) < 1e-6
    assert float((w2c @ c2ws - torch.eye(4, device=w2c.device)).abs().max()) < 1e-5
    assert src.src_w2cs(1, num=2).data_ptr() != 0 and src._w2c_all.shape == src.pose_all.shape


def test_full_size_vs_oracle():
    """1200 x 1600 view, 512 rays incl. all four corners, 7x7 patches: bit-exact gathers, ulp-level floats."""
    rng = np.random.default_rng(0)
    H, W = 1200, 1600
    img = (rng.integers(0, 256, (1, H, W, 3)).astype(np.float32) / np.float32(256))
    msk = np.repeat((rng.random((1, H, W, 1)) > 0.5).astype(np.float32), 3, -1)
    K = np.eye(4, dtype=np.float32); K[0, 0] = 2892.33; K[1, 1] = 2883.18; K[0, 2] = 823.2; K[1, 2] = 619.07
    a = 0.7
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    pose[:3, 3] = [1.5, -0.4, 2.0]
    src = RayBatchSource(img, msk, K[None], pose[None])
    px = rng.integers(0, W, 512); py = rng.integers(0, H, 512)
    px[:4] = [0, W - 1, 0, W - 1]; py[:4] = [0, 0, H - 1, H - 1]
    s = to_np(src.rays_at_pixels(0, torch.from_numpy(px), torch.from_numpy(py), 3, True, with_near_far=True))
    o = ro.gen_rays_patches(img[0], msk[0], src.intrinsics_all_inv[0].cpu().numpy(), pose, px, py, 3, True)
    on, of = ro.near_far_from_sphere(o["rays"][:, :3], o["rays"][:, 3:6])
    np.testing.assert_array_equal(s["rays"][:, 6:], o["rays"][:, 6:])
    np.testing.assert_array_equal(s["rays"][:, :3], o["rays"][:, :3])
    np.testing.assert_array_equal(s["rays_patch_mask"], o["rays_patch_mask"])
    np.testing.assert_array_equal(s["rays_ndc_uv"], o["rays_ndc_uv"])
    np.testing.assert_allclose(s["rays"][:, 3:6], o["rays"][:, 3:6], atol=3e-7)
    np.testing.assert_allclose(s["rays_norm_XYZ_cam"], o["rays_norm_XYZ_cam"], atol=3e-7)
    np.testing.assert_allclose(s["rays_patch_color"], o["rays_patch_color"], atol=3e-6)
    np.testing.assert_allclose(s["near"], on, atol=2e-6); np.testing.assert_allclose(s["far"], of, atol=2e-6)
    # unit directions, and empty batches are a no-op
    np.testing.assert_allclose(np.linalg.norm(s["rays"][:, 3:6], axis=-1), 1.0, atol=1e-6)
    e = src.rays_at_pixels(0, torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), 3, True)
    assert e["rays"].shape == (0, 10) and e["rays_patch_color"].shape == (0, 49, 3)


def test_training_iterations_from_generated_batches():
    """the reference loop body end to end on generated batches: schedules -> nudf_gen_ray_batch -> render -> loss ->
    backward -> FusedAdam, first with the DTU schedule (no blending at iteration 0), then fine-tune (pixel + patch
    blending on, 5x5 ... here 7x7 patches); losses finite, weights move, learning rates follow the schedule."""
    from neuraludf_amd.schedules import Schedules
    from neuraludf_amd.train import Trainer
    g = load_gold()
    rng = np.random.default_rng(0)
    n, H, W = 10, g["images"].shape[1], g["images"].shape[2]
    from neuraludf_amd import synth
    scene = synth.make_scene("tiny")
    images = rng.random((n, H, W, 3), dtype=np.float32)
    src = RayBatchSource(images, np.ones_like(images), scene.intrinsics[:n].numpy(), scene.c2w[:n].numpy())
    conf = dict(n_samples=16, n_importance=8, n_outside=0, up_sample_steps=2, perturb=1.0)
    base = dict(end_iter=300000, learning_rate=5e-4, learning_rate_geo=1e-4, learning_rate_alpha=0.05, warm_up_end=5000.0,
                anneal_end=25000.0, color_base_weight=0.01, color_weight=1.0, color_pixel_weight=0.1, color_patch_weight=0.1)
    for ft in (False, True):
        tr = Trainer(torch.device("cuda"), conf, color_loss_conf=dict(color_pixel_weight=0.1, color_patch_weight=0.1),
                     fused_adam=True, seed=0)
        sched = Schedules(is_finetune=ft, **base)
        w0 = tr.color.state_dict()["lin_base0.weight_v"].clone()
        for it in (600, 601):
            loss, out, s = tr.iteration(src, it, sched, batch_size=64)
            assert torch.isfinite(loss)
        assert s["rays"].shape == (64, 10)
        assert (out["color_pixel"] is not None) == ft and (s["rays_patch_color"] is not None) == ft
        lrs = [g_["lr"] for g_ in tr.optimizer.param_groups]
        assert lrs[1] == pytest.approx(5e-4 * 601 / 5000) and lrs[0] == pytest.approx(1e-4 * 601 / 10000)
        assert not torch.equal(w0, tr.color.state_dict()["lin_base0.weight_v"])


def test_whole_image_rays_match_reference():
    """gen_rays_at / gen_rays_between / gen_random_rays_at against the reference Dataset's own output."""
    g = load_gold()
    src = source(g)
    ro, rv = src.gen_rays_at(2, resolution_level=4)
    assert ro.shape == (src.H // 4, src.W // 4, 3)
    np.testing.assert_allclose(ro.cpu().numpy(), g["rays_at.o"], atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), g["rays_at.v"], atol=5e-6)
    ro, rv = src.gen_rays_between(1, 3, 0.3, resolution_level=8)
    np.testing.assert_allclose(ro.cpu().numpy(), g["rays_between.o"], atol=1e-5)
    np.testing.assert_allclose(rv.cpu().numpy(), g["rays_between.v"], atol=1e-5)
    rays = src.rays_at_pixels(3, torch.from_numpy(g["random_rays_at.px"]), torch.from_numpy(g["random_rays_at.py"]))["rays"]
    ref = g["random_rays_at"]
    np.testing.assert_array_equal(rays.cpu().numpy()[:, [0, 1, 2, 6, 7, 8, 9]], ref[:, [0, 1, 2, 6, 7, 8, 9]])
    np.testing.assert_allclose(rays.cpu().numpy()[:, 3:6], ref[:, 3:6], atol=5e-6)
    assert src.gen_random_rays_at(3, 40).shape == (40, 10)


def test_render_image_chunking_is_exact():
    """a validation view rendered in one chunk and in ragged chunks gives the same image (rays are independent)."""
    from neuraludf_amd.train import Trainer
    g = load_gold()
    src = source(g)
    tr = Trainer(torch.device("cuda"), dict(n_samples=16, n_importance=8, n_outside=4, up_sample_steps=2, perturb=0.0), seed=0)
    a = tr.render_image(src, 1, resolution_level=4, chunk=1 << 20)
    b = tr.render_image(src, 1, resolution_level=4, chunk=100)
    assert a["color"].shape == (src.H // 4, src.W // 4, 3) and a["depth"].shape == (src.H // 4, src.W // 4)
    for k in a:
        assert torch.isfinite(a[k]).all()
        assert float((a[k] - b[k]).abs().max()) < 1e-5, k
