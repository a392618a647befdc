"""Blending branch (pixel / patch warp, view softmax, composites, SSIM patch loss) against the oracle
and against the committed reference fixture."""
import os

import numpy as np
import pytest
import torch

from common import build_modules, perturb_, state_dicts, oracle_nets, grel
from neuraludf_amd import synth
from oracle import udf_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def rel(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1.0))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


from common import smooth_images as _smooth_images  # noqa: E402


@pytest.mark.parametrize("with_bg", [False, True])
def test_blend_stagewise(dev, with_bg):
    from neuraludf_amd.models import blend
    g = torch.Generator().manual_seed(4)
    scene = synth.make_scene("tiny")
    N, S, hps = 21, 37, 3
    n_out = 5 if with_bg else 0
    r = synth.make_rays(scene, 0, N, seed=8, margin=10)
    src = synth.make_source_views(scene, 0, 8)
    imgs = _smooth_images(8, scene.H, scene.W)
    z = torch.sort(r["near"] + (r["far"] - r["near"]) * torch.rand(N, S, generator=g), -1)[0]
    pts = r["rays_o"][:, None] + r["rays_d"][:, None] * z[..., None]
    grad = torch.randn(N, S, 3, generator=g)
    logits = torch.randn(N, S, 10, generator=g)
    w = torch.rand(N, S + n_out, generator=g) * 0.05
    bg_all = torch.rand(N, S + n_out, 3, generator=g) if with_bg else None

    lg = logits.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    bgr = bg_all.clone().requires_grad_(True) if with_bg else None
    pcol, pmask = O.pixel_warp(pts, imgs, src["intrinsics"], src["w2cs"])
    gnorm = grad / (torch.linalg.norm(grad, dim=-1, keepdim=True) + 1e-5)
    cs = (r["rays_d"][:, None] * gnorm).sum(-1, keepdim=True)
    flip = -torch.sign(cs)
    flip[flip == 0] = 1
    tcol, tmask = O.patch_warp(pts, r["rays_uv"], flip * gnorm, imgs, src["intrinsics"][0], src["intrinsics"],
                               src["query_c2w"], torch.inverse(src["w2cs"]), hps)
    pix, _, patch, pm = O.color_blend(lg, pcol, pmask, tcol, tmask)
    if with_bg:
        inside = (torch.linalg.norm(pts, dim=-1) < 1.0).float()
        pix = pix * inside[..., None] + bgr[:, :S] * (1 - inside)[..., None]
        pix = torch.cat([pix, bgr[:, S:]], 1)
    cp_ref = (pix * wr[:, :, None]).sum(1)
    pc_ref = (patch * wr[:, :S, None, None]).sum(1)
    pm_ref = (pm.float().reshape(N, S) * wr[:, :S]).sum(1)
    k1, k2 = torch.randn(N, 3, generator=g), torch.randn(N, 49, 3, generator=g)
    ((cp_ref * k1).sum() + (pc_ref * k2).sum()).backward()

    D = lambda t: t.to(dev)
    lgd = D(logits).requires_grad_(True)
    wd = D(w).requires_grad_(True)
    bgd = D(bg_all).requires_grad_(True) if with_bg else None
    cp, pc, pmk = blend.blend_and_composite(hps, D(pts), lgd, wd, D(grad), D(r["rays_d"]), D(imgs), D(src["w2cs"]),
                                            D(src["intrinsics"]), D(src["query_c2w"]), D(r["rays_uv"].clone()),
                                            bg_in=bgd[:, :S].contiguous() if with_bg else None,
                                            bg_tail=bgd[:, S:].contiguous() if with_bg else None)
    assert rel(cp, cp_ref) < 1e-4
    assert rel(pc, pc_ref) < 1e-4
    assert rel(pmk, pm_ref) < 1e-4
    ((cp * D(k1)).sum() + (pc * D(k2)).sum()).backward()
    assert grel(lgd.grad, lg.grad) < 1e-3
    assert grel(wd.grad, wr.grad) < 1e-3
    if with_bg:
        assert grel(bgd.grad, bgr.grad) < 1e-3


def test_blend_image_layouts_agree(dev):
    """channel-interleaved source images behind an NCHW view (what the reference's dataset hands over,
    dataset.py:147-149) are read in place and give bit-identical pixel / patch colours and gradients as the planar
    NCHW copy; ray counts on both sides of the 4- / 8-waves-per-ray switch."""
    from neuraludf_amd.models import blend
    g = torch.Generator().manual_seed(11)
    scene = synth.make_scene("tiny")
    S, hps = 19, 2
    src = synth.make_source_views(scene, 0, 8)
    planar = _smooth_images(8, scene.H, scene.W).to(dev)
    inter = planar.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert not inter.is_contiguous() and torch.equal(inter, planar)
    for N in (9, 1030):
        r = synth.make_rays(scene, 0, N, seed=8, margin=6)
        z = torch.sort(r["near"] + (r["far"] - r["near"]) * torch.rand(N, S, generator=g), -1)[0]
        pts = (r["rays_o"][:, None] + r["rays_d"][:, None] * z[..., None]).to(dev)
        grad = torch.randn(N, S, 3, generator=g).to(dev)
        logits, w = torch.randn(N, S, 10, generator=g).to(dev), (torch.rand(N, S, generator=g) * 0.05).to(dev)
        k1, k2 = torch.randn(N, 3, generator=g).to(dev), torch.randn(N, 25, 3, generator=g).to(dev)
        res = []
        for imgs in (planar, inter):
            lg, wd = logits.clone().requires_grad_(True), w.clone().requires_grad_(True)
            cp, pc, pmk = blend.blend_and_composite(hps, pts, lg, wd, grad, r["rays_d"].to(dev), imgs, src["w2cs"].to(dev),
                                                    src["intrinsics"].to(dev), src["query_c2w"].to(dev),
                                                    r["rays_uv"].clone().to(dev))
            ((cp * k1).sum() + (pc * k2).sum()).backward()
            res.append((cp.detach(), pc.detach(), pmk.detach(), lg.grad, wd.grad))
        for a, b in zip(*res):
            assert torch.equal(a, b)
        assert float(res[0][1].abs().max()) > 0


def test_ssim_patch_loss(dev):
    from neuraludf_amd.loss.loss import ColorLoss
    g = torch.Generator().manual_seed(6)
    for hps in (3, 5):
        npx = (2 * hps + 1) ** 2
        N = 77
        pred = torch.rand(N, npx, 3, generator=g)
        gt = (pred + 0.1 * torch.randn(N, npx, 3, generator=g)).clamp(0, 1)
        mask = torch.rand(N, 1, generator=g) > 0.2
        cb, col, rgb, cpix = [torch.rand(N, 3, generator=g) for _ in range(4)]
        pr = pred.clone().requires_grad_(True)
        ref = O.color_loss(1.0, 1.0, 0.5, 0.2, hps, cb, col, rgb, cpix, torch.ones(N, 1), pr, gt, mask.clone())
        ref["loss"].backward()
        crit = ColorLoss(1.0, 1.0, 0.5, 0.2, "l1", "ssim", hps)
        pd = pred.to(dev).requires_grad_(True)
        D = lambda t: t.to(dev)
        out = crit(D(cb), D(col), D(rgb), D(cpix), D(torch.ones(N, 1)), pd, D(gt), D(mask.clone()))
        for k in ref:
            assert abs(float(out[k]) - float(ref[k])) < 1e-5 * max(1.0, abs(float(ref[k]))), k
        out["loss"].backward()
        assert grel(pd.grad, pr.grad) < 1e-4


@pytest.mark.parametrize("kind", ["l1", "ssd", "ncc"])
def test_patch_loss_types_l1_ssd_ncc(dev, kind):
    """the patch-error types no shipped conf uses (loss/loss.py:66-73): nudf_patch_metric against the oracle, values
    of the trimmed loss and gradients w.r.t. the predicted patches; patch sizes below and above one wave of pixels."""
    from neuraludf_amd.loss.loss import ColorPatchLoss
    from neuraludf_amd.loss.patch_metric import patch_error
    g = torch.Generator().manual_seed(16)
    for hps in (1, 3, 5):
        npx = (2 * hps + 1) ** 2
        N = 53
        pred = torch.rand(N, npx, 3, generator=g)
        gt = (pred + 0.1 * torch.randn(N, npx, 3, generator=g)).clamp(0, 1)
        mask = torch.rand(N, 1, generator=g) > 0.2
        e_ref = O.patch_error(pred, gt, hps, kind)
        e = patch_error(pred.to(dev), gt.to(dev), hps, kind)
        assert rel(e, e_ref) < 1e-5, (kind, hps)
        pr = pred.clone().requires_grad_(True)
        O.patch_loss(pr, gt, mask.clone(), hps, kind=kind).backward()
        pd = pred.to(dev).requires_grad_(True)
        out = ColorPatchLoss(kind, hps)(pd, gt.to(dev), mask.clone().to(dev))
        out.backward()
        assert grel(pd.grad, pr.grad) < 1e-4, (kind, hps)
    with pytest.raises(ValueError):
        ColorPatchLoss("huber", 3)


def test_render_with_blending_matches_reference_fixture(dev):
    """full render (mix sampling + pixel + patch blending) on the committed reference fixture's inputs;
    compared on the rays whose samples did not take a different quantile bin."""
    from neuraludf_amd.models import fields
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    gold = np.load(os.path.join(HERE, "golden", "ref_mix_blend.npz"))
    mods = perturb_(build_modules(fields, seed=0))
    for m in mods.values():
        m.to(dev)
    rays = {k[4:]: torch.from_numpy(gold[k]).to(dev) for k in gold.files if k.startswith("ray_")}
    src = synth.make_source_views(synth.make_scene("tiny"), 0, 8)
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], n_samples=24,
                               n_importance=12, n_outside=0, up_sample_steps=3, perturb=1.0, upsampling_type="mix",
                               use_norm_grad_for_cosine=True, h_patch_size=3)
    D = lambda t: t.to(dev)
    out = rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=0.7,
                      perturb_overwrite=0, flip_saturation=0.9, color_maps=D(src["color_maps"]), w2cs=D(src["w2cs"]),
                      intrinsics=D(src["intrinsics"]), query_c2w=D(src["query_c2w"]), rays_uv=rays["rays_uv"].clone())
    zerr = (out["z_vals"].cpu() - torch.from_numpy(gold["out_z_vals"])).abs().max(dim=1)[0]
    good = zerr < 1e-4
    assert good.float().mean() > 0.7
    for k in ["color", "color_base", "color_pixel", "patch_colors", "patch_mask", "weights", "depth"]:
        a, b = out[k].detach().cpu()[good], torch.from_numpy(gold["out_" + k])[good]
        assert rel(a, b) < 5e-4, k          # random-noise source images: 1e-4-pixel tap differences show up


def test_patch_projector_interface_per_view_warps(dev):
    """PatchProjector.pixel_warp / .patch_warp (the reference's projector API, models/patch_projector.py:21-164):
    per-view colours and validity masks against the oracle's un-fused warps (which tests/test_oracle_vs_reference.py
    pins to the reference class), both image layouts, including the in-place uv rescale of patch_warp."""
    from neuraludf_amd.models.patch_projector import PatchProjector
    g = torch.Generator().manual_seed(14)
    scene = synth.make_scene("tiny")
    N, S, hps = 19, 23, 3
    r = synth.make_rays(scene, 0, N, seed=18, margin=10)
    src = synth.make_source_views(scene, 0, 8)
    imgs = _smooth_images(8, scene.H, scene.W)
    z = torch.sort(r["near"] + (r["far"] - r["near"]) * torch.rand(N, S, generator=g), -1)[0]
    pts = r["rays_o"][:, None] + r["rays_d"][:, None] * z[..., None]
    normals = torch.nn.functional.normalize(torch.randn(N, S, 3, generator=g), dim=-1)
    pcol, pmask = O.pixel_warp(pts, imgs, src["intrinsics"], src["w2cs"])
    tcol, tmask = O.patch_warp(pts, r["rays_uv"].clone(), normals, imgs, src["intrinsics"][0], src["intrinsics"],
                               src["query_c2w"], torch.inverse(src["w2cs"]), hps)
    D = lambda t: t.to(dev)
    proj = PatchProjector(hps)
    for hwc in (False, True):
        im = D(imgs)
        if hwc:      # the reference's dataset hands over an NCHW view of channel-interleaved memory
            im = im.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        col, msk = proj.pixel_warp(D(pts), im, D(src["intrinsics"]), D(src["w2cs"]))
        assert col.shape == (N, S, 8, 3) and msk.shape == (N, S, 8) and msk.dtype == torch.bool
        assert torch.equal(msk.cpu(), pmask.bool().reshape(N, S, 8))
        assert rel(col, pcol.reshape(N, S, 8, 3)) < 1e-5
        uv = D(r["rays_uv"].clone())
        pc, pm = proj.patch_warp(D(pts), uv, D(normals), im, D(src["intrinsics"][0]), D(src["intrinsics"]),
                                 D(src["query_c2w"]), D(torch.inverse(src["w2cs"])))
        assert pc.shape == (N, S, 8, 49, 3) and pm.shape == (N, S, 8, 49) and pm.dtype == torch.bool
        # uv was rescaled in place from (-1, 1) to pixels, like the reference does (patch_projector.py:75-76)
        assert rel(uv[:, 0], (r["rays_uv"][:, 0] + 1) / 2 * (scene.W - 1)) < 1e-6
        tm = tmask.bool().reshape(N, S, 8, 49)
        agree = (pm.cpu() == tm).float().mean()
        assert float(agree) > 0.999, float(agree)           # a patch pixel exactly on the validity border may flip
        both = (pm.cpu() & tm)
        assert float((pc.cpu() - tcol.reshape(N, S, 8, 49, 3)).abs()[both].max()) < 2e-4
