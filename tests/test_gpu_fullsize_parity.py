"""BASELINE config 2 at FULL size (512 rays x (64 + 64 hierarchical samples in 4 rounds), P = 65 536 points per
render_core) against a fixture produced by the REFERENCE code itself (tests/golden/make_golden_full.py ran
/root/reference/models/udf_renderer_blending.py:586-755 on CPU and committed outputs + every parameter gradient):

  * render_core on the reference's own sample positions: values <= 1e-4, every parameter gradient <= 1e-3 -- with the
    64-point chain tiles, the grouped weight-gradient GEMMs at M = 65 536 and (second case) the wave-private chain kernel;
  * the end-to-end render: fraction of rays whose 128 sample positions match the reference's, and WHERE the others
    diverge first -- every up-sampling round is replayed from the oracle's trace (oracle == reference, pinned by
    tests/test_oracle_*.py), once with the oracle's udf (isolates the up-sampling kernel's own arithmetic) and once with
    the HIP MLP's udf at the oracle's positions (adds the MLP's fp32 summation order);
  * the 16-bit operand mode of BASELINE config 5 against the ORACLE (not against the fp32 HIP path): colour PSNR on
    perturbed weights with the hierarchical schedule on."""
import math
import os

import numpy as np
import pytest
import torch

from common import (build_modules, perturb_, state_dicts, checksum, oracle_nets, weights_vs_reference_up_to_ties,
                    param_grads_vs_reference, grad_report)
from oracle import udf_oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "ref_cfg2_full.npz")
KW = dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0)
VTOL, GTOL = 1e-4, 1e-3


def rel(a, b):
    a = torch.as_tensor(a).detach().float().cpu()
    b = torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1.0))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def setup(dev):
    from neuraludf_amd.models import fields
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    fx = dict(np.load(FIX))
    mods = perturb_(build_modules(fields, seed=0))
    sds = state_dicts(mods)
    for k, v in sds.items():      # the fixture's weights are the seeds' weights: prove it
        assert abs(checksum(v) - float(fx["wsum_" + k])) < 1e-6 * max(1.0, abs(float(fx["wsum_" + k]))), k
    for m in mods.values():
        m.to(dev)
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **KW)
    rays = {k[4:]: torch.from_numpy(v).to(dev) for k, v in fx.items() if k.startswith("ray_")}
    return fx, mods, sds, rend, rays


def _loss(out, rgb):
    return ((out["color"] - rgb).abs().mean() + 0.5 * (out["color_base"] - rgb).abs().mean()
            + 0.1 * out["gradient_error"] + 0.01 * out["gradient_error_near_surface"])


@pytest.mark.parametrize("tile", [0, 128])
def test_cfg2_render_core_and_all_parameter_gradients_vs_reference(dev, setup, tile):
    """tile 0: the default path (64-point workgroup tiles at P = 65 536); tile 128: the wave-private chain kernel."""
    from neuraludf_amd import mlp
    fx, mods, _, rend, rays = setup
    z_ref = torch.from_numpy(fx["out_z_vals"]).to(dev)
    assert z_ref.shape == (512, 128)
    sd = float(((rays["far"] - rays["near"]) / KW["n_samples"]).mean())
    sdd = torch.tensor([sd], device=dev)
    for m in mods.values():
        m.zero_grad()
    old = mlp.CHAIN_TILE
    mlp.CHAIN_TILE = tile
    try:
        out = rend.render_core(rays["rays_o"], rays["rays_d"], z_ref, sdd, 0.7, None, None, None, None, 0.9, s_nominal=128)
        loss = _loss(out, rays["true_rgb"])
        loss.backward()
        torch.cuda.synchronize()
    finally:
        mlp.CHAIN_TILE = old
    for k in ["color", "color_base", "weights", "depth", "udf", "gradients", "normals", "weight_sum", "gradient_error",
              "gradient_error_near_surface", "sparse_error"]:
        # sparse_error = mean sum exp(-25000 udf) amplifies an fp32 ulp of udf to ~1e-3 relative
        assert rel(out[k], fx["out_" + k]) < (2e-3 if k == "sparse_error" else VTOL), k
    # ... which is why its INPUT is held to an absolute bound beside it (measured 1.6e-6 = a few fp32 ulp of a udf ~ 1)
    assert float((out["udf"].detach().cpu() - torch.from_numpy(fx["out_udf"])).abs().max()) < 4e-6
    assert abs(float(loss) - float(fx["loss"])) < 1e-5 * max(1.0, abs(float(fx["loss"])))
    rep = param_grads_vs_reference(mods, fx, "cfg2", gtol=GTOL)
    assert rep["n"] >= 50
    print(f"cfg2 full size (tile {tile}): {grad_report(rep)}")


def test_shipped_dtu_sampling_with_background_nerf_vs_reference(dev):
    """the shipped DTU conf's sampling at full size -- 512 rays x (64 + 50 in 5 rounds) inside the sphere (114 samples:
    ragged 64-sample chunks in the composite) + 32 outside samples through the background NeRF -- on the reference's own
    inside sample positions (fixture tests/golden/ref_dtu_shipped_full.npz, written by make_golden_full.py dtu_shipped
    from the reference code): every output <= 1e-4, every one of the 1 291 482 parameter-gradient floats (UDF, colour,
    variance, beta AND NeRF) <= 1e-3."""
    from neuraludf_amd.models import fields
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    fx = dict(np.load(os.path.join(HERE, "golden", "ref_dtu_shipped_full.npz")))
    kw = dict(n_samples=64, n_importance=50, n_outside=32, up_sample_steps=5, perturb=1.0)
    mods = perturb_(build_modules(fields, seed=0))
    for k, v in state_dicts(mods).items():
        assert abs(checksum(v) - float(fx["wsum_" + k])) < 1e-6 * max(1.0, abs(float(fx["wsum_" + k]))), k
    for m in mods.values():
        m.to(dev)
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **kw)
    rays = {k[4:]: torch.from_numpy(v).to(dev) for k, v in fx.items() if k.startswith("ray_")}
    z_ref = torch.from_numpy(fx["out_z_vals"]).to(dev)
    assert z_ref.shape == (512, 114)
    out = rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=0.7, perturb_overwrite=0,
                      flip_saturation=0.9, z_vals_override=z_ref)
    loss = _loss(out, rays["true_rgb"])
    loss.backward()
    torch.cuda.synchronize()
    for k in ["color", "color_base", "weights", "depth", "udf", "gradients", "normals", "weight_sum", "gradient_error",
              "gradient_error_near_surface", "sparse_error"]:
        assert rel(out[k], fx["out_" + k]) < (2e-3 if k == "sparse_error" else VTOL), k
    assert float((out["udf"].detach().cpu() - torch.from_numpy(fx["out_udf"])).abs().max()) < 4e-6   # sparse_error's input
    assert abs(float(loss) - float(fx["loss"])) < 1e-5 * max(1.0, abs(float(fx["loss"])))
    rep = param_grads_vs_reference(mods, fx, "dtu_shipped", nets=("udf", "color", "var", "beta", "nerf"), gtol=GTOL)
    assert rep["n"] >= 80 and rep["floats"] == 1291482
    print(f"shipped DTU sampling, full size: {grad_report(rep)}")


def test_cfg5_shape_fp32_vs_reference(dev):
    """BASELINE config 5's per-GPU shape in fp32 -- 1024 rays x (128 + 128 in 4 rounds), 262 144 points per render_core:
    four rounds of workgroups per chain launch, blocked saved state, M = 262 144 in the weight-gradient GEMMs -- on the
    reference's own sample positions (fixture ref_cfg5_shape_full.npz, make_golden_full.py cfg5_shape)."""
    from neuraludf_amd.models import fields
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    fx = dict(np.load(os.path.join(HERE, "golden", "ref_cfg5_shape_full.npz")))
    kw = dict(n_samples=128, n_importance=128, n_outside=0, up_sample_steps=4, perturb=1.0)
    mods = perturb_(build_modules(fields, seed=0))
    for k, v in state_dicts(mods).items():
        assert abs(checksum(v) - float(fx["wsum_" + k])) < 1e-6 * max(1.0, abs(float(fx["wsum_" + k]))), k
    for m in mods.values():
        m.to(dev)
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **kw)
    rays = {k[4:]: torch.from_numpy(v).to(dev) for k, v in fx.items() if k.startswith("ray_")}
    z_ref = torch.from_numpy(fx["out_z_vals"]).to(dev)
    assert z_ref.shape == (1024, 256)
    out = rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=0.7, perturb_overwrite=0,
                      flip_saturation=0.9, z_vals_override=z_ref)
    loss = _loss(out, rays["true_rgb"])
    loss.backward()
    torch.cuda.synchronize()
    # `vis_mask = true_cos < 0.01` (udf_renderer_blending.py:399-405) is a hard selection: where true_cos ties with the
    # threshold to an ulp fp32 implementations may select differently, and that ray's LATER weights shift by 1e-3 (its
    # colour by 5e-6).  At most 3 rays may hold such a tie, and their weights in front of it are compared like all others.
    again = lambda: rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=0.7,
                                perturb_overwrite=0, flip_saturation=0.9, z_vals_override=z_ref)
    ok, ties = weights_vs_reference_up_to_ties(rend, again, out["weights"], fx["out_weights"], 256)
    for k in ["color", "color_base", "weight_sum", "gradient_error", "gradient_error_near_surface", "sparse_error"]:
        assert rel(out[k], fx["out_" + k]) < (2e-3 if k == "sparse_error" else VTOL), k
    assert rel(out["depth"].detach().cpu()[ok], torch.from_numpy(fx["out_depth"])[ok]) < VTOL
    assert abs(float(loss) - float(fx["loss"])) < 1e-5 * max(1.0, abs(float(fx["loss"])))
    rep = param_grads_vs_reference(mods, fx, "cfg5_shape", gtol=GTOL, absent_must_be_zero=False)
    assert rep["n"] >= 50
    print(f"cfg5 shape (fp32, P = 262 144): rays with a true_cos tie (ray, tie sample, first differing weight, true_cos) {ties} of 1024; "
          f"{grad_report(rep)}")


def _cfg3_blending_vs_reference(dev, fixture, scene_kind, n_rays, case):
    from common import smooth_images
    from neuraludf_amd import synth
    from neuraludf_amd.loss.loss import ColorLoss
    from neuraludf_amd.models import fields
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    fx = dict(np.load(os.path.join(HERE, "golden", fixture)))
    kw = dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=3, perturb=1.0, upsampling_type="mix",
              use_norm_grad_for_cosine=True, h_patch_size=3)
    mods = perturb_(build_modules(fields, seed=0))
    for k, v in state_dicts(mods).items():
        assert abs(checksum(v) - float(fx["wsum_" + k])) < 1e-6 * max(1.0, abs(float(fx["wsum_" + k]))), k
    for m in mods.values():
        m.to(dev)
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **kw)
    rays = {k[4:]: torch.from_numpy(v).to(dev) for k, v in fx.items() if k.startswith("ray_")}
    assert rays["rays_o"].shape[0] == n_rays
    scene = synth.make_scene(scene_kind)
    src = synth.make_source_views(scene, 0, 8)
    D = lambda t: t.to(dev)
    z_ref = torch.from_numpy(fx["out_z_vals"]).to(dev)
    out = rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=0.7, perturb_overwrite=0,
                      flip_saturation=0.9, color_maps=D(smooth_images(8, scene.H, scene.W)), w2cs=D(src["w2cs"]),
                      intrinsics=D(src["intrinsics"]), query_c2w=D(src["query_c2w"]), rays_uv=rays["rays_uv"].clone(),
                      z_vals_override=z_ref)
    crit = ColorLoss(1.0, 1.0, 0.5, 0.2, "l1", "ssim", 3)
    cl = crit(out["color_base"], out["color"], rays["true_rgb"], out["color_pixel"], rays["mask"], out["patch_colors"],
              D(torch.from_numpy(fx["gt_patch"])), D(torch.from_numpy(fx["pmask"])))
    loss = _loss(out, rays["true_rgb"]) + cl["loss"]
    loss.backward()
    torch.cuda.synchronize()
    pm_ref = torch.from_numpy(fx["out_patch_mask"])                 # per-ray weight of the valid patch samples
    agree = float(((out["patch_mask"].detach().cpu().reshape(pm_ref.shape) - pm_ref).abs() < 1e-3).float().mean())
    assert agree > 0.995, agree                        # a sample exactly on a view's validity border may flip
    # The alpha of a sample is a hard selection between two candidates (udf_renderer_blending.py:414-423): where they tie to
    # an ulp, fp32 implementations may select differently and that ray's later weights shift by ~1e-3 (its colour by
    # ~1e-5) -- see test_cfg5_shape_fp32_vs_reference.  Such rays are counted (<= 3), per-sample arrays compared on the rest.
    again = lambda: rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=0.7,
                                perturb_overwrite=0, flip_saturation=0.9, z_vals_override=z_ref)
    ok, ties = weights_vs_reference_up_to_ties(rend, again, out["weights"], fx["out_weights"], z_ref.shape[1])
    n_flip = int((~ok).sum())
    worst_v = ("", 0.0)
    for k in ["color", "color_base", "weights", "depth", "udf", "gradients", "normals", "weight_sum", "gradient_error",
              "gradient_error_near_surface", "color_pixel", "patch_colors"]:
        if "out_" + k not in fx:
            continue          # the 1024-ray fixture holds no [N, S, 3] arrays
        a, b = out[k].detach().cpu().reshape(fx["out_" + k].shape), torch.from_numpy(fx["out_" + k])
        if k in ("weights", "depth", "normals", "weight_sum") and n_flip:
            a, b = a[ok], b[ok]
        r = rel(a, b)
        if r > worst_v[1]:
            worst_v = (k, r)
        assert r < (1.5e-4 if k in ("color_pixel", "patch_colors") else VTOL), (k, r)     # worst measured 7e-5
    for k in ("loss", "color_base_loss", "color_loss", "color_pixel_loss", "color_patch_loss"):
        assert abs(float(cl[k]) - float(fx["closs_" + k])) < 2e-4 * max(1.0, abs(float(fx["closs_" + k]))), k
    # the trimmed patch loss (loss/loss.py:79-84) drops / keeps WHOLE rays: one ray on the other side of the trimming
    # quantile moves the colour network's gradients by ~1e-3 of their norm -- the bar of this case is 2e-3
    rep = param_grads_vs_reference(mods, fx, case, gtol=2e-3, absent_must_be_zero=False)
    assert rep["n"] >= 50
    print(f"cfg3 mix + blending, {n_rays} rays, scene '{scene_kind}' ({scene.H} x {scene.W}): rays with a flipped alpha selection "
          f"{n_flip}, patch-mask agreement {agree:.4f}, worst value {worst_v[0]} {worst_v[1]:.2e}; {grad_report(rep)}")


def test_cfg3_mix_sampling_and_blending_vs_reference(dev):
    """BASELINE config 3's pipeline at 512 rays x 128 samples -- mix up-sampling geometry, normalised-gradient cosines,
    pixel + patch blending over 8 source views with 7 x 7 patches and the full ColorLoss (L1 terms + trimmed SSIM patch
    loss) -- on the reference's own sample positions (fixture ref_cfg3_blend_full.npz, make_golden_full.py cfg3_blend):
    outputs incl. the blended pixel / patch colours, the loss terms, and all parameter gradients."""
    _cfg3_blending_vs_reference(dev, "ref_cfg3_blend_full.npz", "tiny", 512, "cfg3_blend")


def test_cfg3_garment_geometry_1024_rays_vs_reference(dev):
    """BASELINE config 3 at its REAL geometry: 1024 rays, 8 source views of 1024 x 1024 at f = 886.8
    -- projection / normalisation (projector_utils.py:8-48), the homography validity tests (patch_projector.py:100-131)
    and the bilinear taps at pixel coordinates ~1e3 -- against the reference's own run (fixture ref_cfg3_garment_full.npz,
    make_golden_full.py cfg3_garment; the 100 MB of source images are regenerated from the seed), same tolerances as the
    512-ray / 96 x 128 case above."""
    _cfg3_blending_vs_reference(dev, "ref_cfg3_garment_full.npz", "garment", 1024, "cfg3_garment")


def test_cfg2_end_to_end_matching_rays_and_first_divergence(dev, setup):
    from neuraludf_amd.models import udf_renderer_blending as urb
    fx, mods, sds, rend, rays = setup
    N = 512
    z_ref = torch.from_numpy(fx["out_z_vals"])
    default_flags = urb.UPSAMPLE_FLAGS

    def end_to_end():
        with torch.no_grad():
            out = rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=0.7,
                              perturb_overwrite=0, flip_saturation=0.9)
        good = (out["z_vals"].cpu() - z_ref).abs().max(dim=1)[0] < 1e-4
        mse = float(((out["color"].cpu() - torch.from_numpy(fx["out_color"])) ** 2).mean())
        return out, good, 20.0 * math.log10(1.0 / math.sqrt(mse + 1e-20))

    out, good, psnr = end_to_end()
    frac = float(good.float().mean())
    for k in ["color", "color_base", "depth", "weight_sum"]:
        assert rel(out[k][good.to(dev)], torch.from_numpy(fx["out_" + k])[good]) < 1e-4, k
    wmax = float((out["weights"].cpu()[good] - torch.from_numpy(fx["out_weights"])[good]).abs().max())

    # ---- localisation: replay every round of the oracle's trace, for every arithmetic variant of the up-sampling
    # kernel: scans (wave-parallel fp32 tree | serial double accumulator = torch CPU order) x contraction (on | off) ----
    cpu = {k: v.cpu() for k, v in rays.items()}
    cfg = O.RenderCfg(**{k: v for k, v in KW.items() if k != "perturb"})
    on = oracle_nets(sds)
    trace = []
    z0, _, sd = O.coarse_z(cfg, cpu["near"], cpu["far"], N)
    with torch.no_grad():
        z_oracle = O.importance_sample(on, cfg, cpu["rays_o"], cpu["rays_d"], z0, sd, trace)
    # The fixture was produced on the build container's CPU; the same fp32 PyTorch code on THIS host's CPU may already
    # take other quantile bins on some rays (different BLAS kernels / summation order): reported, and the replay
    # below uses this host's trace, which is self-consistent.
    host_ok = int(((z_oracle - z_ref).abs().max(dim=1)[0] < 1e-4).sum())
    sdd = torch.tensor([sd], device=dev)
    variants = [("tree scans, contraction on (rounds 1-2)", 0), ("serial double scans, contraction on", 512),
                ("tree scans, contraction off", 1024), ("serial double scans, contraction off", 512 | 1024),
                ("serial double scans, contraction off, Sleef sigmoid", 512 | 1024 | 2048),
                ("serial double scans, contraction off, Sleef sigmoid, correctly rounded exp", 512 | 1024 | 2048 | 4096)]
    report = [f"cfg2 512 x 128 end to end vs the reference (default flags {default_flags}): {int(good.sum())} / {N} rays with "
              f"identical samples ({100 * frac:.1f} %), colour PSNR over all rays {psnr:.1f} dB, max |dweights| on those rays {wmax:.2e}",
              f"the CPU oracle re-run on this host reproduces the fixture's samples on {host_ok} / {N} rays"]
    moved_default = None
    try:
        for name, flags in variants:
            urb.UPSAMPLE_FLAGS = flags
            lines, worst = [], 0.0
            for i, t in enumerate(trace):
                k = t["z_new"].shape[1]
                mode = 0 if t["kind"] == "unbias" else 1
                zt, ut = t["z"].to(dev), t["udf"].to(dev)
                with torch.no_grad():
                    z_a, _ = rend._upsample(rays["rays_o"], rays["rays_d"], zt, ut, sdd, k, mode, t["inv_s"], t["beta"], t["gamma"])
                    u_hip = rend._udf_at(rays["rays_o"], rays["rays_d"], zt, sdd)
                    z_b, _ = rend._upsample(rays["rays_o"], rays["rays_d"], zt, u_hip, sdd, k, mode, t["inv_s"], t["beta"], t["gamma"])
                da = (z_a.cpu() - t["z_new"]).abs().max(dim=1)[0]
                bad_a, exact_a = int((da > 1e-4).sum()), int((da == 0).sum())
                bad_b = int(((z_b.cpu() - t["z_new"]).abs().max(dim=1)[0] > 1e-4).sum())
                du = float((u_hip.cpu() - t["udf"]).abs().max())
                worst = max(worst, float(da.max()))
                lines.append(f"   round {i}: {zt.shape[1]:3d} -> +{k} samples | oracle udf in: {bad_a:3d} / {N} rays moved > 1e-4, {exact_a:3d} "
                             f"bit-identical | HIP MLP's udf at the oracle's positions: {bad_b:3d} moved | max |udf_hip - udf_oracle| {du:.2e}")
                if flags == default_flags and i == 0:
                    moved_default = bad_a
            _, g2, p2 = end_to_end()
            report.append(f"{name} (flags {flags}): end to end {int(g2.sum())} / {N} rays identical, PSNR {p2:.1f} dB, worst |dz| with the "
                          f"oracle's udf {worst:.2e}")
            report += lines
    finally:
        urb.UPSAMPLE_FLAGS = default_flags
    print("\n".join(report))
    try:
        os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
        with open(os.path.join(os.path.dirname(HERE), "gpurun_out", "parity_localisation.txt"), "w") as f:
            f.write("\n".join(report) + "\n")
    except OSError:
        pass
    # measured (round 4, the reference's transcendentals in the up-sampling kernel): 467-477 / 512 rays, 93.0 dB; the bounds
    # leave a 2x margin on the number of moved rays / 5 dB so that a regression of the sampling chain fails here
    assert frac > 0.88, frac
    assert psnr > 88.0, psnr


def _psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 20.0 * math.log10(1.0 / math.sqrt(max(mse, 1e-20)))


def test_mixed16_vs_oracle_psnr_hierarchical(dev, setup):
    """16-bit MFMA operands (BASELINE config 5 mode) against the fp32 ORACLE: perturbed weights, hierarchical schedule,
    (i) on the oracle's own sample positions and (ii) end to end.  Bar: colour PSNR >= 60 dB."""
    from neuraludf_amd import mlp
    fx, mods, sds, rend, rays = setup
    n = 256
    sub = {k: v[:n].contiguous() for k, v in rays.items()}
    cpu = {k: v.cpu() for k, v in sub.items()}
    cfg = O.RenderCfg(**{k: v for k, v in KW.items() if k != "perturb"})
    on = oracle_nets(sds)
    with torch.no_grad():
        ref = O.render(on, cfg, cpu["rays_o"], cpu["rays_d"], cpu["near"], cpu["far"], cos_anneal_ratio=0.7, flip_saturation=0.9)
    sd = float(((cpu["far"] - cpu["near"]) / KW["n_samples"]).mean())
    sdd = torch.tensor([sd], device=dev)
    base = mlp.PRECISION          # the library default (bf16x3), or whatever NUDF_PRECISION selects
    try:
        mlp.set_precision("mixed16")
        with torch.no_grad():
            core = rend.render_core(sub["rays_o"], sub["rays_d"], ref["z_vals"].to(dev), sdd, 0.7, None, None, None, None, 0.9,
                                    s_nominal=128)
            e2e = rend.render(sub["rays_o"], sub["rays_d"], sub["near"], sub["far"], cos_anneal_ratio=0.7, perturb_overwrite=0,
                              flip_saturation=0.9)
    finally:
        mlp.set_precision(base)
    p_core = _psnr(core["color"].cpu(), ref["color"])
    p_base = _psnr(core["color_base"].cpu(), ref["color_base"])
    p_e2e = _psnr(e2e["color"].cpu(), ref["color"])
    w_err = float((core["weights"].cpu() - ref["weights"]).abs().max())
    print(f"mixed16 vs oracle (256 rays x 64+64/4, perturbed weights): colour PSNR {p_core:.1f} dB on the oracle's samples "
          f"(colour_base {p_base:.1f} dB, max |dweights| {w_err:.2e}), {p_e2e:.1f} dB end to end")
    assert p_core >= 60.0, p_core
    assert p_base >= 60.0, p_base
    assert p_e2e >= 50.0, p_e2e


def test_mixed16_at_cfg5_shape_vs_reference(dev):
    """The 16-bit operand mode at BASELINE config 5's OWN per-GPU shape -- 1024 rays x (128 + 128 in 4 rounds), 262 144
    points per render_core -- against the REFERENCE's fp32 outputs (fixture ref_cfg5_shape_full.npz, the one the fp32 test
    above uses), on the reference's sample positions.  Bar: colour / colour_base PSNR >= 60 dB;
    max |d weights| is reported."""
    from neuraludf_amd import mlp
    from neuraludf_amd.models import fields
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    fx = dict(np.load(os.path.join(HERE, "golden", "ref_cfg5_shape_full.npz")))
    kw = dict(n_samples=128, n_importance=128, n_outside=0, up_sample_steps=4, perturb=1.0)
    mods = perturb_(build_modules(fields, seed=0))
    for m in mods.values():
        m.to(dev)
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **kw)
    rays = {k[4:]: torch.from_numpy(v).to(dev) for k, v in fx.items() if k.startswith("ray_")}
    z_ref = torch.from_numpy(fx["out_z_vals"]).to(dev)
    assert z_ref.shape == (1024, 256)
    base = mlp.PRECISION
    try:
        mlp.set_precision("mixed16")
        out = rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=0.7,
                          perturb_overwrite=0, flip_saturation=0.9, z_vals_override=z_ref)
        loss = _loss(out, rays["true_rgb"])
        loss.backward()          # the 16-bit backward sweeps + bf16 saved state at M = 262 144
        torch.cuda.synchronize()
    finally:
        mlp.set_precision(base)
    p_col = _psnr(out["color"].detach().cpu(), torch.from_numpy(fx["out_color"]))
    p_base = _psnr(out["color_base"].detach().cpu(), torch.from_numpy(fx["out_color_base"]))
    w_err = float((out["weights"].detach().cpu() - torch.from_numpy(fx["out_weights"])).abs().max())
    ws_err = float((out["weight_sum"].detach().cpu() - torch.from_numpy(fx["out_weight_sum"])).abs().max())
    # parameter gradients: cosine against the reference's fp32 gradients (the mode's own bar, tests/test_gpu_mixed16.py)
    cos_min, cos_key = 1.0, ""
    for net in ("udf", "color"):
        for pn, p in mods[net].named_parameters():
            key = f"grad_{net}_{pn}"
            if key not in fx or p.grad is None:
                continue
            a, b = p.grad.detach().cpu().double().flatten(), torch.from_numpy(fx[key]).double().flatten()
            if float(b.norm()) < 1e-12:
                continue
            c = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
            if c < cos_min:
                cos_min, cos_key = c, key
    print(f"mixed16 at 1024 x 256 vs the reference (its samples): colour PSNR {p_col:.1f} dB, colour_base {p_base:.1f} dB, "
          f"max |dweights| {w_err:.2e}, max |dweight_sum| {ws_err:.2e}, worst parameter-gradient cosine {cos_min:.5f} ({cos_key}), "
          f"loss {float(loss):.6f} vs {float(fx['loss']):.6f}")
    assert p_col >= 60.0, p_col
    assert p_base >= 60.0, p_base
    assert cos_min > 0.99, (cos_key, cos_min)
    assert abs(float(loss) - float(fx["loss"])) < 2e-3


def test_bench_inputs_vs_reference_fixture(dev):
    """the EXACT inputs bench.py times and scores -- dtu scene (1600 x 1200, f = 2892), camera 0, 512 rays drawn with seed
    1234, seed-0 weights without perturbation, cos_anneal_ratio = flip_saturation = 1 -- against the reference's own CPU
    run on them (fixture ref_bench_cfg2_full.npz, make_golden_full.py bench_cfg2): render_core on the reference's samples
    (values 1e-4, all parameter gradients 1e-3) and the end-to-end render (rays with identical samples, colours on them)."""
    from neuraludf_amd import synth
    from neuraludf_amd.models import fields
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    fx = dict(np.load(os.path.join(HERE, "golden", "ref_bench_cfg2_full.npz")))
    mods = build_modules(fields, seed=0)
    for k, v in state_dicts(mods).items():
        assert abs(checksum(v) - float(fx["wsum_" + k])) < 1e-6 * max(1.0, abs(float(fx["wsum_" + k]))), k
    for m in mods.values():
        m.to(dev)
    rays_cpu = synth.make_rays(synth.make_scene("dtu"), 0, 512, seed=1234)               # bench.py's own call
    # (another host CPU builds the same rays to within an ulp of the directions: the fixture's stored rays are rendered)
    assert float(np.abs(fx["ray_rays_d"] - rays_cpu["rays_d"].numpy()).max()) < 1e-6
    assert np.array_equal(fx["ray_pixels"], rays_cpu["pixels"].numpy())
    rays = {k[4:]: torch.from_numpy(v).to(dev) for k, v in fx.items() if k.startswith("ray_")}
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **KW)
    z_ref = torch.from_numpy(fx["out_z_vals"]).to(dev)
    out = rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=1.0, perturb_overwrite=0,
                      flip_saturation=1.0, z_vals_override=z_ref)
    loss = _loss(out, rays["true_rgb"])
    loss.backward()
    torch.cuda.synchronize()
    again = lambda: rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=1.0,
                                perturb_overwrite=0, flip_saturation=1.0, z_vals_override=z_ref)
    ok, ties = weights_vs_reference_up_to_ties(rend, again, out["weights"], fx["out_weights"], 128)   # see the cfg5 test
    for k in ["color", "color_base", "weight_sum"]:
        assert rel(out[k], fx["out_" + k]) < VTOL, k
    assert rel(out["depth"].detach().cpu()[ok], torch.from_numpy(fx["out_depth"])[ok]) < VTOL
    assert abs(float(loss) - float(fx["loss"])) < 1e-5 * max(1.0, abs(float(fx["loss"])))
    rep = param_grads_vs_reference(mods, fx, "bench_cfg2", gtol=GTOL, absent_must_be_zero=False)
    assert rep["n"] >= 50
    with torch.no_grad():
        e2e = rend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=1.0, perturb_overwrite=0,
                          flip_saturation=1.0)
    zerr = (e2e["z_vals"].cpu() - torch.from_numpy(fx["out_z_vals"])).abs().max(dim=1)[0]
    good, close = zerr < 1e-4, zerr < 1e-5
    assert float(good.float().mean()) > 0.88, int(good.sum())          # measured 477 / 512
    assert float(close.float().mean()) > 0.75, int(close.sum())        # measured 419 / 512
    # seed-0 weights are the geometric initialisation, a sharp sphere: 1e-4 of sample position is 1.4e-4 of colour there, so
    # the 1e-4 colour bound is held on the rays whose samples agree to 1e-5 (and 1.5x looser on those within 1e-4)
    c_close = rel(e2e["color"][close.to(dev)], torch.from_numpy(fx["out_color"])[close])
    c_good = rel(e2e["color"][good.to(dev)], torch.from_numpy(fx["out_color"])[good])
    assert c_close < VTOL, c_close
    assert c_good < 1.5 * VTOL, c_good
    print(f"bench inputs vs the reference: {grad_report(rep)}; rays with a true_cos "
          f"tie {ties}; end to end {int(close.sum())} / 512 rays with samples within 1e-5 (colour {c_close:.1e}), "
          f"{int(good.sum())} within 1e-4 (colour {c_good:.1e})")
