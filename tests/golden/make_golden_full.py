"""Full-size golden fixture: BASELINE config 2 (512 rays x (64 + 64 hierarchical samples in 4 rounds), P = 65 536
points per render_core) run through the REFERENCE code itself (imported read-only from /root/reference, build
container only):

    python tests/golden/make_golden_full.py

Writes tests/golden/ref_cfg2_full.npz (fp16-free, ~7 MB): rays, the reference's sample positions, every per-ray /
per-sample output the parity test compares, the loss, and the gradient of EVERY parameter that takes part (UDF,
colour, variance, beta nets).  Weights are rebuilt by the test from the seeds (see make_golden.py)."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from common import build_modules, perturb_, state_dicts, checksum, smooth_images  # noqa: E402
from refload import load_reference  # noqa: E402
from neuraludf_amd import synth  # noqa: E402

CASES = {
    # BASELINE config 2 (the headline): file ref_cfg2_full.npz
    "cfg2": dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0),
    # the shipped DTU conf's sampling (confs/udf_dtu_blending.conf): 64 + 50 in 5 rounds inside the sphere (114 samples:
    # ragged 64-sample chunks) + 32 outside samples through the background NeRF: file ref_dtu_shipped_full.npz
    "dtu_shipped": dict(n_samples=64, n_importance=50, n_outside=32, up_sample_steps=5, perturb=1.0),
    # BASELINE config 3's sampling and blending at 512 rays: mix up-sampling (64 + 64 in 3 rounds), normalised-gradient
    # cosines, pixel + patch blending over 8 source views with 7 x 7 patches, full ColorLoss: file ref_cfg3_blend_full.npz
    # BASELINE config 5's per-GPU shape in fp32: 1024 rays x (128 + 128 in 4 rounds) = 262 144 points per render_core (four
    # rounds of workgroups per chain launch, M = 262 144 in the weight-gradient GEMMs): file ref_cfg5_shape_full.npz
    "cfg5_shape": dict(n_samples=128, n_importance=128, n_outside=0, up_sample_steps=4, perturb=1.0),
    "cfg3_blend": dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=3, perturb=1.0, upsampling_type="mix",
                       use_norm_grad_for_cosine=True, h_patch_size=3),
    # BASELINE config 3 at its REAL geometry ("DeepFashion3D scan320-shaped"): 1024 rays, 8 source
    # views of 1024 x 1024 at f = 886.8 (pixel coordinates ~10x those of the "tiny" scene: projection, homography validity
    # tests and bilinear taps at x ~ 1e3), same pipeline as cfg3_blend: file ref_cfg3_garment_full.npz.  The 100 MB of
    # source images are not stored: common.smooth_images(8, 1024, 1024) regenerates them from the seed.
    "cfg3_garment": dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=3, perturb=1.0, upsampling_type="mix",
                         use_norm_grad_for_cosine=True, h_patch_size=3),
    # the EXACT inputs bench.py times and scores (`psnr_vs_ref`): dtu scene (1600 x 1200, f = 2892), camera 0, 512 rays
    # drawn with seed 1234 and no margin, seed-0 weights WITHOUT the perturbation, cos_anneal_ratio = flip_saturation = 1:
    # file ref_bench_cfg2_full.npz (per-ray outputs, sample positions, weights)
    "bench_cfg2": dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0),
}
SCENE = {"cfg3_garment": "garment", "bench_cfg2": "dtu"}
RAYS = {"cfg5_shape": 1024, "cfg3_garment": 1024}
# garment scene: fov 60 deg from radius 2.2 -- the central 424 x 424 pixels see the object (elsewhere weight_sum ~ 0 and
# the blending terms would be compared on empty rays)
MARGIN = {"cfg3_garment": 300}
KW = CASES["cfg2"]
N_RAYS = 512
KEYS = ["z_vals", "color", "color_base", "weights", "depth", "udf", "gradients", "normals", "weight_sum",
        "gradient_error", "gradient_error_near_surface", "sparse_error"]


def loss_of(out, rays):
    # sparse_error = mean sum exp(-25000 udf) is left out of the differentiated loss: its gradient amplifies an fp32
    # ulp of udf into percents, for the fp32 reference as much as for any other fp32 evaluation (its value is compared,
    # its backward is checked on identical inputs in test_gpu_kernels.py::test_composite_stagewise)
    return ((out["color"] - rays["true_rgb"]).abs().mean() + 0.5 * (out["color_base"] - rays["true_rgb"]).abs().mean()
            + 0.1 * out["gradient_error"] + 0.01 * out["gradient_error_near_surface"])


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    kw = CASES[case]
    rf, rr, rl = load_reference()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    mods = build_modules(rf, seed=0)
    if case != "bench_cfg2":
        mods = perturb_(mods)
    sums = {k: checksum(v) for k, v in state_dicts(mods).items()}
    scene = synth.make_scene(SCENE.get(case, "tiny"))
    n_rays = RAYS.get(case, N_RAYS)
    if case == "bench_cfg2":
        rays = synth.make_rays(scene, 0, n_rays, seed=1234)            # bench.py's own call
    else:
        rays = synth.make_rays(scene, 0, n_rays, seed=11, margin=MARGIN.get(case, 6))
    r = rr.UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **kw)
    t0 = time.time()
    bkw, keys = {}, list(KEYS)
    blend = "h_patch_size" in kw
    if blend:
        src = synth.make_source_views(scene, 0, 8)
        src["color_maps"] = smooth_images(8, scene.H, scene.W)       # band-limited: taps insensitive to 1e-4-pixel noise
        bkw = dict(color_maps=src["color_maps"], w2cs=src["w2cs"], intrinsics=src["intrinsics"],
                   query_c2w=src["query_c2w"], rays_uv=rays["rays_uv"].clone())
        keys += ["color_pixel", "patch_colors", "patch_mask"]
    sched = dict(cos_anneal_ratio=1.0, flip_saturation=1.0) if case == "bench_cfg2" else dict(cos_anneal_ratio=0.7,
                                                                                              flip_saturation=0.9)
    out = r.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], perturb_overwrite=0, **sched, **bkw)
    loss = loss_of(out, rays)
    extra = {}
    if blend:
        g = torch.Generator().manual_seed(3)
        gt_patch = torch.rand(n_rays, 49, 3, generator=g)
        pmask = (out["patch_mask"].detach() > 0.3).reshape(-1, 1)
        crit = rl.ColorLoss(color_base_weight=1.0, color_weight=1.0, color_pixel_weight=0.5, color_patch_weight=0.2,
                            pixel_loss_type="l1", patch_loss_type="ssim", h_patch_size=3)
        cl = crit(out["color_base"], out["color"], rays["true_rgb"], out["color_pixel"], rays["mask"],
                  out["patch_colors"], gt_patch, pmask.clone())
        loss = loss + cl["loss"]
        extra = {"gt_patch": gt_patch.numpy(), "pmask": pmask.numpy()}
        extra.update({"closs_" + k: np.float64(float(v)) for k, v in cl.items()})
    loss.backward()
    print("reference fwd+bwd %.1f s, loss %.6f" % (time.time() - t0, loss.item()))
    data = {"ray_" + k: v.numpy() for k, v in rays.items()}
    if case == "cfg5_shape":      # keep the file small: per-ray outputs, sample positions and weights only
        keys = ["z_vals", "color", "color_base", "weights", "depth", "weight_sum", "gradient_error",
                "gradient_error_near_surface", "sparse_error"]
    if case == "bench_cfg2":
        keys = ["z_vals", "color", "color_base", "weights", "depth", "weight_sum"]
    if case == "cfg3_garment":    # per-ray outputs, sample positions, weights and udf (no [N,S,3] arrays)
        keys = ["z_vals", "color", "color_base", "weights", "depth", "udf", "normals", "weight_sum", "gradient_error",
                "gradient_error_near_surface", "sparse_error", "color_pixel", "patch_colors", "patch_mask"]
    data.update({"out_" + k: out[k].detach().numpy().astype(np.float32) for k in keys})
    data.update(extra)
    data["loss"] = np.float64(loss.item())
    n = 0
    for net in ("udf", "color", "var", "beta") + (("nerf",) if kw["n_outside"] > 0 else ()):
        for pn, p in mods[net].named_parameters():
            if p.grad is not None:
                data[f"grad_{net}_{pn}"] = p.grad.numpy()
                n += p.grad.numel()
    for k, v in sums.items():
        data["wsum_" + k] = np.float64(v)
    path = os.path.join(HERE, "ref_%s_full.npz" % case)
    np.savez_compressed(path, **data)
    print("wrote", path, "%.1f MB" % (os.path.getsize(path) / 1e6), "param-grad floats", n)


if __name__ == "__main__":
    main()
