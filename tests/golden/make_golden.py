"""Generate the golden fixtures by running the REFERENCE code itself (imported read-only from
/root/reference, build container only):

    python tests/golden/make_golden.py

Writes tests/golden/ref_*.npz.  Weights are not stored (5 MB): they are the reference constructors'
output under torch.manual_seed(0) (exp_runner_blending.py:125-129 order) plus the deterministic
perturbation of tests/common.py; the fixtures carry per-network checksums so the tests can prove they
rebuilt the same weights."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from common import build_modules, perturb_, state_dicts, checksum  # noqa: E402
from refload import load_reference  # noqa: E402
from neuraludf_amd import synth  # noqa: E402

CASES = {
    # BASELINE config 1: 64 rays x 32 samples, no importance sampling / outside
    "cfg1_flat": dict(n_rays=64, kw=dict(n_samples=32, n_importance=0, n_outside=0, up_sample_steps=1, perturb=1.0)),
    "classical_bg": dict(n_rays=32, kw=dict(n_samples=32, n_importance=20, n_outside=8, up_sample_steps=5,
                                            perturb=1.0)),
    # the reference's other sdf2alpha branch ('theorical', udf_renderer_blending.py:321-323): core + up-sampling
    "theorical_bg": dict(n_rays=32, kw=dict(n_samples=32, n_importance=20, n_outside=8, up_sample_steps=5, perturb=1.0,
                                            sdf2alpha_type="theorical")),
    # the reference's other udf_out branches (fields.py:184-190); same weights, the head differs
    "square_bg": dict(n_rays=32, udf_type="square",
                      kw=dict(n_samples=32, n_importance=20, n_outside=8, up_sample_steps=5, perturb=1.0)),
    # ('sdf' is pinned at the network level only, tests/test_gpu_kernels.py: with it the reference's own up-sampling
    # produces NaN samples on these weights -- negative "udf" inside the surface -- and stops in pdb, :97-101)
    # the colour network's other branch (any mode but 'no_normal', fields.py:456-461): detached unit normals in its input
    "idr_bg": dict(n_rays=32, build=dict(color_mode="idr"),
                   kw=dict(n_samples=32, n_importance=20, n_outside=8, up_sample_steps=5, perturb=1.0)),
    "mix_blend": dict(n_rays=24, kw=dict(n_samples=24, n_importance=12, n_outside=0, up_sample_steps=3, perturb=1.0,
                                         upsampling_type="mix", use_norm_grad_for_cosine=True, h_patch_size=3),
                      blend=True),
}
KEYS = ["z_vals", "color", "color_base", "weights", "depth", "udf", "gradients", "normals", "vis_prob", "alpha",
        "weight_sum", "weight_sum_fg_bg", "gradient_error", "gradient_error_near_surface", "sparse_error", "true_cos"]
GRAD_KEYS = [("udf", "lin0.weight_v"), ("udf", "lin4.weight_g"), ("udf", "lin8.bias"), ("color", "lin_base0.weight_v"),
             ("color", "lin4.weight_v"), ("var", "variance"), ("beta", "beta")]


def main():
    rf, rr, rl = load_reference()
    mods0 = perturb_(build_modules(rf, seed=0))
    sums0 = {k: checksum(v) for k, v in state_dicts(mods0).items()}
    scene = synth.make_scene("tiny")
    only = sys.argv[1:]
    for name, case in CASES.items():
        if only and name not in only:
            continue
        mods, sums = mods0, sums0
        if case.get("build"):
            mods = perturb_(build_modules(rf, seed=0, **case["build"]))
            sums = {k: checksum(v) for k, v in state_dicts(mods).items()}
        for m in mods.values():
            m.zero_grad()
        n = case["n_rays"]
        mods["udf"].udf_type = case.get("udf_type", "abs")      # read at call time by UDFNetwork.udf_out
        rays = synth.make_rays(scene, 0, n, seed=5, margin=6)
        r = rr.UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **case["kw"])
        kw = {}
        src = None
        if case.get("blend"):
            src = synth.make_source_views(scene, 0, 8)
            kw = dict(color_maps=src["color_maps"], w2cs=src["w2cs"], intrinsics=src["intrinsics"],
                      query_c2w=src["query_c2w"], rays_uv=rays["rays_uv"].clone())
        out = r.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=0.7,
                       perturb_overwrite=0, flip_saturation=0.9, **kw)
        loss = ((out["color"] - rays["true_rgb"]).abs().mean() + 0.5 * (out["color_base"] - rays["true_rgb"]).abs().mean()
                + 0.1 * out["gradient_error"] + 0.01 * out["gradient_error_near_surface"] + 0.001 * out["sparse_error"])
        keys = list(KEYS)
        if case.get("blend"):
            keys += ["color_pixel", "patch_colors", "patch_mask"]
            g = torch.Generator().manual_seed(3)
            gt_patch = torch.rand(n, 49, 3, generator=g)
            pmask = (out["patch_mask"].detach() > 0.3).reshape(-1, 1)
            crit = rl.ColorLoss(color_base_weight=1.0, color_weight=1.0, color_pixel_weight=0.5,
                                color_patch_weight=0.2, pixel_loss_type="l1", patch_loss_type="ssim", h_patch_size=3)
            cl = crit(out["color_base"], out["color"], rays["true_rgb"], out["color_pixel"], rays["mask"],
                      out["patch_colors"], gt_patch, pmask.clone())
            loss = loss + cl["loss"]
        loss.backward()
        data = {"ray_" + k: v.numpy() for k, v in rays.items()}
        data.update({"out_" + k: out[k].detach().numpy() for k in keys})
        data["loss"] = np.float64(loss.item())
        for net, pn in GRAD_KEYS:
            p = dict(mods[net].named_parameters())[pn]
            if p.grad is not None:
                data[f"grad_{net}_{pn}"] = p.grad.numpy()
        if case.get("blend"):
            data["gt_patch"] = gt_patch.numpy()
            data["pmask"] = pmask.numpy()
            for k in cl:
                data["closs_" + k] = np.float64(float(cl[k]))
        for k, v in sums.items():
            data["wsum_" + k] = np.float64(v)
        np.savez_compressed(os.path.join(HERE, f"ref_{name}.npz"), **data)
        print(name, "loss", loss.item(), {k: tuple(out[k].shape) for k in ("z_vals", "weights")})


if __name__ == "__main__":
    main()
