"""How far is the fp32 REFERENCE's own parameter gradient from the exact one?  Calibration of the gradient-parity bar.

    python tests/golden/make_golden_eref.py [case ...]      (cases of make_golden_full.py; default: all of them)

The full-size fixtures (ref_<case>_full.npz) hold every parameter gradient `loss.backward()` produced in the fp32
reference (/root/reference/exp_runner_blending.py:367-375 through models/fields.py:219-231, create_graph=True).  Those are
second-order quantities with cancellation, so the fp32 reference itself is some distance away from the exact gradient;
a parity bar of "1e-3 relative" is only meaningful for tensors whose reference value is better than that.  This script
evaluates the ORACLE (oracle/udf_oracle.py, pinned against the reference by tests/test_oracle_vs_reference.py) in float64
on the fixture's inputs -- same rays, same weights, the reference's own fp32 sample positions `out_z_vals` as constants
(the up-sampling runs under no_grad in the reference, udf_renderer_blending.py:723-755) -- with the fixture's loss, and
writes per gradient tensor into tests/golden/ref_eref.json:

    inf  = max|g_ref32 - g_64| / max|g_64|          l2 = ||g_ref32 - g_64||_2 / ||g_64||_2
    ninf = max|g_64|   n2 = ||g_64||_2              (so that a reader sees which tensors are tiny)
    g64  = the float64 gradient itself for tensors of <= 4 elements (variance, beta)

tests/test_gpu_fullsize_parity.py asserts  grel(hip, ref32) < max(1e-3, 3 * inf)  and the same with the 2-norms."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from common import build_modules, perturb_, state_dicts, checksum, oracle_nets, smooth_images  # noqa: E402
from make_golden_full import CASES, SCENE, loss_of  # noqa: E402
from neuraludf_amd import synth  # noqa: E402
from oracle import udf_oracle as O  # noqa: E402

OUT = os.path.join(HERE, "ref_eref.json")
GRAD_CASES = ["cfg2", "dtu_shipped", "cfg5_shape", "cfg3_blend", "cfg3_garment", "bench_cfg2"]


def run(case):
    kw = CASES[case]
    fx = dict(np.load(os.path.join(HERE, "ref_%s_full.npz" % case)))
    from neuraludf_amd.models import fields       # same constructors / initialisation as the reference's (checksum below)
    mods = build_modules(fields, seed=0)
    if case != "bench_cfg2":                       # make_golden_full.py: the bench inputs use the seed-0 weights as they are
        mods = perturb_(mods)
    sds = state_dicts(mods)
    for k, v in sds.items():
        assert abs(checksum(v) - float(fx["wsum_" + k])) < 1e-6 * max(1.0, abs(float(fx["wsum_" + k]))), k
    rays = {k[4:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("ray_")}
    cfg = O.RenderCfg(**{k: v for k, v in kw.items() if k != "perturb"})
    n = rays["rays_o"].shape[0]
    z32 = torch.from_numpy(fx["out_z_vals"])
    _, z_out32, sd = O.coarse_z(cfg, rays["near"], rays["far"], n)
    blend32 = None
    if "h_patch_size" in kw:       # built under the float32 default: torch.rand draws other numbers for another dtype
        scene = synth.make_scene(SCENE.get(case, "tiny"))
        src = synth.make_source_views(scene, 0, 8)
        blend32 = dict(color_maps=smooth_images(8, scene.H, scene.W), w2cs=src["w2cs"], intrinsics=src["intrinsics"],
                       query_c2w=src["query_c2w"], rays_uv=rays["rays_uv"].clone())
    t0 = time.time()
    torch.set_default_dtype(torch.float64)
    try:
        on = oracle_nets(sds, requires_grad=True, dtype=torch.float64)
        ro, rd = rays["rays_o"].double(), rays["rays_d"].double()
        z = z32.double()
        ba = bc = None
        if cfg.n_outside:
            zo = z_out32.double()
            zf, _ = torch.sort(torch.cat([z, zo.expand(n, -1) if zo.dim() == 1 or zo.shape[0] != n else zo], -1), -1)
            ba, bc = O.render_core_outside(on, cfg, ro, rd, zf, sd)
        blend = {k: v.double() for k, v in blend32.items()} if blend32 is not None else None
        car, fs = (1.0, 1.0) if case == "bench_cfg2" else (0.7, 0.9)      # the schedules make_golden_full.py rendered with
        out = O.render_core(on, cfg, ro, rd, z, sd, car, None, ba, bc, fs, blend)
        r64 = {k: v.double() for k, v in rays.items()}
        loss = loss_of(out, r64)
        if blend is not None:
            cl = O.color_loss(1.0, 1.0, 0.5, 0.2, 3, out["color_base"], out["color"], r64["true_rgb"], out["color_pixel"],
                              r64["mask"], out["patch_colors"], torch.from_numpy(fx["gt_patch"]).double(),
                              torch.from_numpy(fx["pmask"]))
            loss = loss + cl["loss"]
        loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    rec = {"loss64": float(loss), "loss_ref32": float(fx["loss"]), "seconds": round(time.time() - t0, 1), "tensors": {}}
    for net in ("udf", "color", "var", "beta", "nerf"):
        sd_net = getattr(on, net)
        if sd_net is None:
            continue
        for pn, t in sd_net.items():
            key = f"grad_{net}_{pn}"
            if key not in fx or t.grad is None:
                continue
            g64 = t.grad.reshape(-1)
            g32 = torch.from_numpy(fx[key]).double().reshape(-1)
            ninf, n2 = float(g64.abs().max()), float(g64.norm())
            e = {"inf": float((g32 - g64).abs().max() / max(ninf, 1e-300)), "l2": float((g32 - g64).norm() / max(n2, 1e-300)),
                 "ninf": ninf, "n2": n2, "numel": int(g64.numel())}
            if g64.numel() <= 4:
                e["g64"] = [float(x) for x in g64]
                e["g_ref32"] = [float(x) for x in g32]
            rec["tensors"][key] = e
    worst = max(rec["tensors"].items(), key=lambda kv: kv[1]["inf"])
    print(f"{case}: {len(rec['tensors'])} gradient tensors, fp64 oracle {rec['seconds']} s, loss {rec['loss64']:.8f} "
          f"(reference fp32 {rec['loss_ref32']:.8f}); the fp32 reference's worst distance from fp64: {worst[0]} "
          f"inf {worst[1]['inf']:.2e} l2 {worst[1]['l2']:.2e}")
    return rec


def main():
    cases = sys.argv[1:] or GRAD_CASES
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    for c in cases:
        data[c] = run(c)
        with open(OUT, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
