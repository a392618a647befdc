#!/usr/bin/env python
"""Headline benchmark: ray-samples/s of one full train step of the NeuralUDF volume-rendering hot
path (render + colour/eikonal loss + backward + Adam) on synthetic DTU-scan24-shaped rays.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no launcher: bench.py starts the N ranks itself through torch.distributed.run)

Workload (BASELINE.json configs[1]): 512 rays x 128 samples per GPU (64 coarse + 64 hierarchical in
4 rounds, no outside samples); arithmetic: fp32 EMULATED on the 16-bit matrix pipe (`--precision bf16x3`, the default; `fp32`
= the exact v_mfma_f32_32x32x2_f32 kernels, also timed here as the secondary `fp32_exact` leg).  With N GPUs the global
batch is 512*N rays, ray-sharded, one packed loss all-reduce + one gradient all-reduce per step (weak scaling);
`--scaling strong --global-rays 4096` (BASELINE configs[3]) fixes the global batch and gives every rank 4096 / N rays.
The timed region is `--windows` (5) windows of `--steps` steps each, every window bracketed by barrier + synchronize and
reduced with MAX over ranks; `ms_per_step` is the MEDIAN window, `window_ms` lists them all.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel class of the step (the fused layer-chain kernel `mlp_chain_kernel`): flops its
                  launches execute / their summed duration, measured with HIP events on the launch stream during one extra
                  instrumented step.  Default mode "bf16x3" (fp32 emulated on the 16-bit matrix pipe: three fp16 MFMA products per
                  fp32 product, six bf16 ones where a switch asks for them -- `dtype` says which): executed flops = 3 x / 6 x
                  2 M N K per launch against the dense 16-bit peak 2.5 PFLOP/s, with the
                  fp32-equivalent rate beside the 157.3 TFLOP/s of the fp32 MFMA pipe (`--precision fp32`: the exact
                  v_mfma_f32_32x32x2_f32 kernels against 157.3); traffic = HBM bytes per launch from the committed rocprofv3
                  PMC passes over this command (profiles/r06_traffic_mlp_chain_<mode>.json, see `traffic_stale`).
  roofline_composite -- the fused sample+composite kernels against the 8 TB/s HBM roof
                  (48 B/sample + 68 B/ray forward, 84 B/sample + 68 B/ray backward).
  cpu_baseline -- the reference's own classes (oracle/_ref/reference_tree, kind "reference"; the oracle's port when that
                  tree is absent) timed on this host's cores, one pinned thread per physical core, on a bounded sample of
                  the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_F16_PEAK_TFLOPS = 2500.0     # dense fp16 / bf16 (MI355X_MICROARCH.md)
# bf16x3 (the default mode): every fp32 product is SIX bf16 MFMA products (both operands split exactly into three bf16
# parts; hi hi, hi mid, mid hi, hi lo, lo hi, mid mid; fp32 accumulate) -- the flops a launch EXECUTES on the bf16 pipe
X3_PRODUCTS = 6
DTYPE = {"fp32": "f32",
         "mixed16": "f16/bf16 MFMA operands, f32 accumulate, bf16 saved state (config 5 mode)"}


def dtype_string(precision):
    """what the arithmetic of this run was, from the switches actually in force (mlp.FWD_F16X2, mlp.TN_F16X2)"""
    if precision != "bf16x3":
        return DTYPE[precision]
    from neuraludf_amd import mlp
    b3 = "bf16x3 (operands split exactly into 3 bf16 parts, 6 bf16 MFMA products per f32 product)"
    f2 = "f16x2 (x = hi + 2^-11 lo in fp16, 3 fp16 MFMA products, correction terms in their own accumulator)"
    fwd = f2 if mlp.FWD_F16X2 != "0" else b3
    bwd = (f2 + ", every tile of 64 points multiplied by its own power of two so that the loss adjoints lie in fp16's range "
           "(NudfChain.tile_scale: exact, memory holds the unscaled values)") if mlp._sweep_dtype("bwd") == "f16x2" else b3
    tn = ("f16x2 for the UDF and colour networks (each operand scaled by a power of two from its maximum, tracked on the device "
          "by the sweep that wrote it), bf16x3 for the background NeRF") if mlp.TN_F16X2 else "bf16x3"
    return ("f32 emulated on the 16-bit matrix pipe, f32 accumulate, f32 state / epilogues / optimizer.  Forward-order sweeps (UDF "
            f"value, input gradient, colour / NeRF forward): {fwd if fwd is f2 else 'bf16x3 (NUDF_FWD_F16X2=0)'}.  Backward sweeps "
            f"(tangent, adjoint, ReLU backward): {bwd}.  Weight-gradient GEMMs: {tn}.  fp32-level accuracy against float64 "
            "in each (tests/test_gpu_bf16x3.py, tests/test_gpu_round6.py)")


HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (rays per GPU, renderer conf, scene)
    "dtu_scan24_512x128": (512, dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0),
                           "dtu"),
    "dtu_shipped_512x114+32": (512, dict(n_samples=64, n_importance=50, n_outside=32, up_sample_steps=5,
                                         perturb=1.0), "dtu"),
    # BASELINE config 5 per GPU: 8192 rays x 256 samples over 8 GPUs = 1024 rays x (128 + 128) each
    "dtu_scan24_1024x256": (1024, dict(n_samples=128, n_importance=128, n_outside=0, up_sample_steps=4, perturb=1.0),
                            "dtu"),
    # BASELINE config 3: garment scene, mix schedule, pixel + patch blending over 8 source views (7x7 patches)
    "garment_blend_1024x128": (1024, dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=3, perturb=1.0,
                                          upsampling_type="mix", use_norm_grad_for_cosine=True, h_patch_size=3),
                               "garment"),
}
BLEND_WORKLOADS = {"garment_blend_1024x128": dict(color_pixel_weight=0.5, color_patch_weight=0.1)}
# BASELINE configs[3]: 4096 GLOBAL rays x (64 + 64), ray-sharded over the ranks (strong scaling: 4096 / N rays per GPU)
WORKLOADS["dtu_scan118_4096x128"] = (4096, dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0),
                                     "dtu")
STRONG_WORKLOADS = {"dtu_scan118_4096x128"}


def _cpu_info():
    """(cpu model string, logical CPUs one per physical core within this process's affinity)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    firsts, seen = [], set()
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            firsts.append(c)
    return model, firsts


class _PowerSampler:
    """socket power / shader clock while the timed step replays (amdgpu hwmon: power1_average | power1_input in uW, freq1_input in
    Hz, power1_cap; rocm-smi --csv when sysfs has none).  Round 5 found the step running against the package power limit
    (profiles/r05_tn_power.txt): the clock the kernels see is part of the measurement."""

    def __init__(self, period=0.02, device=0):
        import glob
        self.period, self.samples, self._stop, self._th = period, [], False, None
        self.cards = []
        want = None
        try:                                       # the card of THIS process's device (a node's other GPUs are visible in sysfs)
            pr = torch.cuda.get_device_properties(device)
            want = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            pci = os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(h))))
            if want and not pci.startswith(want):
                continue
            pw = next((os.path.join(h, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, n))), None)
            if pw:
                self.cards.append({"power": pw, "freq": os.path.join(h, "freq1_input"), "cap": os.path.join(h, "power1_cap"), "pci": pci})

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _smi(self):
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            rows = [r for r in out.strip().splitlines() if r.startswith("card")]
            hdr = [r for r in out.strip().splitlines() if r.startswith("device")][0].split(",")
            best = None
            for r in rows:
                d = dict(zip(hdr, r.split(",")))
                w = float([v for k, v in d.items() if "Power (W)" in k][0])
                mhz = float(d.get("sclk clock speed:", "(0Mhz)").strip("()").lower().replace("mhz", ""))
                if best is None or w > best[0]:
                    best = (w, mhz)
            return best
        except Exception:
            return None

    def _run(self):
        while not self._stop:
            if self.cards:
                best = None
                for c in self.cards:
                    w, f = self._read(c["power"]), self._read(c["freq"])
                    if w is not None and (best is None or w > best[0]):
                        best = (w / 1e6, (f or 0.0) / 1e6)
                if best:
                    self.samples.append(best)
                time.sleep(self.period)
            else:
                b = self._smi()
                if b:
                    self.samples.append(b)

    def __enter__(self):
        import threading
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self._th.join(timeout=10)

    def summary(self):
        if not self.samples:
            return None
        s = self.samples[len(self.samples) // 4:] or self.samples            # drop the ramp
        cap = None
        for c in self.cards:
            v = self._read(c["cap"])
            cap = v / 1e6 if v else cap
        return {"avg_w": sum(x[0] for x in s) / len(s), "max_w": max(x[0] for x in s), "cap_w": cap,
                "sclk_mhz_avg": sum(x[1] for x in s) / len(s), "sclk_mhz_min": min(x[1] for x in s), "samples": len(s),
                "source": ("amdgpu hwmon of %s (power1_average|input, freq1_input)" % ",".join(c["pci"] for c in self.cards))
                if self.cards else "rocm-smi --showpower --showclocks",
                "what": "this rank's card while the timed step replays for ~1 s AFTER the timed windows (not inside them); the "
                        "dense bf16 peak of `roofline` assumes 2400 MHz"}


def _vs_reference_fixture(workload, n_rays, rays, rend, dev):
    """the same comparison against outputs the REFERENCE ITSELF produced (build container, CPU) on the timed inputs:
    tests/golden/ref_bench_cfg2_full.npz, written by tests/golden/make_golden_full.py bench_cfg2 -- bench.py's own ray
    construction (dtu scene, camera 0, seed 1234), seed-0 weights, cos_anneal_ratio = flip_saturation = 1.  The fixture's
    stored rays are rendered (this host's CPU builds the same rays to within an ulp of the ray directions; `ray_ulp_diff`
    says by how much).  None for other workloads."""
    import math
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "ref_bench_cfg2_full.npz")
    if workload != "dtu_scan24_512x128" or n_rays != 512 or not os.path.exists(path):
        return None
    fx = np.load(path)
    fr = {k: torch.from_numpy(fx["ray_" + k]) for k in ("rays_o", "rays_d", "near", "far")}
    ray_diff = float((fr["rays_d"] - rays["rays_d"]).abs().max())
    if ray_diff > 1e-6:
        return {"error": "fixture rays differ from the timed rays by %g" % ray_diff}
    with torch.no_grad():
        got = rend.render(fr["rays_o"].to(dev), fr["rays_d"].to(dev), fr["near"].to(dev), fr["far"].to(dev),
                          cos_anneal_ratio=1.0, flip_saturation=1.0, perturb_overwrite=0)
    got = {k: got[k].cpu() for k in ("color", "z_vals", "weights")}
    col, zv, w = (torch.from_numpy(fx[k]) for k in ("out_color", "out_z_vals", "out_weights"))
    zerr = (got["z_vals"] - zv).abs().max(dim=1)[0]
    same, exact = zerr < 1e-4, zerr < 1e-5
    mse = float(((got["color"] - col) ** 2).mean())
    return {"value_db": 20.0 * math.log10(1.0 / math.sqrt(mse + 1e-30)), "max_abs_diff": float((got["color"] - col).abs().max()),
            "rays_with_identical_samples": int(same.sum()), "rays_with_samples_within_1e-5": int(exact.sum()),
            "max_abs_diff_on_rays_with_samples_within_1e-5": float((got["color"] - col)[exact].abs().max()) if bool(exact.any()) else None,
            "max_abs_diff_on_rays_with_identical_samples": float((got["color"] - col)[same].abs().max()) if bool(same.any()) else None,
            "weights_max_abs_diff_on_rays_with_identical_samples": float((got["weights"] - w)[same].abs().max()) if bool(same.any()) else None,
            "weights_max_abs_diff_on_rays_with_samples_within_1e-5": float((got["weights"] - w)[exact].abs().max()) if bool(exact.any()) else None,
            "ray_ulp_diff": ray_diff,
            "fixture": "tests/golden/ref_bench_cfg2_full.npz (the reference's own CPU run on these rays and weights)"}


def cpu_baseline(workload, seconds_budget=25.0, dev=None, precision="fp32", n_rays=None):
    """The CPU leg of BASELINE's metric: one full train step (render + loss + backward + Adam) on the host cores, on a
    bounded sample of the timed workload.  kind "reference": the reference's OWN UDFRendererBlending / ColorLoss classes
    + torch.optim.Adam, imported from oracle/_ref/reference_tree (the git-ignored copy oracle/make_ref_tree.py stages in
    the build container; it travels with the gpurun snapshot -- /root/reference itself is never read here); kind "port":
    the oracle's restatement, when that tree is absent.  Threads are pinned one per physical core (<= 32, the fastest of a
    few counts on a probe pass: more threads than that only slow these small ops down).  With `dev` also the second half of
    the metric: PSNR of the HIP path's colours (and the weights) against the oracle's on the same rays and weights (the
    oracle is the checker here, nothing else)."""
    from neuraludf_amd import synth
    from neuraludf_amd.train import DTU_MODEL_CONF
    from oracle import udf_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    rays_cfg, rconf, scene_kind = WORKLOADS[workload]
    # the timed configuration's own ray count (512 for the headline): a warm-up + >= 3 full steps fit the budget
    n_rays = int(n_rays or min(rays_cfg, 512))
    # seeded reference-identical init through the drop-in modules (CPU construction only, no kernels)
    from neuraludf_amd.models import fields
    from common import build_modules, state_dicts
    mods = build_modules(fields, seed=0)
    sds = state_dicts(mods)
    nets = O.Nets(**{k: {n: t.clone().requires_grad_(t.is_floating_point()) for n, t in sds[k].items()}
                     for k in ("udf", "color", "var", "beta", "nerf")})
    nets.beta["gamma"].requires_grad_(False)
    nets.beta["zeta"].requires_grad_(False)
    cfg = O.RenderCfg(**{k: v for k, v in rconf.items() if k != "perturb"})
    scene = synth.make_scene(scene_kind)
    rays = synth.make_rays(scene, 0, n_rays, seed=1234)
    s_core = rconf["n_samples"] + rconf["n_importance"]

    psnr = None
    if dev is not None:
        import math
        from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
        g = {k: m.to(dev) for k, m in build_modules(fields, seed=0).items()}      # same seed -> the same weights
        rend = UDFRendererBlending(g["nerf"], g["udf"], g["var"], g["color"], g["beta"], **rconf)
        with torch.no_grad():
            got = rend.render(rays["rays_o"].to(dev), rays["rays_d"].to(dev), rays["near"].to(dev), rays["far"].to(dev),
                              cos_anneal_ratio=1.0, flip_saturation=1.0, perturb_overwrite=0)
            got = {k: got[k].cpu() for k in ("color", "z_vals", "weights")}
            ref = O.render(nets, cfg, rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=1.0,
                           flip_saturation=1.0)
        mse = float(((got["color"] - ref["color"]) ** 2).mean())          # exp_runner_blending.py:341-342 with mask = 1
        dz = (got["z_vals"] - ref["z_vals"]).abs()
        same = dz.max(dim=1)[0] < 1e-4
        exact = dz.max(dim=1)[0] < 1e-5
        dw = (got["weights"] - ref["weights"]).abs()
        moved = dz >= 1e-4                                                 # individual samples at other positions
        wsum = float(ref["weights"].sum())
        psnr = {"value_db": 20.0 * math.log10(1.0 / math.sqrt(mse + 1e-30)),
                "max_abs_diff": float((got["color"] - ref["color"]).abs().max()),
                "max_abs_diff_on_rays_with_identical_samples": float((got["color"] - ref["color"])[same].abs().max()) if bool(same.any()) else None,
                "weights_max_abs_diff_on_rays_with_identical_samples": float(dw[same].max()) if bool(same.any()) else None,
                "weights_max_abs_diff_on_rays_with_samples_within_1e-5": float(dw[exact].max()) if bool(exact.any()) else None,
                "rays_with_samples_within_1e-5": int(exact.sum()),
                "weights_max_abs_diff": float(dw.max()),
                "weight_mass_on_moved_samples": (float(ref["weights"][:, :moved.shape[1]][moved].sum()) / wsum) if wsum > 0 else None,
                "rays": n_rays, "rays_with_identical_samples": int(same.sum()), "samples_per_ray": s_core, "precision": precision,
                "vs_reference_fixture": _vs_reference_fixture(workload, n_rays, rays, rend, dev),
                "what": "HIP colours / weights vs the oracle's END TO END on identical rays / network weights: the hierarchical "
                        "sampling runs on both sides, so rays whose quantile bins flip (tests/test_gpu_fullsize_parity.py) enter "
                        "with different sample positions; weight_mass_on_moved_samples = share of the reference's total "
                        "compositing weight that sits on samples whose position differs by >= 1e-4"}

    # ---- the timed CPU step ----
    import refload
    kind = "reference" if refload.have_reference(refload.REF_TREE) else "port"
    if kind == "reference":
        import contextlib
        import io
        rf, rr, rl = refload.load_reference(refload.REF_TREE)
        with contextlib.redirect_stdout(io.StringIO()):
            rmods = build_modules(rf, seed=0)                # the reference's own classes, the runner's order and seed
            rrend = rr.UDFRendererBlending(rmods["nerf"], rmods["udf"], rmods["var"], rmods["color"], rmods["beta"], **rconf)
            crit = rl.ColorLoss(color_base_weight=0.01, color_weight=1.0, color_pixel_weight=0.0, color_patch_weight=0.0,
                                pixel_loss_type="l1", patch_loss_type="ssim", h_patch_size=3)
        geo = list(rmods["udf"].parameters())
        other = list(rmods["var"].parameters()) + list(rmods["color"].parameters()) + list(rmods["beta"].parameters())
        opt = torch.optim.Adam([{"params": geo, "lr": 1e-4}, {"params": other}, {"params": list(rmods["nerf"].parameters())}],
                               lr=5e-4)                      # exp_runner_blending.py:136-139

        def step():
            with contextlib.redirect_stdout(io.StringIO()):
                out = rrend.render(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=1.0,
                                   flip_saturation=1.0)
                cl = crit(out["color_base"], out["color"], rays["true_rgb"], out["color_pixel"], None, out["patch_colors"],
                          None, None)
            loss = cl["loss"] + 0.1 * out["gradient_error"]   # exp_runner_blending.py:367-371 with the bench's weights
            opt.zero_grad()
            loss.backward()
            opt.step()

        def probe():          # (the reference's render differentiates inside: no no_grad here)
            with contextlib.redirect_stdout(io.StringIO()):
                rrend.render(rays["rays_o"][:16], rays["rays_d"][:16], rays["near"][:16], rays["far"][:16],
                             cos_anneal_ratio=1.0, flip_saturation=1.0, perturb_overwrite=0)
        what = "oracle/_ref/reference_tree: the reference's UDFRendererBlending + ColorLoss + torch.optim.Adam"
    else:
        params = [t for d in (nets.udf, nets.color, nets.var, nets.beta, nets.nerf) for t in d.values() if t.requires_grad]
        opt = torch.optim.Adam(params, lr=5e-4)

        def step():
            t_rand = torch.rand(n_rays, 1) - 0.5
            out = O.render(nets, cfg, rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=1.0,
                           flip_saturation=1.0, t_rand=t_rand,
                           t_rand_out=torch.rand(cfg.n_outside) if cfg.n_outside else None)
            cl = O.color_loss(0.01, 1.0, 0.0, 0.0, 3, out["color_base"], out["color"], rays["true_rgb"], None, None,
                              None, None, None)
            loss = cl["loss"] + 0.1 * out["gradient_error"]
            opt.zero_grad()
            loss.backward()
            opt.step()

        def probe():
            with torch.no_grad():
                O.render(nets, cfg, rays["rays_o"][:16], rays["rays_d"][:16], rays["near"][:16], rays["far"][:16],
                         cos_anneal_ratio=1.0, flip_saturation=1.0)
        what = "oracle/udf_oracle.py (a port of the reference's PyTorch path; oracle/_ref/reference_tree absent)"

    # one thread per PHYSICAL core, pinned; all cores is pathological for these small ops on a many-core host (256 threads
    # -> > 100 s/step measured), so the fastest of a few counts on a probe pass is used
    model, cores = _cpu_info()
    old_aff = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    old_threads = torch.get_num_threads()
    best = None
    try:
        for nt in sorted({min(len(cores), c) for c in (8, 16, 32)}):
            if old_aff is not None:
                os.sched_setaffinity(0, set(cores[:nt]))
            torch.set_num_threads(nt)
            probe()
            t0 = time.time()
            probe()
            dtt = time.time() - t0
            if best is None or dtt < best[0]:
                best = (dtt, nt)
        nt = best[1]
        if old_aff is not None:
            os.sched_setaffinity(0, set(cores[:nt]))
        torch.set_num_threads(nt)
        step()  # warm-up
        times = []
        t_start = time.time()
        while len(times) < 40 and (len(times) < 3 or (time.time() - t_start) < seconds_budget):
            t0 = time.time()
            step()
            times.append(time.time() - t0)
    finally:
        if old_aff is not None:
            os.sched_setaffinity(0, old_aff)
        torch.set_num_threads(old_threads)
    times.sort()
    med = times[len(times) // 2]
    res = {"value": n_rays * s_core / med, "unit": "ray-samples/s", "cores": nt, "kind": kind, "rays": n_rays,
           "cpu_model": model, "physical_cores_available": len(cores), "pinned": old_aff is not None,
           "sample": f"{n_rays} rays x {s_core} samples, {len(times)} full train steps, fp32, median {med:.3f} s/step ({what})"}
    return res, psnr


def _self_launch(n):
    """re-run this command line as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same args>`;
    -> the launcher's exit code.  The children inherit stdout / stderr, so rank 0's JSON line is this process's output."""
    import socket
    import subprocess
    with socket.socket() as sk:                      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="dtu_scan24_512x128", choices=list(WORKLOADS))
    ap.add_argument("--rays-per-gpu", type=int, default=0)
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak (default): --rays-per-gpu rays on every rank; strong: --global-rays rays in total, "
                         "global / N per rank (default for the dtu_scan118_4096x128 workload, BASELINE configs[3])")
    ap.add_argument("--global-rays", type=int, default=0, help="strong scaling: rays of the whole job (default: the workload's)")
    ap.add_argument("--no-power", action="store_true", help="skip the ~1 s power / shader-clock sampling window after the timed region")
    ap.add_argument("--windows", type=int, default=5,
                    help="timed windows of --steps steps each; ms_per_step is the median window (box-to-box and "
                         "window-to-window spread of a 0.1 s region is ~3 %%)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the secondary exact-fp32 timing leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the extra forward-only timing (profiling runs)")
    ap.add_argument("--fused-adam", type=int, default=1)
    ap.add_argument("--graph", type=int, default=int(os.environ.get("NUDF_BENCH_GRAPH", "1")),
                    help="1: the timed steps are replays of the train step captured in a HIP graph (train.GraphedStep: "
                         "bit-identical to eager steps, one graph launch per step); 0: eager launches")
    ap.add_argument("--precision", default=os.environ.get("NUDF_PRECISION", "bf16x3"), choices=["fp32", "bf16x3", "mixed16"],
                    help="bf16x3 (default, the headline) = fp32 products EMULATED on the bf16 matrix pipe (exact 3-way operand "
                         "split, 6 bf16 MFMA products, fp32 accumulate: fp32-level accuracy, not bit-comparable with fp32 "
                         "MFMA); fp32 = the exact v_mfma_f32_32x32x2_f32 kernels (also timed as the `fp32_exact` leg); "
                         "mixed16 = BASELINE config 5 (16-bit MFMA operands, fp32 accumulate) -- never the headline")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU through
    # torch.distributed.run on 127.0.0.1) and relay them -- rank 0 of the child job prints the JSON line.  Under a launcher
    # (WORLD_SIZE set) the two must agree: a job that silently ran another number of ranks than it was asked for is an error.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(_self_launch(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"error": "--gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"])}))
        sys.exit(2)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible: bench.py measures the HIP path only"}))
        sys.exit(1)
    # one rank per GPU.  NUDF_DIST_BACKEND=gloo lets the 2-rank flow be exercised on a ONE-GPU box (both ranks on
    # cuda:0, collectives staged through the host) -- tests only; the measured configuration is nccl (= RCCL).
    backend = os.environ.get("NUDF_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from neuraludf_amd import mlp, synth
    from neuraludf_amd import dist as nd
    from neuraludf_amd.train import Trainer

    mlp.set_precision(args.precision)
    rays_per_gpu, rconf, scene_kind = WORKLOADS[args.workload]
    scaling = args.scaling or ("strong" if args.workload in STRONG_WORKLOADS else "weak")
    if scaling == "strong":
        global_rays = args.global_rays or (rays_per_gpu if args.workload in STRONG_WORKLOADS else rays_per_gpu * 8)
        if global_rays % world:
            print(json.dumps({"error": "strong scaling: %d global rays do not divide over %d ranks" % (global_rays, world)}))
            sys.exit(1)
        rays_per_gpu = global_rays // world
    elif args.rays_per_gpu:
        rays_per_gpu = args.rays_per_gpu
    elif args.workload in STRONG_WORKLOADS:
        rays_per_gpu = rays_per_gpu // 8          # the workload's per-GPU share on the 8-GPU node it is quoted on
    fused = bool(args.fused_adam)
    try:
        import neuraludf_amd.optim  # noqa: F401
    except Exception:
        fused = False
    lconf = BLEND_WORKLOADS.get(args.workload)
    tr = Trainer(dev, rconf, color_loss_conf=lconf, seed=0, data_parallel=(world > 1), fused_adam=fused)
    tr.renderer.diagnostics = False
    scene = synth.make_scene(scene_kind)
    rays = synth.make_rays(scene, 0, rays_per_gpu * world, seed=1234, margin=8 if lconf else 0)
    batch = {k: nd.shard(v, rank, world).contiguous().to(dev) for k, v in rays.items()}
    step_kw = {}
    if lconf:      # source views resident in HBM, ground-truth patches as the batch generator would crop them
        step_kw["blend"] = {k: v.to(dev) for k, v in synth.make_source_views(scene, 0, 8, hwc=True).items()}
        npx = (2 * rconf["h_patch_size"] + 1) ** 2
        batch["gt_patch_colors"] = torch.rand(batch["rays_o"].shape[0], npx, 3, device=dev)
    s_core = rconf["n_samples"] + rconf["n_importance"]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    from neuraludf_amd.train import GraphedStep
    # a ray-sharded step (world > 1) is captured WITH its two RCCL all-reduces (they capture and replay on this image,
    # tests/test_gpu_dist.py): the eager step costs 3.8 ms of Python enqueueing per 4.2 ms of kernels, and 8 such processes
    # share one host.  NUDF_DP_GRAPH=0 keeps the eager launches; the gloo test backend stages through the host and cannot be
    # captured; a capture that fails falls back to eager launches by itself (GraphedStep) -- `config.launch` says which ran.
    dp_graph = os.environ.get("NUDF_DP_GRAPH", "1") == "1" and backend == "nccl"
    gstep = GraphedStep(tr, eager_steps=2, capture_collectives=dp_graph) if args.graph else None
    coll_eager = None
    use_graph = bool(gstep is not None and gstep.enabled)
    run_step = (lambda: gstep(batch, **step_kw)) if use_graph else (lambda: tr.step(batch, **step_kw))
    if use_graph:                      # set-up, not warm-up: two eager steps, then the capture (+ its first replay)
        nd.collective_counts(reset=True)
        run_step()
        coll_eager = dict(nd.collective_counts())      # what ONE step issues (a replay issues them from inside the graph)
        for _ in range(2):
            run_step()
        st_in = gstep.static_inputs()
        if gstep.replays != 1 or st_in is None:      # this configuration could not be captured: eager launches
            use_graph = False
            run_step = lambda: tr.step(batch, **step_kw)
        else:
            # the batch is resident (as in the eager loop): hand the replays the captured step's own input tensors, so
            # that no per-step input copy is enqueued
            batch, blend_static = st_in
            if blend_static is not None:
                step_kw["blend"] = blend_static
    for _ in range(args.warmup):
        run_step()
    torch.cuda.synchronize()
    # A generation-2 pass of Python's cyclic GC over the ~10^6 objects a torch process holds takes ~80 ms of host time
    # (measured, scripts/fwd_only_probe.py: it is what made the forward-only figure of round 1 read 6.26 ms on one box
    # and 2.29 ms on the next -- that pass is host-bound at ~1 ms of enqueueing per 2.3 ms of kernels).  Freeze the
    # set-up objects into the permanent generation, as a training loop should: the timed regions then see only the
    # cheap young-generation collections.
    import gc
    gc.collect()
    gc.freeze()
    def timed_window(fn, steps):
        """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; -> seconds, MAX over ranks"""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dtw = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dtw], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtw = float(t.item())
        return dtw

    nd.collective_counts(reset=True)
    n_win = max(1, args.windows)
    wins = [timed_window(run_step, args.steps) for _ in range(n_win)]
    coll = nd.collective_counts()
    dt = sorted(wins)[len(wins) // 2]                      # the median window
    ms_per_step = dt / args.steps * 1e3
    value = world * rays_per_gpu * s_core / (dt / args.steps)
    if use_graph and coll_eager is not None:               # replays issue the collectives from inside the graph
        coll = {k: v * args.steps * n_win for k, v in coll_eager.items()}

    result = {
        "metric": "ray-samples/sec (train step)", "value": value, "unit": "ray-samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None,
        "window_ms": [w / args.steps * 1e3 for w in wins], "windows": n_win,
        "dtype": dtype_string(args.precision),
        "data": "synthetic",
        "config": {"workload": args.workload, "rays_per_gpu": rays_per_gpu, "global_rays": rays_per_gpu * world,
                   "samples_per_ray": s_core, "n_outside": rconf["n_outside"],
                   "step": "render + L1/eikonal loss + backward + Adam", "parallelism": f"ray-sharded dp{world}",
                   "optimizer": "fused HIP Adam" if fused else "torch.optim.Adam",
                   # (the mode name is historical: which split each sweep class / the GEMMs actually ran -- see `dtype`)
                   "operand_mode": {"precision": args.precision, "forward_order_sweeps": mlp._sweep_dtype("fwd"),
                                    "input_gradient_sweep": mlp._sweep_dtype("grad"), "backward_sweeps": mlp._sweep_dtype("bwd"),
                                    "weight_gradient_gemms": ("f16x2 (UDF, colour) / bf16x3 (NeRF)" if (mlp.TN_F16X2 and
                                                              args.precision == "bf16x3") else mlp._sweep_dtype("bwd")
                                                              if args.precision != "bf16x3" else "bf16x3")},
                   "launch": ("HIP graph replay: one graph launch + one 0.8 KB H2D copy of the step's scalars per step "
                              "(train.GraphedStep; replays are bit-identical to eager steps, tests/test_gpu_graph.py)"
                              + ("; the two RCCL all-reduces of the ray-sharded step are nodes of the graph" if world > 1 else ""))
                   if use_graph else ("eager launches" + (" (ray-sharded step not captured: NUDF_DP_GRAPH=0, a non-RCCL "
                                                         "backend, or the capture failed)" if world > 1 and args.graph else ""))},
        # data-parallel exchange of the timed region (neuraludf_amd/dist.py): packed loss sums + gradient bucket
        "rccl_ranks": world if backend == "nccl" else 0, "dist_backend": backend if world > 1 else None,
        "collectives_per_step": {k: v / (args.steps * n_win) for k, v in coll.items()} if world > 1 else None,
        "gradient_message_floats": (tr.bucket.last_message_floats if world > 1 else None),
    }

    # ---- socket power and shader clock under the same replayed step (outside the timed windows; every rank steps, rank 0 samples)
    if not args.no_power:
        try:
            n_pw = max(args.steps, int(1.0 / max(dt / args.steps, 1e-4)))       # (dt is the max over ranks: the same everywhere)
            if rank == 0:
                with _PowerSampler(device=dev.index or 0) as ps:
                    timed_window(run_step, n_pw)
                result["power"] = ps.summary()
            else:
                timed_window(run_step, n_pw)
        except Exception as ex:                               # never fatal: the line is still valid without it
            result["power"] = {"error": repr(ex)}

    # ---- forward only ("also forward-only ray-samples/s"): the same batch rendered without autograd, all
    # ranks (the ray-sharded render holds a collective), same bracketing
    with torch.no_grad():
        for _ in range(0 if args.no_forward_only else 2):
            tr.loss(batch, **({"blend": step_kw["blend"]} if step_kw else {}))
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(0 if args.no_forward_only else args.steps):
            tr.loss(batch, **({"blend": step_kw["blend"]} if step_kw else {}))
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dtf = (time.perf_counter() - t0) / args.steps
    if not args.no_forward_only:
        result["forward_only"] = {"value": world * rays_per_gpu * s_core / dtf, "unit": "ray-samples/s", "ms": dtf * 1e3,
                                  "what": "render + loss under no_grad, rank-local clock"}

    def instrumented(precision):
        """ONE extra step with HIP events around every libnudf launch class on the launch stream (mlp.PROFILE).  EVERY rank
        takes the step (it contains the data-parallel collectives); rank 0 returns (roofline, kernels), the others None."""
        mlp.PROFILE = [] if rank == 0 else None
        tr.step(batch, **step_kw)
        torch.cuda.synchronize()
        barrier()
        prof, mlp.PROFILE = mlp.PROFILE, None
        if rank != 0:
            return None
        # flops the launches EXECUTE on the pipe they run on: algorithmic (2 M N K) for the fp32 and 16-bit modes; in the
        # bf16x3 mode six bf16 MFMA products per fp32 product where a launch contracts with the 3-way bf16 split, THREE fp16
        # products where it runs the f16x2 split (mlp.FWD_F16X2 / BWD_F16X2 / TN_F16X2: by default every sweep and the UDF /
        # colour weight-gradient GEMMs) -- every launch reports its own executed flops (mlp.MFMA_PRODUCTS by step / group)
        x3 = precision == "bf16x3"
        cls_factor = lambda k: (X3_PRODUCTS if x3 and (k != "gemm_tn" or mlp.TN_SPLIT) else 1)
        agg, per = {}, {}
        for name, flops, s_ev, e_ev, detail, nbytes, xflops in prof:
            dur = s_ev.elapsed_time(e_ev) * 1e-3
            xf = (xflops if xflops is not None else flops * cls_factor(name)) if flops > 0 else 0.0
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += dur
            a[3] += xf
            b = per.setdefault(detail, [name, 0, 0.0, 0.0, 0.0, 0.0])
            b[1] += 1
            b[2] += flops
            b[3] += dur
            b[4] += nbytes
            b[5] += xf
        mfma_cls = {k: v for k, v in agg.items() if v[1] > 0}          # (the HBM-bound classes carry units <= 0 there)
        dom = max(mfma_cls, key=lambda k: mfma_cls[k][2])
        n, fl, sec, xfl = agg[dom]
        factor = {k: (v[3] / v[1] if v[1] > 0 else 1) for k, v in agg.items()}      # executed / algorithmic, per class
        peak = MFMA_F32_PEAK_TFLOPS if precision == "fp32" else MFMA_F16_PEAK_TFLOPS
        roof = {"bound": "mfma", "kernel": dom + "_kernel", "achieved": xfl / sec / 1e12,
                "peak": peak, "unit": "TFLOP/s", "frac": xfl / sec / 1e12 / peak,
                "traffic": None, "launches_per_step": n, "avg_launch_us": sec / n * 1e6,
                "algorithmic_gflop_per_step": fl / 1e9, "executed_flops_per_algorithmic_flop": factor[dom],
                # the convention of `achieved` / `frac`, in one key
                "flops": ("executed on the 16-bit matrix pipe: every launch reports its own products per fp32 product -- 3 (f16x2) or "
                          "6 (bf16x3), see `dtype` for which sweeps run which -- %.2f x the algorithmic 2 M N K of SURVEY 8(d) "
                          "over the class" % factor[dom] if x3 else "algorithmic 2 M N K"),
                "frac_algorithmic_vs_fp32_pipe": fl / sec / 1e12 / MFMA_F32_PEAK_TFLOPS}
        if x3:
            roof["what"] = (
                "achieved = EXECUTED 16-bit MFMA flops (per launch: 6 x or 3 x its algorithmic fp32 flops 2 M N K) / summed "
                "HIP-event time of the class, against the dense 16-bit peak; fp32_equivalent = the algorithmic flops per second, "
                "next to the 157.3 TFLOP/s of the fp32 MFMA pipe this mode replaces")
            roof["fp32_equivalent"] = {"achieved": fl / sec / 1e12, "fp32_mfma_peak": MFMA_F32_PEAK_TFLOPS,
                                       "ratio": fl / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s"}
        cls_bytes = sum(b[4] for b in per.values() if b[0] == dom)
        roof["vs_hbm_algorithmic"] = {"achieved": cls_bytes / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": cls_bytes / sec / 1e9 / HBM_PEAK_GBS,
                                      "what": "the class's algorithmic stored-state bytes (every operand / output "
                                              "array of every step once) / the same time"}
        if dom == "mlp_chain":
            # the class is several instantiations in a rocprofv3 summary: sum them when comparing the average duration
            roof["kernel_names"] = (
                ["mlp_chain_kernel<64|32, 2> (every chain launch of the bf16x3 mode)"] if x3 else
                ["mlp_chain_tq_kernel<0|1|2> (UDF sweeps, > 16384 points, fp32)",
                 "mlp_chain_kernel<64|32, 0|1|3> (all other chain launches)"])
        # every launch class of the step on its own.  MFMA classes: instantiation + sweep + size, launches, algorithmic
        # GFLOP, summed HIP-event time, and BOTH roofs -- MFMA (fp32 157.3 TFLOP/s, or 2.5 PFLOP/s for the 16-bit pipe) and
        # HBM (the launch's algorithmic stored-state bytes, every operand / output array once, against 8 TB/s); "binding"
        # names the larger of the two floors, i.e. the roof that launch could at best run into.  HBM-bound classes
        # (composite, upsample, the blending gathers): algorithmic bytes against 8 TB/s and their unit rate.
        pk, by_bind = [], {"mfma": 0.0, "hbm": 0.0}
        for detail, (name, cnt, f, t, by, fxx) in sorted(per.items(), key=lambda kv: -kv[1][3]):
            if f <= 0:
                unit = {"patch_blend": "taps", "pixel_blend": "taps"}.get(name, "samples")
                pk.append({"kernel": detail, "class": name, "launches": cnt, "us": t * 1e6, "algorithmic_mb": by / 1e6,
                           "gbs": by / t / 1e9, "frac_hbm": by / t / 1e9 / HBM_PEAK_GBS, "binding": "hbm",
                           "frac_of_binding_roof": by / t / 1e9 / HBM_PEAK_GBS, unit: -f, unit + "_per_s": -f / t})
                continue
            # executed flops against the peak of the pipe the launch runs on (bf16x3 / mixed16: the dense 16-bit peak)
            fx = fxx
            t_mfma, t_hbm = fx / (peak * 1e12), by / (HBM_PEAK_GBS * 1e9)
            bind = "hbm" if t_hbm > t_mfma else "mfma"
            if name == dom:
                by_bind[bind] += t
            pk.append({"kernel": detail, "class": name, "launches": cnt, "gflop": f / 1e9, "executed_gflop": fx / 1e9,
                       "us": t * 1e6, "tflops": fx / t / 1e12, "frac_mfma": fx / t / 1e12 / peak,
                       "fp32_equivalent_tflops": f / t / 1e12,
                       "algorithmic_mb": by / 1e6, "gbs": by / t / 1e9, "frac_hbm": by / t / 1e9 / HBM_PEAK_GBS,
                       "binding": bind, "frac_of_binding_roof": max(t_mfma, t_hbm) / t})
        roof["per_kernel"] = pk
        # every launch's algorithmic bytes (each operand / output array once), summed over the step: the saved state of the
        # double backward is what the step moves (SURVEY 8(d) budgets no HBM bytes for the MLP class)
        roof["hbm_gb_per_step"] = sum(e["algorithmic_mb"] for e in pk) / 1e3
        roof["bytes_per_core_sample"] = sum(e["algorithmic_mb"] for e in pk) * 1e6 / (rays_per_gpu * s_core)
        # `bound` of the class = the roof that binds most of its launch time; the per-launch `binding` is the precise statement
        roof["bound"] = "hbm" if by_bind["hbm"] > by_bind["mfma"] else "mfma"
        roof["class_ms_by_binding_roof"] = {k: v * 1e3 for k, v in by_bind.items()}
        kernels = {k: ({"launches": v[0], "gflop": v[1] / 1e9, "ms": v[2] * 1e3, "tflops": v[1] / v[2] / 1e12,
                        "executed_tflops": v[3] / v[2] / 1e12} if v[1] > 0 else
                       {"launches": v[0], "ms": v[2] * 1e3})
                   for k, v in agg.items()}
        roof["traffic"] = pmc_traffic(dom, args.workload, precision)
        roof["traffic_source"] = getattr(pmc_traffic, "source", None) if roof["traffic"] else None
        if roof["traffic"]:
            roof["traffic_stale"] = bool(getattr(pmc_traffic, "stale", True))
            roof["traffic_recorded_on_source_digest"] = getattr(pmc_traffic, "recorded_on", None)
            roof["traffic_note"] = ("PMC passes are recorded beforehand over this command (rocprofv3 cannot wrap the timed "
                                    "process from inside); the file is keyed on workload + precision + round")
        if precision != "fp32" and roof["traffic"]:
            # priced against BOTH roofs: the HBM side = measured bytes per launch / average launch time
            gbs = roof["traffic"] / (sec / n) / 1e9
            roof["vs_hbm"] = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
        if roof["bound"] == "hbm":
            # most of the class's launch time belongs to launches whose floor is HBM (config 5's 16-bit mode: bf16 state of
            # 262 144 points per sweep): the headline numbers of the object are the HBM view -- PMC traffic when recorded, the
            # algorithmic bytes otherwise -- and the matrix-pipe view moves to `vs_mfma`
            roof["vs_mfma"] = {k: roof[k] for k in ("achieved", "peak", "unit", "frac")}
            h = roof.get("vs_hbm") or roof["vs_hbm_algorithmic"]
            roof.update({"achieved": h["achieved"], "peak": h["peak"], "unit": h["unit"], "frac": h["frac"],
                         "bytes": "PMC traffic per launch" if roof.get("vs_hbm") else "algorithmic stored-state bytes"})
        return roof, kernels

    inst = None
    if not args.no_roofline:
        inst = instrumented(args.precision)
    if rank == 0 and inst is not None:
        result["roofline"], result["kernels"] = inst
        pw = result.get("power") or {}
        if result["roofline"].get("bound") == "mfma" and result["roofline"].get("unit") == "TFLOP/s" and args.precision != "fp32":
            # what the matrix pipe ALONE sustains on this card at its 1400 W package limit (a loop of nothing but
            # v_mfma_f32_32x32x16_bf16, two waves per SIMD; by operand bit activity) -- a recorded measurement, not re-run here
            r = result["roofline"]
            r["sustained_peak_recorded"] = {"tflops_range": [1311.0, 1721.0], "frac_range": [r["achieved"] / 1721.0, r["achieved"] / 1311.0],
                                            "source": "profiles/r05_mfma_bf16_power.txt (scripts/ubench/mfma_bf16_power.hip)",
                                            "what": "`peak` above is the nominal 2500 TFLOP/s at 2400 MHz, which the power limit does "
                                                    "not let a saturated pipe hold (1.2-1.6 GHz)"}
        if pw.get("sclk_mhz_avg") and result["roofline"].get("bound") == "mfma":
            # the step runs against the package power limit: the matrix pipe's peak at the clock the card held
            r = result["roofline"]
            r["at_sampled_clock"] = {"sclk_mhz": pw["sclk_mhz_avg"], "peak": r["peak"] * pw["sclk_mhz_avg"] / 2400.0,
                                     "frac": r["achieved"] / (r["peak"] * pw["sclk_mhz_avg"] / 2400.0),
                                     "what": "peak scaled from 2400 MHz to the step-average shader clock of `power` (the heavy "
                                             "launches run below that average)"}
        # ---- fused composite kernel alone (HBM roof) ----
        try:
            result["roofline_composite"] = composite_roofline(dev, rays_per_gpu, s_core)
        except Exception as ex:  # pragma: no cover
            result["roofline_composite"] = {"error": repr(ex)}

    # ---- secondary leg: the EXACT fp32 kernels (v_mfma_f32_32x32x2_f32) on the same trainer, batch and process -- the
    # conservative number beside the emulated-fp32 headline, timed by the same clock: its own captured step, 3 warm-up
    # replays, the median of three windows of min(steps, 10) steps, and its own instrumented step for the fraction of the fp32
    # pipe's peak
    if world == 1 and args.precision == "bf16x3" and not args.no_fp32_leg:
        try:
            mlp.set_precision("fp32")
            g32 = GraphedStep(tr, eager_steps=1) if args.graph else None
            step32 = (lambda: g32(batch, **step_kw)) if (g32 is not None and g32.enabled) else (lambda: tr.step(batch, **step_kw))
            for _ in range(5):                 # 1 eager + capture + replays
                step32()
            si32 = g32.static_inputs() if (g32 is not None and g32.enabled) else None
            if si32 is not None:               # replay from the capture's own input tensors (no per-step input copies)
                b32, bl32 = si32
                kw32 = dict(step_kw)
                if bl32 is not None:
                    kw32["blend"] = bl32
                step32 = lambda: g32(b32, **kw32)
                step32()
            n32 = min(args.steps, 10)
            gc.collect()                       # (the leg's set-up left garbage: keep a collector pass out of its windows)
            w32 = sorted(timed_window(step32, n32) for _ in range(3))
            dt32 = w32[1]                      # median of three windows (one 50 ms window caught a 17 ms host stall once)
            leg = {"ms_per_step": dt32 / n32 * 1e3, "value": rays_per_gpu * s_core / (dt32 / n32), "unit": "ray-samples/s",
                   "steps": n32, "windows": 3, "window_ms": [w / n32 * 1e3 for w in w32], "dtype": DTYPE["fp32"],
                   "launch": "HIP graph replay" if (g32 is not None and g32.replays > 0) else "eager launches"}
            if not args.no_roofline:
                r32, k32 = instrumented("fp32")
                m32 = r32.get("vs_mfma", r32)          # (the fraction of the fp32 MFMA roof, whatever binds the class)
                leg.update({"frac": m32["frac"], "achieved": m32["achieved"], "peak": m32["peak"], "unit_roof": "TFLOP/s",
                            "kernel": r32["kernel"], "kernels": k32})
            result["fp32_exact"] = leg
        except Exception as ex:  # pragma: no cover - the headline above must still be reported
            result["fp32_exact"] = {"error": repr(ex)}
        finally:
            mlp.set_precision(args.precision)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"], psnr = cpu_baseline(args.workload, dev=dev, precision=args.precision, n_rays=rays_per_gpu)
            if psnr is not None:
                result["psnr_vs_ref"] = psnr
        except Exception as ex:  # pragma: no cover - the GPU numbers above must still be reported
            result["cpu_baseline"] = {"error": repr(ex)}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        barrier()              # rank 0 did the extra single-rank measurements; leave the group together
        dist.destroy_process_group()


def pmc_traffic(kernel, workload, precision="fp32"):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes over this same command (FETCH_SIZE and
    WRITE_SIZE, separate passes, scripts/pmc_traffic.sh; committed under profiles/).  rocprofv3 cannot wrap the
    timed process from inside, so the counters are collected beforehand; corrected as MI355X_MICROARCH.md
    prescribes and as calibrated in DESIGN.md section 5: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
    if (workload, precision) == ("dtu_scan24_512x128", "bf16x3"):
        names = ["r%02d_traffic_%s_bf16x3.json" % (r, kernel) for r in (6, 5, 4)]
    elif (workload, precision) == ("dtu_scan24_512x128", "fp32"):
        names = ["r%02d_traffic_%s.json" % (r, kernel) for r in (3, 2, 1)]
    elif (workload, precision) == ("dtu_scan24_1024x256", "mixed16"):
        names = ["r%02d_traffic_%s_cfg5_mixed16.json" % (r, kernel) for r in (6, 5, 3, 2)]
    elif (workload, precision) == ("garment_blend_1024x128", "bf16x3"):
        names = ["r%02d_traffic_%s_garment_bf16x3.json" % (r, kernel) for r in (6, 5)]
    else:
        return None
    path = next((q for q in (os.path.join(ROOT, "profiles", nm) for nm in names) if os.path.exists(q)), None)
    if path is None:
        return None
    pmc_traffic.source = os.path.relpath(path, ROOT) + " (rocprofv3 PMC passes recorded beforehand, scripts/pmc_traffic.sh)"
    try:
        d = json.load(open(path))
        from neuraludf_amd import build as _b
        rec = d.get("_recorded_on") or {}
        # recorded on other kernel sources than the ones this tree holds (or before round 6, when files carried no digest)
        pmc_traffic.stale = rec.get("source_digest") != _b.source_digest(kernel)
        pmc_traffic.recorded_on = rec.get("source_digest")
        f = w = n = 0
        for name, c in d.items():
            if kernel not in name or name.startswith("_"):
                continue
            f += sum(c["FETCH_SIZE"]["list_KB"]) * c["FETCH_SIZE"]["n"] / max(1, len(c["FETCH_SIZE"]["list_KB"]))
            w += sum(c["WRITE_SIZE"]["list_KB"]) * c["WRITE_SIZE"]["n"] / max(1, len(c["WRITE_SIZE"]["list_KB"]))
            n += c["FETCH_SIZE"]["n"]
        return (2.0 * f + w) * 1024.0 / n if n else None
    except Exception:
        return None


def composite_roofline(dev, N, S, reps=50):
    """time nudf_composite_fwd alone on resident inputs at the step's own size and at two larger ones.  Only a working
    set beyond the 256 MB Infinity Cache is labelled "hbm" (the launches re-read the same buffers): the step's own
    size is L2-resident and launch-bound, 8192 x 256 (101 MB) is MALL-resident."""
    from neuraludf_amd.models.udf_renderer_blending import _CompositeFn
    out = {}
    for (n, s) in [(N, S), (8192, 256), (32768, 256)]:
        g = torch.Generator(device="cpu").manual_seed(0)
        z = torch.sort(torch.rand(n, s, generator=g) * 2 + 1.5, -1)[0].to(dev)
        ro = torch.randn(n, 3, generator=g).to(dev)
        rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        udf = (torch.rand(n, s, generator=g) * 0.3).to(dev).requires_grad_(True)
        grad = torch.randn(n, s, 3, generator=g).to(dev).requires_grad_(True)
        col = torch.rand(n, s, 3, generator=g).to(dev).requires_grad_(True)
        cb = torch.rand(n, s, 3, generator=g).to(dev).requires_grad_(True)
        scal = torch.tensor([64.0, 128.0, 20.0], device=dev)
        sd = torch.tensor([2.0 / 64], device=dev)
        c = dict(s_nominal=s, cos_anneal=1.0, flip_saturation=1.0, use_norm_grad=False, sparse_scale=25000.0,
                 diagnostics=False)
        with torch.no_grad():
            for _ in range(3):
                _CompositeFn.apply(c, ro, rd, z, sd, None, udf, grad, col, cb, None, None, None, scal)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # isolate the kernel from the allocator: time the ctypes launches only
            from neuraludf_amd._lib import Composite, call, ptr
            from neuraludf_amd.models.udf_renderer_blending import _fill_composite
            a = Composite()
            a.rays_o, a.rays_d, a.z, a.udf, a.grad = ptr(ro), ptr(rd), ptr(z), ptr(udf.detach()), ptr(grad.detach())
            a.color, a.color_base = ptr(col.detach()), ptr(cb.detach())
            a.scal, a.sample_dist = ptr(scal), ptr(sd)
            _fill_composite(a, c, n, s, 0)
            bufs = [torch.empty(n, s, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev),
                    torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev),
                    torch.empty(n, device=dev), torch.empty(5, device=dev)]
            (a.weights, a.out_color, a.out_color_base, a.out_depth, a.out_normals, a.out_wsum, a.out_wsum_all,
             a.sums) = [ptr(b) for b in bufs]
            ws = torch.empty(5 * ((n + 3) // 4), device=dev)
            a.ws = ptr(ws)
            e0.record()
            for _ in range(reps):
                call("nudf_composite_fwd", a)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            # the weights-first form of no-grad rendering: the same launch without the two colour arrays
            # -- their compositing sums are taken inside the colour network's epilogues (NudfChainStep.row_w)
            a.color = a.color_base = None
            for _ in range(3):
                call("nudf_composite_fwd", a)
            e0.record()
            for _ in range(reps):
                call("nudf_composite_fwd", a)
            e1.record()
            torch.cuda.synchronize()
            us_w = e0.elapsed_time(e1) / reps * 1e3
        bytes_fwd = 48.0 * n * s + 68.0 * n
        level = "hbm" if bytes_fwd > 256e6 else ("mall (256 MB Infinity Cache resident)" if bytes_fwd > 32e6 else "l2 / launch latency")
        out[f"fwd_{n}x{s}"] = {"bound": level, "achieved": bytes_fwd / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": bytes_fwd / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               "avg_launch_us": us, "algorithmic_bytes": bytes_fwd}
        bytes_w = 24.0 * n * s + 44.0 * n          # z, udf, grad in; weights out; per ray: o, d in, depth / normals / sums out
        out[f"fwd_weights_first_{n}x{s}"] = {
            "bound": "hbm" if bytes_w > 256e6 else ("mall (256 MB Infinity Cache resident)" if bytes_w > 32e6 else "l2 / launch latency"),
            "achieved": bytes_w / (us_w * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_w / (us_w * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_launch_us": us_w, "algorithmic_bytes": bytes_w,
            "what": "nudf_composite_fwd with color = color_base = NULL: 24 B/sample instead of 48; the colour sums ride in the "
                    "colour network's sigmoid-head epilogues (12 B/sample + 68 B/ray of the fused pair against 72 B/sample + "
                    "68 B/ray with the colours written and read back)"}
    return out


if __name__ == "__main__":
    main()
