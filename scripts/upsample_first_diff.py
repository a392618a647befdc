"""close row a3 with evidence.  For every arithmetic variant of the up-sampling kernel (incl.
NUDF_UP_SLEEF) replay round 0 of BASELINE config 2's hierarchical sampling on the oracle's own (z, udf) and, for the rays
whose new samples still move by > 1e-4, compare the kernel's per-section intermediates (NudfUpsample.dbg) with the
oracle's tensors stage by stage: which stage differs FIRST on each such ray, and by how many ulp.

    python scripts/upsample_first_diff.py [out.txt]        (GPU box; the oracle runs on the box's host CPU)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import build_modules, perturb_, state_dicts, oracle_nets  # noqa: E402
from oracle import udf_oracle as O  # noqa: E402

STAGES = ["cos_val", "vis", "alpha_plus", "alpha_minus", "alpha", "weights", "cdf"]


def ulps(a, b):
    """|a - b| in units of the last place of b (fp32), elementwise."""
    ai = a.contiguous().view(torch.int32).to(torch.int64)
    bi = b.contiguous().view(torch.int32).to(torch.int64)
    ai = torch.where(ai < 0, -(ai & 0x7fffffff), ai)
    bi = torch.where(bi < 0, -(bi & 0x7fffffff), bi)
    return (ai - bi).abs()


def main():
    from neuraludf_amd.models import fields, udf_renderer_blending as urb
    dev = torch.device("cuda:0")
    fx = dict(np.load(os.path.join(ROOT, "tests", "golden", "ref_cfg2_full.npz")))
    KW = dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0)
    mods = perturb_(build_modules(fields, seed=0))
    sds = state_dicts(mods)
    for m in mods.values():
        m.to(dev)
    rend = urb.UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **KW)
    rays = {k[4:]: torch.from_numpy(v).to(dev) for k, v in fx.items() if k.startswith("ray_")}
    cpu = {k: v.cpu() for k, v in rays.items()}
    cfg = O.RenderCfg(**{k: v for k, v in KW.items() if k != "perturb"})
    on = oracle_nets(sds)
    N = 512
    trace = []
    z0, _, sd = O.coarse_z(cfg, cpu["near"], cpu["far"], N)
    with torch.no_grad():
        O.importance_sample(on, cfg, cpu["rays_o"], cpu["rays_d"], z0, sd, trace)
    sdd = torch.tensor([sd], device=dev)
    lines = ["torch CPU capability of this host: %s ; %s" % (torch.backends.cpu.get_cpu_capability(),
                                                           open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"))]
    variants = [("serial double scans, contraction off (round-3 default)", 512 | 1024),
                ("  + Sleef sigmoid (NUDF_UP_SLEEF)", 512 | 1024 | 2048),
                ("  + Sleef sigmoid + correctly rounded exp (NUDF_UP_EXPCR)", 512 | 1024 | 2048 | 4096),
                ("  + correctly rounded exp only", 512 | 1024 | 4096)]
    for ri, t in enumerate(trace):
        if t["kind"] != "unbias":
            continue
        M = t["z"].shape[1]
        k = t["z_new"].shape[1]
        od = {}
        with torch.no_grad():
            z_o = O.up_sample_unbias(cpu["rays_o"], cpu["rays_d"], t["z"], t["udf"], sd, k, t["inv_s"], t["beta"], t["gamma"],
                                     dbg=od)
        assert torch.equal(z_o, t["z_new"])
        for name, flags in variants:
            urb.UPSAMPLE_FLAGS = flags
            dbg = torch.zeros(N, 7, M, device=dev)
            with torch.no_grad():
                z_k, _ = rend._upsample(rays["rays_o"], rays["rays_d"], t["z"].to(dev), t["udf"].to(dev), sdd, k, 0, t["inv_s"],
                                        t["beta"], t["gamma"], dbg=dbg)
            dz = (z_k.cpu() - t["z_new"]).abs().max(dim=1)[0]
            moved = dz > 1e-4
            d = dbg.cpu()
            lines.append(f"round {ri} (M = {M}, +{k}) | {name} (flags {flags}): {int(moved.sum())} / {N} rays moved > 1e-4, "
                         f"{int((dz == 0).sum())} bit-identical")
            first = {}
            worst = {s: 0 for s in STAGES}
            nbad = {s: 0 for s in STAGES}
            for si, s in enumerate(STAGES):
                w = M if s == "cdf" else M - 1
                u = ulps(d[:, si, :w], od[s][:, :w].float())
                nbad[s] = int((u.max(dim=1)[0] > 0).sum())
                worst[s] = int(u.max())
                for r in torch.nonzero(moved).flatten().tolist():
                    if r not in first and int(u[r].max()) > 0:
                        first[r] = (s, int(u[r].max()), int(u[r].argmax()))
            lines.append("     rays with any difference per stage (all 512 rays): " +
                         ", ".join(f"{s} {nbad[s]} (max {worst[s]} ulp)" for s in STAGES))
            hist = {}
            for r, (s, u, i) in first.items():
                hist.setdefault(s, []).append(u)
            lines.append("     first differing stage on the MOVED rays: " +
                         (", ".join(f"{s}: {len(v)} rays (<= {max(v)} ulp)" for s, v in hist.items()) or "none") +
                         f" ; moved rays identical in every dumped stage: {int(moved.sum()) - len(first)}")
            # the moved rays' character: how much of their pdf sits on the 1e-5 floor
            if moved.any():
                wsum = od["weights"][moved].sum(dim=1)
                lines.append(f"     moved rays: sum of weights median {float(wsum.median()):.3e} (pdf floor mass {(M - 1) * 1e-5:.1e}), "
                             f"max |dz| {float(dz.max()):.2e}")
        break_after = os.environ.get("NUDF_FD_ROUNDS")
        if break_after and ri + 1 >= int(break_after):
            break
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
