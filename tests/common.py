"""Shared helpers for the tests / golden generator / bench: shipped-conf kwargs, seeded
'trained-like' weights, oracle glue."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# kwargs of the shipped DTU conf (confs/udf_dtu_blending.conf:56-118)
CONF = dict(
    udf=dict(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
             geometric_init=True, weight_norm=True, udf_type="abs"),
    color=dict(d_feature=256, mode="no_normal", d_in=6, d_out=3, d_hidden=128, n_layers=4, weight_norm=True,
               multires_view=4, squeeze_out=True, blending_cand_views=10),
    var=dict(init_val=0.3),
    beta=dict(init_var_beta=0.5, init_var_gamma=0.3, init_var_zeta=0.3, beta_min=0.00005,
              requires_grad_beta=True, requires_grad_gamma=False, requires_grad_zeta=False),
    nerf=dict(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
              use_viewdirs=True),
)


def build_modules(fields_mod, seed=0, udf_type="abs", color_mode="no_normal"):
    """Instantiate the five networks in the runner's order (exp_runner_blending.py:125-129).  `udf_type` does not enter the
    initialisation: the same seed gives the same weights for 'abs', 'square' and 'sdf'."""
    import contextlib
    import io
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        nerf = fields_mod.NeRF(**CONF["nerf"])
        udf = fields_mod.UDFNetwork(**{**CONF["udf"], "udf_type": udf_type})
        var = fields_mod.SingleVarianceNetwork(**CONF["var"])
        cc = dict(CONF["color"])
        if color_mode != "no_normal":     # fields.py:456-461: base input = [pts, n, -n, feat] -> d_in - 3 = 9
            cc.update(mode=color_mode, d_in=12)
        color = fields_mod.ResidualRenderingNetwork(**cc)
        beta = fields_mod.BetaNetwork(**CONF["beta"])
    return dict(nerf=nerf, udf=udf, var=var, color=color, beta=beta)


@torch.no_grad()
def perturb_(mods, seed=1, scale=0.02):
    """Make the geometric-init weights 'trained-like': the init zeroes the positional-encoding
    columns of lin0/lin4, which would hide channel-order bugs.  Deterministic."""
    g = torch.Generator().manual_seed(seed)
    for name in ["nerf", "udf", "color"]:
        for p in mods[name].parameters():
            p.add_(torch.randn(p.shape, generator=g) * scale * (p.abs().mean() + 0.05))
    return mods


def state_dicts(mods):
    return {k: {n: t.detach().clone() for n, t in m.state_dict().items()} for k, m in mods.items()}


def oracle_nets(sds, requires_grad=False, dtype=torch.float32):
    from oracle import udf_oracle as O
    def prep(sd):
        out = {}
        for k, v in sd.items():
            t = v.detach().clone().to(dtype)
            if requires_grad:
                t.requires_grad_(True)
            out[k] = t
        return out
    return O.Nets(udf=prep(sds["udf"]), color=prep(sds["color"]), var=prep(sds["var"]),
                  beta=prep(sds["beta"]), nerf=prep(sds["nerf"]) if "nerf" in sds else None)


def checksum(sd):
    """order-stable float64 checksum of a state dict."""
    s = 0.0
    for i, (k, v) in enumerate(sorted(sd.items())):
        s += float((v.double() * ((i % 7) + 1)).sum())
    return s


def smooth_images(v, h, w, seed=0):
    """band-limited images: bilinear taps then differ by O(1e-6) for sub-1e-4-pixel coordinate differences
    (i.i.d. noise images would turn fp32 coordinate rounding into 1e-4 colour noise in BOTH implementations)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    imgs = torch.zeros(v, 3, h, w)
    for i in range(v):
        for c in range(3):
            f = torch.rand(4, generator=g) * 6 + 1
            imgs[i, c] = 0.5 + 0.25 * torch.sin(f[0] * xx + f[1] * yy + i) + 0.2 * torch.cos(f[2] * xx - f[3] * yy + c)
    return imgs


def weights_vs_reference_up_to_ties(rend, render_again, w_hip, w_ref, n_core, tol=1e-4, tie=2e-5, max_rays=3):
    """Compositing weights of ALL rays against the reference's, where the reference's own arithmetic is discontinuous:
    `vis_mask = (true_cos < 0.01)` (/root/reference/models/udf_renderer_blending.py:399-405) is a hard selection that
    switches a factor of the running visibility product between `1 - alpha_occ` and 1, i.e. the alpha of every LATER
    sample of that ray; where true_cos ties with the threshold to an ulp two fp32 implementations may select differently.

    Contract checked here (instead of exempting such rays wholesale): at most `max_rays` rays contain a weight that differs
    by more than `tol`; on each of them every weight BEFORE the first sample whose true_cos lies within `tie` of the
    threshold agrees to `tol`, i.e. the first difference sits at or behind a tie.  `render_again()` re-renders with
    `rend.diagnostics = True` and returns the result dict (only called when a ray differs).  Returns the [N] bool mask
    of rays without a difference (per-ray outputs downstream of the weights are compared on those) and the tie list."""
    import torch
    w_hip = torch.as_tensor(w_hip).detach().float().cpu()
    w_ref = torch.as_tensor(w_ref).detach().float().cpu()
    bad = (w_hip - w_ref).abs() > tol
    rays = bad.any(dim=1)
    n = int(rays.sum())
    assert n <= max_rays, f"{n} rays differ from the reference's weights by > {tol}"
    ties = []
    if n:
        old = rend.diagnostics
        rend.diagnostics = True
        try:
            with torch.no_grad():
                tc = render_again()["true_cos"].detach().float().cpu().reshape(w_hip.shape[0], -1)
        finally:
            rend.diagnostics = old
        for r in rays.nonzero().flatten().tolist():
            k = int(bad[r].float().argmax())                       # first differing weight
            upto = min(k, n_core - 1)
            near = ((tc[r, :upto + 1] - 0.01).abs() < tie).nonzero().flatten().tolist()
            assert near, (f"ray {r}: weights differ from sample {k} on (max {float((w_hip[r] - w_ref[r]).abs().max()):.2e}) but no "
                          f"true_cos within {tie} of the 0.01 threshold at or before it: "
                          f"{[round(float(x), 6) for x in tc[r, max(0, upto - 3):upto + 1]]}")
            m = near[0]
            # everything in front of the tie agrees (by definition of k >= m this is every sample < m)
            assert float((w_hip[r, :m] - w_ref[r, :m]).abs().max()) <= tol if m > 0 else True
            ties.append((r, m, k, float(tc[r, m])))
    return ~rays, ties
