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


def weights_vs_reference_up_to_ties(rend, render_again, w_hip, w_ref, n_core, tol=1e-4, tie=2e-5, max_rays=3, wrel=1e-4,
                                    udf_tie=1e-6):
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
    # ... and on the rays without such a difference the bound is RELATIVE to the largest weight present (weights of 128-256
    # samples per ray are far below 1: `tol` alone would be loose there)
    if n < w_hip.shape[0]:
        wr = float((w_hip[~rays] - w_ref[~rays]).abs().max() / w_ref[~rays].abs().max().clamp(min=1e-12))
        assert wr < wrel, f"weights on the tie-free rays: {wr:.2e} of the largest reference weight (bar {wrel})"
    ties = []
    if n:
        old = rend.diagnostics
        rend.diagnostics = True
        try:
            with torch.no_grad():
                again = render_again()
                tc = again["true_cos"].detach().float().cpu().reshape(w_hip.shape[0], -1)
                ud = again["udf"].detach().float().cpu().reshape(w_hip.shape[0], -1)
        finally:
            rend.diagnostics = old
        for r in rays.nonzero().flatten().tolist():
            k = int(bad[r].float().argmax())                       # first differing weight
            upto = min(k, n_core - 1)
            # the second hard selection of the reference: udf = |v| (fields.py:184-190), so d udf / d x carries sign(v), and
            # with it true_cos, flip_sign and the choice between alpha_plus and alpha_minus.  A sample ON the surface
            # (|v| within the rounding noise of a 256-term fp32 sum, `udf_tie`) has an implementation-defined sign.
            near = (((tc[r, :upto + 1] - 0.01).abs() < tie) | (ud[r, :upto + 1] < udf_tie)).nonzero().flatten().tolist()
            assert near, (f"ray {r}: weights differ from sample {k} on (max {float((w_hip[r] - w_ref[r]).abs().max()):.2e}) but no "
                          f"true_cos within {tie} of the 0.01 threshold and no udf below {udf_tie} at or before it: true_cos "
                          f"{[round(float(x), 6) for x in tc[r, max(0, upto - 3):upto + 1]]} udf "
                          f"{[float(x) for x in ud[r, max(0, upto - 3):upto + 1]]}")
            m = near[0]
            # everything in front of the tie agrees (by definition of k >= m this is every sample < m)
            assert float((w_hip[r, :m] - w_ref[r, :m]).abs().max()) <= tol if m > 0 else True
            ties.append((r, m, k, float(tc[r, m])))
    return ~rays, ties


# --------------------------------------------------------------------------------------------------------------------- #
# gradient parity: TRUE relative error per tensor ("gradients of the loss w.r.t. every parameter to 1e-3 rel")
# --------------------------------------------------------------------------------------------------------------------- #
def grel(a, b):
    """max|a - b| / max|b| -- no floor at 1: a mean-reduced loss leaves most gradient tensors far below 1, where a clamp
    would turn the bound into an absolute one (an all-zero gradient would pass)."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def grel2(a, b):
    """||a - b||_2 / ||b||_2 -- one cancelling element cannot hide a tensor, one large element cannot carry it."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


_EREF = None


def eref(case):
    """tests/golden/ref_eref.json (make_golden_eref.py): per gradient tensor of a reference fixture, the fp32 REFERENCE's own
    distance from the float64 evaluation of the same loss on the same inputs."""
    global _EREF
    if _EREF is None:
        import json
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eref.json")) as f:
            _EREF = json.load(f)
    return _EREF[case]["tensors"]


def param_grads_vs_reference(mods, fx, case, nets=("udf", "color", "var", "beta"), gtol=1e-3, absent_must_be_zero=True,
                             slack=None):
    """Every parameter gradient of `mods` against the reference's (`fx['grad_<net>_<name>']`, produced by loss.backward()
    in the reference, /root/reference/exp_runner_blending.py:367-375), TRUE relative per tensor in the max norm AND in the
    2-norm:   err < max(gtol, 3 * e_ref32)   where e_ref32 is the fp32 reference's own distance from float64 for that tensor
    and norm (`eref(case)`).  `slack`: {key: bar} for documented discontinuities (a trimmed loss dropping a whole ray).
    Returns a report: worst (key, inf, l2, bar) per network, tensor / float counts, and for tensors of <= 4 elements the
    error against the float64 value itself."""
    er = eref(case)
    rep = {"n": 0, "floats": 0, "worst": {}, "vs64": {}}
    for net in nets:
        worst = None
        for pn, p in mods[net].named_parameters():
            key = f"grad_{net}_{pn}"
            if key not in fx:
                if absent_must_be_zero:
                    assert p.grad is None or float(p.grad.abs().max()) == 0.0, key
                continue
            assert p.grad is not None, key
            e = er[key]
            ri, r2 = grel(p.grad, fx[key]), grel2(p.grad, fx[key])
            bi = max(gtol, 3.0 * e["inf"], (slack or {}).get(key, 0.0))
            b2 = max(gtol, 3.0 * e["l2"], (slack or {}).get(key, 0.0))
            rep["n"] += 1
            rep["floats"] += p.grad.numel()
            if worst is None or ri / bi > worst[1] / worst[3]:
                worst = (key, ri, r2, bi)
            if "g64" in e:
                g64 = torch.tensor(e["g64"], dtype=torch.float64).reshape(p.grad.shape)
                rep["vs64"][key] = (grel(p.grad, g64), e["inf"])
            assert ri < bi, (key, "max-norm relative", ri, "bar", bi, "reference fp32 vs fp64", e["inf"], "|g|max", e["ninf"])
            assert r2 < b2, (key, "2-norm relative", r2, "bar", b2, "reference fp32 vs fp64", e["l2"])
        if worst is not None:
            rep["worst"][net] = worst
    return rep


def grad_report(rep):
    w = "; ".join(f"{net}: {k} inf {ri:.2e} l2 {r2:.2e} (bar {b:.1e})" for net, (k, ri, r2, b) in rep["worst"].items())
    v = "; ".join(f"{k} vs float64 {a:.2e} (reference fp32 vs float64 {b:.2e})" for k, (a, b) in rep["vs64"].items())
    return f"{rep['n']} parameter-gradient tensors ({rep['floats']} floats), TRUE relative worst per network -- {w}.  {v}"
