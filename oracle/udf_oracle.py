"""CPU oracle for the NeuralUDF volume-rendering hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch (CPU, fp32) restatement of
the reference algorithm.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker / the
timed CPU baseline -- never as part of the product path (`neuraludf_amd/`), which
must fail loudly when the HIP library is missing.

Parity pinning: the reference (xxlong0/NeuralUDF) ships no tests or golden
vectors.  The oracle is pinned instead against outputs of the reference code
itself, imported from /root/reference in the build container:
  * tests/golden/make_golden.py  runs the reference and commits the fixtures,
  * tests/test_oracle_golden.py  checks this file against those fixtures,
  * tests/test_oracle_vs_reference.py re-runs the live comparison when
    /root/reference is present.

Weights are passed as flat dicts keyed exactly like the reference modules'
`state_dict()` (`lin0.weight_g`, `lin0.weight_v`, `lin0.bias`, `lin_base0.*`,
`pts_linears.0.weight`, ..., `variance`, `beta`, `gamma`).  Every function cites
the reference file:line (paths relative to /root/reference) it follows.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------- #
# positional encoding                       models/embedder.py:6-51
# --------------------------------------------------------------------------- #
def posenc(x: Tensor, n_freq: int) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...]  (embedder.py:15-36)."""
    if n_freq <= 0:
        return x
    out = [x]
    freqs = 2.0 ** torch.linspace(0.0, n_freq - 1, n_freq)          # embedder.py:22-23
    for f in freqs:
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def wn_weight(sd: SD, name: str) -> Tensor:
    """weight_norm: W = g * v / ||v||_row   (torch.nn.utils.weight_norm, dim=0;
    applied at fields.py:175-176, 433-446).  Plain Linear falls back to `.weight`."""
    if name + ".weight_v" in sd:
        v = sd[name + ".weight_v"]
        g = sd[name + ".weight_g"]
        return torch._weight_norm(v, g, 0)      # same primitive the weight_norm hook calls
    return sd[name + ".weight"]


def softplus100(x: Tensor) -> Tensor:
    """nn.Softplus(beta=100), threshold 20   (fields.py:180)."""
    return F.softplus(x, beta=100.0, threshold=20.0)


# --------------------------------------------------------------------------- #
# UDF network                                models/fields.py:115-231
# --------------------------------------------------------------------------- #
@dataclass
class UDFCfg:
    d_in: int = 3
    d_out: int = 257
    d_hidden: int = 256
    n_layers: int = 8
    skip_in: tuple = (4,)
    multires: int = 6
    scale: float = 1.0
    udf_type: str = "abs"

    @property
    def n_lin(self) -> int:
        return self.n_layers + 1


def udf_forward(sd: SD, x: Tensor, cfg: UDFCfg = UDFCfg()) -> Tensor:
    """UDFNetwork.forward  (fields.py:192-211): [P,3] -> [P,d_out] (udf | features)."""
    inp = x * cfg.scale
    emb = posenc(inp, cfg.multires)
    h = emb
    for l in range(cfg.n_lin):
        if l in cfg.skip_in:
            h = torch.cat([h, emb], 1) / np.sqrt(2)                  # fields.py:202-203
        h = F.linear(h, wn_weight(sd, f"lin{l}"), sd[f"lin{l}.bias"])
        if l < cfg.n_lin - 1:
            h = softplus100(h)
    head = h[:, :1]
    if cfg.udf_type == "abs":                                        # fields.py:184-190
        head = head.abs()
    elif cfg.udf_type == "square":
        head = head ** 2
    return torch.cat([head / cfg.scale, h[:, 1:]], -1)


def udf_gradient(sd: SD, x: Tensor, cfg: UDFCfg = UDFCfg(), create_graph: bool = True) -> Tensor:
    """UDFNetwork.gradient  (fields.py:219-231): d udf / d x by autograd, [P,3]."""
    with torch.enable_grad():
        xg = x.detach().requires_grad_(True)
        y = udf_forward(sd, xg, cfg)[:, :1]
        (g,) = torch.autograd.grad(y, xg, torch.ones_like(y), create_graph=create_graph,
                                   retain_graph=True)
    return g


def udf_gradient_analytic(sd: SD, x: Tensor, cfg: UDFCfg = UDFCfg()) -> Tensor:
    """Same quantity as `udf_gradient`, by an explicit reverse sweep (the form the
    HIP kernels implement); used to cross-check the kernels' algebra on CPU."""
    inp = x * cfg.scale
    emb = posenc(inp, cfg.multires)
    h = emb
    sig = []
    Ws = []
    for l in range(cfg.n_lin):
        if l in cfg.skip_in:
            h = torch.cat([h, emb], 1) / np.sqrt(2)
        W = wn_weight(sd, f"lin{l}")
        Ws.append(W)
        a = F.linear(h, W, sd[f"lin{l}.bias"])
        if l < cfg.n_lin - 1:
            sig.append(torch.where(a * 100 > 20, torch.ones_like(a), torch.sigmoid(100 * a)))
            h = softplus100(a)
        else:
            h = a
    s = (torch.sign(h[:, :1]) if cfg.udf_type == "abs" else
         2.0 * h[:, :1] if cfg.udf_type == "square" else torch.ones_like(h[:, :1]))      # udf_out' (fields.py:184-190)
    delta = s * Ws[-1][0:1, :] / cfg.scale                           # d udf / d h_last
    d_emb = torch.zeros_like(emb)
    for l in range(cfg.n_lin - 2, -1, -1):
        da = delta * sig[l]
        delta = da @ Ws[l]
        if l in cfg.skip_in:
            delta = delta / np.sqrt(2)
            d_emb = d_emb + delta[:, -emb.shape[1]:]
            delta = delta[:, :-emb.shape[1]]
    d_emb = d_emb + delta
    # chain through the embedding: d/dx [x, sin(f x), cos(f x)]
    g = d_emb[:, :3].clone()
    freqs = 2.0 ** torch.linspace(0.0, cfg.multires - 1, cfg.multires)
    for k, f in enumerate(freqs):
        ds = d_emb[:, 3 + 6 * k: 6 + 6 * k]
        dc = d_emb[:, 6 + 6 * k: 9 + 6 * k]
        g = g + f * (ds * torch.cos(inp * f) - dc * torch.sin(inp * f))
    return g * cfg.scale


# --------------------------------------------------------------------------- #
# colour network                             models/fields.py:400-495
# --------------------------------------------------------------------------- #
@dataclass
class ColorCfg:
    d_feature: int = 256
    mode: str = "no_normal"
    d_in: int = 6
    d_out: int = 3
    d_hidden: int = 128
    n_layers: int = 4
    multires_view: int = 4
    blending_cand_views: int = 10

    @property
    def n_lin(self) -> int:
        return self.n_layers + 1


def color_forward(sd: SD, pts: Tensor, normals: Tensor, dirs: Tensor, feat: Tensor,
                  cfg: ColorCfg = ColorCfg()):
    """ResidualRenderingNetwork.forward (fields.py:452-495)
    -> (color_base [P,3], color [P,3], blending logits [P,10])."""
    vd = posenc(dirs, cfg.multires_view) if cfg.mode != "no_view_dir" else dirs
    if cfg.mode == "no_normal":
        x = torch.cat([pts, feat], -1)
    else:
        n = normals.detach()
        x = torch.cat([pts, n, -1 * n, feat], -1)
    hidden = None
    for l in range(cfg.n_lin):
        x = F.linear(x, wn_weight(sd, f"lin_base{l}"), sd[f"lin_base{l}.bias"])
        if l < cfg.n_lin - 1:
            x = F.relu(x)
        if l == cfg.n_lin - 2:
            hidden = x                                               # fields.py:472-473
    color_base = torch.sigmoid(x[:, :cfg.d_out])
    x = torch.cat([vd, color_base, hidden], -1)
    for l in range(cfg.n_lin):
        x = F.linear(x, wn_weight(sd, f"lin{l}"), sd[f"lin{l}.bias"])
        if l < cfg.n_lin - 1:
            x = F.relu(x)
    color = torch.sigmoid(x[:, :cfg.d_out])
    return color_base, color, x[:, cfg.d_out:]


def rendering_forward(sd: SD, pts: Tensor, normals: Tensor, dirs: Tensor, feat: Tensor, mode: str = "idr",
                      multires_view: int = 4, d_out: int = 3, squeeze_out: bool = True):
    """RenderingNetwork.forward (fields.py:364-397), the plain colour MLP -> (color [P,d_out], extra columns)."""
    vd = posenc(dirs, multires_view) if (multires_view > 0 and mode != "no_view_dir") else dirs
    n = normals.detach() if normals is not None else None
    if mode == "idr":
        x = torch.cat([pts, vd, n, -1 * n, feat], -1)
    elif mode == "no_view_dir":
        x = torch.cat([pts, n, -1 * n, feat], -1)
    else:
        x = torch.cat([pts, vd, feat], -1)
    n_lin = len([k for k in sd if k.endswith(".bias")])
    for l in range(n_lin):
        w = wn_weight(sd, f"lin{l}") if f"lin{l}.weight_g" in sd else sd[f"lin{l}.weight"]
        x = F.linear(x, w, sd[f"lin{l}.bias"])
        if l < n_lin - 1:
            x = F.relu(x)
    color = torch.sigmoid(x[:, :d_out]) if squeeze_out else x[:, :d_out]
    return color, x[:, d_out:]


# --------------------------------------------------------------------------- #
# background NeRF                            models/fields.py:541-642
# --------------------------------------------------------------------------- #
@dataclass
class NerfCfg:
    D: int = 8
    W: int = 256
    d_in: int = 4
    d_in_view: int = 3
    multires: int = 10
    multires_view: int = 4
    skips: tuple = (4,)


def nerf_forward(sd: SD, pts: Tensor, dirs: Tensor, cfg: NerfCfg = NerfCfg()):
    """NeRF.forward with use_viewdirs=True (fields.py:599-628) -> (sigma_raw [P,1], rgb [P,3])."""
    e = posenc(pts, cfg.multires)
    ev = posenc(dirs, cfg.multires_view)
    h = e
    for i in range(cfg.D):
        h = F.relu(F.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"]))
        if i in cfg.skips:
            h = torch.cat([e, h], -1)                                # fields.py:609-610
    alpha = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
    feat = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
    h = torch.cat([feat, ev], -1)
    h = F.relu(F.linear(h, sd["views_linears.0.weight"], sd["views_linears.0.bias"]))
    rgb = F.linear(h, sd["rgb_linear.weight"], sd["rgb_linear.bias"])
    return alpha, rgb


# --------------------------------------------------------------------------- #
# scalar networks                            models/fields.py:645-700
# --------------------------------------------------------------------------- #
def inv_s_of(var_sd: SD) -> Tensor:
    """SingleVarianceNetwork.forward -> exp(10*variance), clipped at the call site
    (fields.py:654-655, udf_renderer_blending.py:373)."""
    return torch.exp(var_sd["variance"] * 10.0).reshape(1, 1).clip(1e-6, 1e6)


def beta_of(beta_sd: SD, beta_min: float = 0.00005) -> Tensor:
    """BetaNetwork.get_beta + call-site clip (fields.py:674-675, renderer :376)."""
    return torch.exp(beta_sd["beta"] * 10).clip(0, 1.0 / beta_min).clip(1e-6, 1e6)


def gamma_of(beta_sd: SD) -> Tensor:
    """BetaNetwork.get_gamma + call-site clip (fields.py:677-678, renderer :377)."""
    return torch.exp(beta_sd["gamma"] * 10).clip(1e-6, 1e6)


# --------------------------------------------------------------------------- #
# renderer math                              models/udf_renderer_blending.py
# --------------------------------------------------------------------------- #
def udf2logistic(udf, inv_s, gamma=20.0, abs_cos=1.0):
    """udf_renderer_blending.py:151-159 (cos_anneal_ratio=None on every call site)."""
    e = torch.exp(-inv_s * udf)
    return abs_cos * inv_s * e / (1 + e) ** 2 * gamma


def sdf2alpha(sdf, true_cos, dists, inv_s, cos_anneal_ratio=None, kind="numerical"):
    """udf_renderer_blending.py:292-325: 'numerical' (:308-320, the shipped setting) or 'theorical' (:321-323)."""
    if cos_anneal_ratio is not None:
        it = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio)
               + F.relu(-true_cos) * cos_anneal_ratio)
    else:
        it = true_cos
    if kind == "theorical":
        raw = it.abs() * inv_s * (1 - torch.sigmoid(sdf * inv_s))
        return 1.0 - torch.exp(-F.relu(raw) * dists)
    assert kind == "numerical", kind
    nxt = sdf + it * dists * 0.5
    prv = sdf - it * dists * 0.5
    pc = torch.sigmoid(prv * inv_s)
    nc = torch.sigmoid(nxt * inv_s)
    return ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)


def excl_cumprod(x: Tensor) -> Tensor:
    """cumprod(cat([1, x]))[:, :-1] idiom (udf_renderer_blending.py:183, 249-251, 261-262, 407-410, 508)."""
    one = torch.ones_like(x[:, :1])
    return torch.cumprod(torch.cat([one, x], -1), -1)[:, :-1]


def sample_pdf_det(bins: Tensor, weights: Tensor, k: int, dbg: Optional[dict] = None) -> Tensor:
    """sample_pdf(..., det=True)  (udf_renderer_blending.py:66-104).  `dbg`: receives the intermediates (tests only)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if dbg is not None:
        dbg.update(w_sum=torch.sum(w, -1), pdf=pdf, cdf=cdf)
    u = torch.linspace(0.5 / k, 1.0 - 0.5 / k, steps=k).expand(list(cdf.shape[:-1]) + [k]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    c0 = torch.gather(cdf, 1, below)
    c1 = torch.gather(cdf, 1, above)
    b0 = torch.gather(bins, 1, below)
    b1 = torch.gather(bins, 1, above)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    t = (u - c0) / den
    return b0 + t * (b1 - b0)


def up_sample_unbias(rays_o, rays_d, z, udf, sample_dist, k, inv_s, beta, gamma, kind="numerical",
                     dbg: Optional[dict] = None) -> Tensor:
    """udf_renderer_blending.py:197-272 -> new z [N,k].  `dbg`: receives the per-section intermediates (tests only)."""
    n, m = z.shape
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., :, None]
    radius = torch.linalg.norm(pts, ord=2, dim=-1)
    inside = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)
    udf = udf.reshape(n, m)
    d_raw = torch.cat([z[:, 1:] - z[:, :-1], torch.full((n, 1), float(sample_dist))], -1)
    pu, nu = udf[:, :-1], udf[:, 1:]
    pz, nz = z[:, :-1], z[:, 1:]
    mid_udf = (pu + nu) * 0.5
    dists = nz - pz
    true_cos = (nu - pu) / (nz - pz + 1e-5)
    cos_val = -1 * torch.abs(true_cos)
    prev = torch.cat([torch.zeros(n, 1), cos_val[:, :-1]], -1)
    cos_val = torch.minimum(prev, cos_val)
    cos_val = cos_val.clip(-1e3, 0.0) * inside
    vis_mask = torch.cat([torch.ones(n, 1), (true_cos < 0.05).float()], -1)
    raw_occ = udf2logistic(udf, beta, 1.0, 1.0)
    alpha_occ = 1.0 - torch.exp(-F.relu(raw_occ) * gamma * d_raw)
    vis = excl_cumprod((1.0 - alpha_occ + vis_mask).clip(0, 1) + 1e-7)
    sp = vis[:, :-1]
    a_p = sdf2alpha(mid_udf, cos_val, dists, inv_s, kind=kind)
    a_m = sdf2alpha(-mid_udf, cos_val, dists, inv_s, kind=kind)
    alpha = a_p * sp + a_m * (1 - sp)
    w = alpha * excl_cumprod(1.0 - alpha + 1e-7)
    if dbg is not None:
        dbg.update(cos_val=cos_val, vis=sp, alpha_plus=a_p, alpha_minus=a_m, alpha=alpha, weights=w)
    return sample_pdf_det(z, w, k, dbg)


def up_sample_no_occ_aware(z, udf, sample_dist, k, beta, gamma) -> Tensor:
    """udf_renderer_blending.py:834-866 (inv_s arg is unused there)."""
    n, m = z.shape
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((n, 1), float(sample_dist))], -1)
    raw = udf2logistic(udf.reshape(n, m), beta, gamma, 1.0)
    alpha_occ = 1.0 - torch.exp(-F.relu(raw) * dists)
    return sample_pdf_det(z, alpha_occ[:, :-1], k)


def merge_sorted(z: Tensor, z_new: Tensor, udf: Optional[Tensor], udf_new: Optional[Tensor]):
    """cat + sort + gather part of cat_z_vals (udf_renderer_blending.py:278-288)."""
    zc = torch.cat([z, z_new], -1)
    zs, idx = torch.sort(zc, -1)
    us = None
    if udf is not None and udf_new is not None:
        us = torch.gather(torch.cat([udf, udf_new], -1), 1, idx)
    return zs, us


@dataclass
class RenderCfg:
    n_samples: int = 64
    n_importance: int = 64
    n_outside: int = 0
    up_sample_steps: int = 4
    perturb: float = 1.0
    upsampling_type: str = "classical"
    sdf2alpha_type: str = "numerical"
    sparse_scale_factor: float = 25000.0
    h_patch_size: int = 3
    use_norm_grad_for_cosine: bool = False
    beta_min: float = 0.00005
    udf: UDFCfg = field(default_factory=UDFCfg)
    color: ColorCfg = field(default_factory=ColorCfg)
    nerf: NerfCfg = field(default_factory=NerfCfg)


@dataclass
class Nets:
    udf: SD
    color: SD
    var: SD
    beta: SD
    nerf: Optional[SD] = None


def importance_sample(nets: Nets, cfg: RenderCfg, rays_o, rays_d, z, sample_dist, trace=None):
    """classical: udf_renderer_blending.py:723-755 ; mix: :762-832."""
    n = rays_o.shape[0]
    with torch.no_grad():
        def udf_at(zz):
            p = rays_o[:, None, :] + rays_d[:, None, :] * zz[..., :, None]
            return udf_forward(nets.udf, p.reshape(-1, 3), cfg.udf)[:, 0].reshape(n, -1)

        udf = udf_at(z)
        steps = cfg.up_sample_steps
        if cfg.upsampling_type == "classical":
            k = cfg.n_importance // steps
            for i in range(steps):
                g = float(np.clip(20 * 2 ** (steps - i), 20, 320))
                z_new = up_sample_unbias(rays_o, rays_d, z, udf, sample_dist, k,
                                         64 * 2 ** i, 64 * 2 ** (i + 1), g, kind=cfg.sdf2alpha_type)
                last = (i + 1 == steps)
                if trace is not None:
                    trace.append(dict(z=z.clone(), udf=udf.clone(), z_new=z_new.clone(),
                                      inv_s=64 * 2 ** i, beta=64 * 2 ** (i + 1), gamma=g, kind="unbias"))
                z, udf = merge_sorted(z, z_new, None if last else udf, None if last else udf_at(z_new))
        else:
            k = cfg.n_importance // (steps + 1)
            gamma = gamma_of(nets.beta)
            for i in range(steps):
                z_new = up_sample_no_occ_aware(z, udf, sample_dist, k, 64 * 2 ** (i + 1), gamma)
                if trace is not None:
                    trace.append(dict(z=z.clone(), udf=udf.clone(), z_new=z_new.clone(),
                                      inv_s=64 * 2 ** i, beta=64 * 2 ** (i + 1), gamma=float(gamma), kind="noocc"))
                z, udf = merge_sorted(z, z_new, udf, udf_at(z_new))
            i = steps - 1
            z_new = up_sample_unbias(rays_o, rays_d, z, udf, sample_dist, k,
                                     64 * 2 ** i, 64 * 2 ** (i + 1), 20 if i < 4 else 10, kind=cfg.sdf2alpha_type)
            if trace is not None:
                trace.append(dict(z=z.clone(), udf=udf.clone(), z_new=z_new.clone(),
                                  inv_s=64 * 2 ** i, beta=64 * 2 ** (i + 1), gamma=20 if i < 4 else 10,
                                  kind="unbias"))
            z, _ = merge_sorted(z, z_new, None, None)
    return z


def render_core_outside(nets: Nets, cfg: RenderCfg, rays_o, rays_d, z, sample_dist):
    """udf_renderer_blending.py:161-195 (only sampled_color / alpha are consumed downstream)."""
    n, s = z.shape
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((n, 1), float(sample_dist))], -1)
    mid = z + dists * 0.5
    pts = rays_o[:, None, :] + rays_d[:, None, :] * mid[..., :, None]
    r = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).clip(1.0, 1e10)
    pts4 = torch.cat([pts / r, 1.0 / r], -1)
    dirs = rays_d[:, None, :].expand(n, s, 3)
    raw, rgb = nerf_forward(nets.nerf, pts4.reshape(-1, 4), dirs.reshape(-1, 3), cfg.nerf)
    alpha = 1.0 - torch.exp(-F.relu(raw.reshape(n, s)) * dists)
    return alpha, rgb.reshape(n, s, 3)


def render_core(nets: Nets, cfg: RenderCfg, rays_o, rays_d, z, sample_dist,
                cos_anneal_ratio=None, background_rgb=None, bg_alpha=None, bg_color=None,
                flip_saturation=0.0, blend=None):
    """udf_renderer_blending.py:327-584.  `blend` = dict(color_maps, w2cs, intrinsics,
    query_c2w, rays_uv) or None."""
    n, s = z.shape
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((n, 1), float(sample_dist))], -1)
    mid = z + dists * 0.5
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid[..., :, None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(n, s, 3).reshape(-1, 3)

    out = udf_forward(nets.udf, pts, cfg.udf)
    udf = out[:, :1]
    feat = out[:, 1:]
    grad = udf_gradient(nets.udf, pts, cfg.udf, create_graph=torch.is_grad_enabled())
    gmag = torch.linalg.norm(grad, ord=2, dim=-1, keepdim=True)
    gnorm = grad / (gmag + 1e-5)

    inv_s = inv_s_of(nets.var)
    beta = beta_of(nets.beta, cfg.beta_min)
    gamma = gamma_of(nets.beta)

    true_cos = (dirs * (gnorm if cfg.use_norm_grad_for_cosine else grad)).sum(-1, keepdim=True)
    with torch.no_grad():
        c = (dirs * gnorm).sum(-1, keepdim=True)
        flip = torch.sign(c) * -1
        flip[flip == 0] = 1

    raw_occ = udf2logistic(udf, beta, 1.0, 1.0).reshape(n, s)
    alpha_occ = 1.0 - torch.exp(-F.relu(raw_occ) * gamma * dists)
    vis_mask = (true_cos < 0.01).float().reshape(n, s)
    vis_mask = torch.cat([vis_mask[:, 1:], torch.ones(n, 1)], -1)
    vis = excl_cumprod((1.0 - alpha_occ + flip_saturation * vis_mask).clip(0, 1) + 1e-7).clip(0, 1)

    a_p = sdf2alpha(udf, -1 * torch.abs(true_cos), dists.reshape(-1, 1), inv_s, cos_anneal_ratio,
                    kind=cfg.sdf2alpha_type).reshape(n, s)
    a_m = sdf2alpha(-udf, -1 * torch.abs(true_cos), dists.reshape(-1, 1), inv_s, cos_anneal_ratio,
                    kind=cfg.sdf2alpha_type).reshape(n, s)
    alpha = a_p * vis + a_m * (1 - vis)
    udf = udf.reshape(n, s)

    cb, col, logits = color_forward(nets.color, pts, gnorm, dirs, feat, cfg.color)
    cb = cb.reshape(n, s, 3)
    col = col.reshape(n, s, 3)
    logits = logits.reshape(n, s, -1)

    pix = None
    patch = None
    patch_mask = None
    if blend is not None and blend.get("color_maps") is not None:
        pts3 = pts.reshape(n, s, 3)
        pcol, pmask = pixel_warp(pts3, blend["color_maps"], blend["intrinsics"], blend["w2cs"])
        tcol, tmask = None, None
        if blend.get("rays_uv") is not None:
            tcol, tmask = patch_warp(pts3, blend["rays_uv"], (flip.reshape(n, s, 1) * gnorm.reshape(n, s, 3)).detach(),
                                     blend["color_maps"], blend["intrinsics"][0], blend["intrinsics"],
                                     blend["query_c2w"], torch.inverse(blend["w2cs"]), cfg.h_patch_size)
        pix, _, patch, patch_mask = color_blend(logits, pcol, pmask, tcol, tmask)

    pn = torch.linalg.norm(pts, ord=2, dim=-1).reshape(n, s)
    inside = (pn < 1.0).float()
    relax = (pn < 1.2).float()
    near = (udf < 0.05).float().detach()

    if bg_alpha is not None:                                         # :490-506
        alpha = torch.cat([alpha, bg_alpha[:, s:]], -1)
        cb = torch.cat([cb, bg_color[:, s:]], 1)
        col = torch.cat([col, bg_color[:, s:]], 1)
        if pix is not None:
            pix = pix * inside[:, :, None] + bg_color[:, :s] * (1.0 - inside)[:, :, None]
            pix = torch.cat([pix, bg_color[:, s:]], 1)

    weights = alpha * excl_cumprod(1.0 - alpha + 1e-7)
    wsum = weights.sum(-1, keepdim=True)
    color_base = (cb * weights[:, :, None]).sum(1)
    color = (col * weights[:, :, None]).sum(1)
    color_pixel = (pix * weights[:, :, None]).sum(1) if pix is not None else None
    patch_colors = None
    fused_patch_mask = None
    if patch is not None:
        patch_colors = (patch * weights[:, :s, None, None]).sum(1)
        fused_patch_mask = (patch_mask.float().reshape(n, s) * weights[:, :s]).sum(1)
    depth = (mid * weights[:, :s]).sum(1, keepdim=True)
    if background_rgb is not None:
        color = color + background_rgb * (1.0 - wsum)

    g3 = grad.reshape(n, s, 3)
    ge = (torch.linalg.norm(g3, ord=2, dim=-1) - 1.0) ** 2
    gradient_error = (relax * ge).sum() / (relax.sum() + 1e-5)
    gradient_error_ns = (near * ge).sum() / (near.sum() + 1e-5)
    gflip = flip.reshape(n, s, 1) * g3
    sparse_error = torch.exp(-cfg.sparse_scale_factor * udf).sum(1).mean()

    return dict(color_base=color_base, color=color, color_pixel=color_pixel, patch_colors=patch_colors,
                patch_mask=fused_patch_mask, weights=weights, s_val=1.0 / inv_s, beta=1.0 / beta, gamma=gamma,
                depth=depth, gradient_error=gradient_error, gradient_error_near_surface=gradient_error_ns,
                normals=(gflip * weights[:, :s, None]).sum(1), gradients=g3, gradients_flip=gflip,
                inside_sphere=inside, udf=udf, gradient_mag=gmag.reshape(n, s), true_cos=true_cos.reshape(n, s),
                vis_prob=vis, alpha=alpha[:, :s], alpha_plus=a_p, alpha_minus=a_m, mid_z_vals=mid, dists=dists,
                sparse_error=sparse_error, alpha_occ=alpha_occ, raw_occ=raw_occ,
                # extras for stage-wise kernel tests (not in the reference dict)
                _feat=feat, _pts=pts, _sampled_color=col, _sampled_color_base=cb, _logits=logits)


def coarse_z(cfg: RenderCfg, near: Tensor, far: Tensor, n: int, t_rand=None, t_rand_out=None):
    """udf_renderer_blending.py:605-630."""
    sample_dist = ((far - near) / cfg.n_samples).mean().item()
    z = near + (far - near) * torch.linspace(0.0, 1.0, cfg.n_samples)[None, :]
    if z.shape[0] == 1 and n > 1 and t_rand is None:
        pass
    z_out = None
    if cfg.n_outside > 0:
        z_out = torch.linspace(1e-3, 1.0 - 1.0 / (cfg.n_outside + 1.0), cfg.n_outside)
    if t_rand is not None:
        z = z + t_rand * 2.0 / cfg.n_samples
        if cfg.n_outside > 0 and t_rand_out is not None:
            mids = .5 * (z_out[1:] + z_out[:-1])
            upper = torch.cat([mids, z_out[-1:]], -1)
            lower = torch.cat([z_out[:1], mids], -1)
            z_out = lower + (upper - lower) * t_rand_out
    if cfg.n_outside > 0:
        z_out = far / torch.flip(z_out, dims=[-1]) + 1.0 / cfg.n_samples
    return z, z_out, sample_dist


def render(nets: Nets, cfg: RenderCfg, rays_o, rays_d, near, far, cos_anneal_ratio=None,
           background_rgb=None, flip_saturation=0.0, blend=None, t_rand=None, t_rand_out=None,
           trace=None):
    """UDFRendererBlending.render (udf_renderer_blending.py:586-721).  Randomness is an
    input (`t_rand` [N,1] in [-0.5,0.5), `t_rand_out` [n_outside] in [0,1)); None = no
    perturbation (= perturb_overwrite=0).  `sparse_random_error` (:681-686, unused by
    the runner) is not produced."""
    n = rays_o.shape[0]
    if not isinstance(near, torch.Tensor):
        near = torch.tensor([near], dtype=torch.float32).view(1, 1)
        far = torch.tensor([far], dtype=torch.float32).view(1, 1)
    z, z_out, sample_dist = coarse_z(cfg, near, far, n, t_rand, t_rand_out)
    if z.shape[0] != n:
        z = z.expand(n, -1)
    s = cfg.n_samples
    if cfg.n_importance > 0:
        z = importance_sample(nets, cfg, rays_o, rays_d, z, sample_dist, trace)
        s = cfg.n_samples + cfg.n_importance
    bg_alpha = bg_color = None
    if cfg.n_outside > 0:
        zf, _ = torch.sort(torch.cat([z, z_out.expand(n, -1) if z_out.dim() == 1 or z_out.shape[0] != n else z_out], -1), -1)
        bg_alpha, bg_color = render_core_outside(nets, cfg, rays_o, rays_d, zf, sample_dist)
    ret = render_core(nets, cfg, rays_o, rays_d, z, sample_dist, cos_anneal_ratio, background_rgb,
                      bg_alpha, bg_color, flip_saturation, blend)
    w = ret["weights"]
    ret["weight_sum"] = w[:, :s].sum(-1, keepdim=True)
    ret["weight_sum_fg_bg"] = w.sum(-1, keepdim=True)
    ret["variance"] = ret["s_val"]
    ret["z_vals"] = z
    ret["_sample_dist"] = sample_dist
    return ret


# --------------------------------------------------------------------------- #
# blending: projection, bilinear taps, view softmax
# --------------------------------------------------------------------------- #
def bilinear_zeros(img: Tensor, x: Tensor, y: Tensor) -> Tensor:
    """F.grid_sample(bilinear, padding zeros, align_corners=True) on PIXEL coordinates.
    img [C,H,W]; x,y [...] float pixel coords -> [..., C].  Restates the ATen
    grid_sampler_2d arithmetic used at projector_utils.py:78 and patch_projector.py:143."""
    c, h, w = img.shape
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    x1 = x0 + 1
    y1 = y0 + 1
    wx1 = x - x0
    wx0 = x1 - x
    wy1 = y - y0
    wy0 = y1 - y
    flat = img.reshape(c, -1)

    def tap(xi, yi):
        ok = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
        idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long()
        v = flat[:, idx.reshape(-1)].reshape(c, *idx.shape)
        return v * ok.to(v.dtype)

    out = (tap(x0, y0) * (wx0 * wy0) + tap(x1, y0) * (wx1 * wy0)
           + tap(x0, y1) * (wx0 * wy1) + tap(x1, y1) * (wx1 * wy1))
    return out.movedim(0, -1)


def pixel_warp(pts: Tensor, imgs: Tensor, intrinsics: Tensor, w2cs: Tensor):
    """PatchProjector.pixel_warp -> sample_ptsFeatures_from_featureMaps -> cam2pixel
    (patch_projector.py:21-43, projector_utils.py:52-85, 8-48).
    pts [N,S,3], imgs [V,3,H,W] -> colours [N,S,V,3], mask [N,S,V]."""
    v, _, h, w = imgs.shape
    proj = torch.matmul(intrinsics[:, :3, :3], w2cs[:, :3, :])       # [V,3,4]
    p = pts.reshape(-1, 3)
    pc = torch.einsum("vij,pj->vpi", proj[:, :, :3], p) + proj[:, None, :, 3]
    zc = pc[..., 2].clamp(min=1e-3)
    xn = 2 * (pc[..., 0] / zc) / (w - 1) - 1
    yn = 2 * (pc[..., 1] / zc) / (h - 1) - 1
    xn = torch.where((xn > 1) | (xn < -1), torch.full_like(xn, 2.0), xn)
    yn = torch.where((yn > 1) | (yn < -1), torch.full_like(yn, 2.0), yn)
    mask = (xn.abs() < 1.0) & (yn.abs() < 1.0)
    xp = (xn + 1) / 2 * (w - 1)                                       # align_corners=True unnormalise
    yp = (yn + 1) / 2 * (h - 1)
    cols = torch.stack([bilinear_zeros(imgs[i], xp[i], yp[i]) for i in range(v)], 0)   # [V,P,3]
    n, s = pts.shape[:2]
    return cols.permute(1, 0, 2).reshape(n, s, v, 3), mask.t().reshape(n, s, v)


def patch_offsets(h: int) -> Tensor:
    """build_patch_offset (patch_projector.py:211-214): (dx,dy), dy-major (row by row)."""
    o = torch.arange(-h, h + 1)
    yy, xx = torch.meshgrid(o, o, indexing="ij")
    return torch.stack([xx, yy], -1).reshape(1, -1, 2).float()


def patch_warp(pts, uv_ndc, normals, src_imgs, ref_intr, src_intrs, ref_c2w, src_c2ws, hps):
    """PatchProjector.patch_warp + patch_homography (patch_projector.py:45-164).
    uv_ndc [N,2] in (-1,1) is NOT mutated here (the reference rescales it in place, :75-76).
    -> colours [N,S,V,Npx,3], mask [N,S,V,Npx]."""
    n, s, _ = pts.shape
    p_tot = n * s
    v, _, H, W = src_imgs.shape
    uv = torch.stack([(uv_ndc[:, 0] + 1) / 2. * (W - 1), (uv_ndc[:, 1] + 1) / 2. * (H - 1)], -1)
    K_ref_inv = torch.inverse(ref_intr[:3, :3])
    K_src = src_intrs[:, :3, :3]
    inv_ref_pose = torch.inverse(ref_c2w)
    inv_src = torch.inverse(src_c2ws)
    cam = ref_c2w[:3, 3].unsqueeze(0)
    sdist = torch.norm(pts - cam, dim=-1).reshape(-1)
    rel = inv_src @ ref_c2w
    R_rel = rel[:, :3, :3]
    t_rel = rel[:, :3, 3:]
    R_ref = inv_ref_pose[:3, :3]
    t_ref = inv_ref_pose[:3, 3:]
    P = pts.reshape(-1, 3)
    Nn = normals.reshape(-1, 3)
    with torch.no_grad():
        rn = (R_ref @ Nn.unsqueeze(-1))                               # [P,3,1]
        pr = R_ref @ P.unsqueeze(-1) + t_ref
        d1 = torch.sum(rn * pr, dim=1).unsqueeze(1)                   # [P,1,1]
        d2 = torch.sum(rn.unsqueeze(1) * (-R_rel.transpose(1, 2) @ t_rel).unsqueeze(0), dim=2)  # [P,V,1]
        valid = (d1.abs() > 1e-3) & ((d1 - d2).abs() > 1e-3) & ((d2 / d1) < 1)
        d1s = d1.squeeze()
        sg = torch.sign(d1s)
        sg[sg == 0] = 1
        d = torch.clamp(d1s.abs(), 1e-8) * sg
        Hm = K_src.unsqueeze(1) @ (R_rel.unsqueeze(1) + t_rel.unsqueeze(1) @ rn.view(1, p_tot, 1, 3)
                                   / d.view(1, p_tot, 1, 1)) @ K_ref_inv.view(1, 1, 3, 3)
        zax = torch.tensor([0., 0., 1.]).view(1, 1, 1, 3).expand(-1, p_tot, -1, -1)
        Hi = K_src.unsqueeze(1) @ (R_rel.unsqueeze(1) + t_rel.unsqueeze(1) @ zax
                                   / sdist.view(1, p_tot, 1, 1)) @ K_ref_inv.view(1, 1, 3, 3)
        bad = ~valid.view(-1, v).t()
        Hm[bad] = Hi[bad]
    px = uv.view(n, 1, 2) + patch_offsets(hps)
    npx = px.shape[1]
    hom = torch.cat([px, torch.ones(n, npx, 1)], -1)
    tmp = torch.einsum("vprik,pok->vproi", Hm.view(v, n, s, 3, 3), hom).reshape(v, -1, 3)
    grid = tmp[..., :2] / torch.clamp(tmp[..., 2:], 1e-8)
    m = tmp[..., 2] > 0
    m = m & (grid[..., 0] < (W - hps)) & (grid[..., 1] < (H - hps)) & (grid >= hps).all(dim=-1)
    m = m.view(v, n, s, npx)
    gx = torch.clamp(2 * grid[..., 0] / (W - 1) - 1, -10, 10)
    gy = torch.clamp(2 * grid[..., 1] / (H - 1) - 1, -10, 10)
    xp = (gx + 1) / 2 * (W - 1)
    yp = (gy + 1) / 2 * (H - 1)
    cols = torch.stack([bilinear_zeros(src_imgs[i], xp[i], yp[i]) for i in range(v)], 0)
    cols = cols.view(v, n, s, npx, 3).permute(1, 2, 0, 3, 4).contiguous()
    return cols, m.permute(1, 2, 0, 3).contiguous()


def color_blend(logits, pix_col, pix_mask, patch_col=None, patch_mask=None):
    """fields.py:498-537 with img_index=None."""
    v = pix_col.shape[-2]
    x = logits[:, :, :v]
    w = torch.softmax(x, -1) * pix_mask
    w = w / (w.float().sum(-1, keepdim=True) + 1e-8)
    pix = (pix_col * w[:, :, :, None]).sum(-2)
    pmask = pix_mask.float().sum(-1, keepdim=True) > 0
    pcol = None
    pm = None
    if patch_col is not None:
        npx = patch_col.shape[3]
        vm = patch_mask.sum(-1) > npx - 1
        w2 = torch.softmax(x, -1) * vm
        w2 = w2 / (w2.float().sum(-1, keepdim=True) + 1e-8)
        pcol = (patch_col * w2[:, :, :, None, None]).sum(-3)
        pm = vm.sum(-1, keepdim=True) > 0
    return pix, pmask, pcol, pm


# --------------------------------------------------------------------------- #
# losses                                     loss/loss.py, loss/patch_metric.py
# --------------------------------------------------------------------------- #
def pixel_l1(pred, gt, mask):
    """ColorPixelLoss (loss/loss.py:21-44): the mask only enters the denominator."""
    e = (pred - gt).abs()
    if mask is not None:
        return e.sum() / (mask.sum() + 1e-4)
    return e.mean()


def ssim_window(hps: int, std: float = 1.5) -> Tensor:
    """create_window (loss/patch_metric.py:9-18): outer product of a normalised 1-D Gaussian."""
    ws = 2 * hps + 1
    g = torch.tensor([math.exp(-(x - ws // 2) ** 2 / float(2 * std ** 2)) for x in range(ws)])
    g = g / g.sum()
    return (g[:, None] @ g[None, :]).reshape(-1)


def ssim_patch_error(pred, gt, hps: int) -> Tensor:
    """SSIM.forward/_ssim with a full-patch window (loss/patch_metric.py:21-41, 76-84).
    pred, gt [N,Npx,3] -> [N]."""
    w = ssim_window(hps)[None, :, None]
    mu1 = (pred * w).sum(1)
    mu2 = (gt * w).sum(1)
    s1 = (pred * pred * w).sum(1) - mu1 ** 2
    s2 = (gt * gt * w).sum(1) - mu2 ** 2
    s12 = (pred * gt * w).sum(1) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    val = 1 - ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))
    return val.sum(-1) / 2


def ncc_patch(pred, gt, hps: int) -> Tensor:
    """NCC.forward/_ncc with a full-patch window (loss/patch_metric.py:44-67, 86-107).  pred, gt [N,Npx,3] -> [N]."""
    w = ssim_window(hps)[None, :, None]
    mu1 = (pred * w).sum(1)
    mu2 = (gt * w).sum(1)
    sigma1 = torch.sqrt((pred * pred * w).sum(1) - mu1 ** 2 + 1e-4)
    sigma2 = torch.sqrt((gt * gt * w).sum(1) - mu2 ** 2 + 1e-4)
    pn = (pred - mu1[:, None]) / (sigma1[:, None] + 1e-8)
    gn = (gt - mu2[:, None]) / (sigma2[:, None] + 1e-8)
    return (pn * gn * w).sum(1).mean(-1)


def patch_error(pred, gt, hps: int, kind: str = "ssim") -> Tensor:
    """the four per-ray errors of ColorPatchLoss.forward (loss/loss.py:66-73)."""
    if kind == "l1":
        return (pred - gt).abs().mean(-1).sum(-1)
    if kind == "ssd":
        return ((pred - gt) ** 2).mean(-1).sum(-1)
    if kind == "ncc":
        return 1 - ncc_patch(pred, gt, hps)
    return ssim_patch_error(pred, gt, hps)


def patch_loss(pred, gt, mask, hps: int, penalize_ratio: float = 0.3, kind: str = "ssim"):
    """ColorPatchLoss (loss/loss.py:56-84)."""
    err = patch_error(pred, gt, hps, kind) * mask[:, 0].float()
    err, idx = torch.sort(err, descending=True)
    m = mask[idx].clone()
    m[:int(penalize_ratio * m.sum())] = False
    return err[m.squeeze(-1)].mean()


def color_loss(w_base, w_color, w_pixel, w_patch, hps, color_base, color, gt, color_pixel, pixel_mask,
               patch_colors, gt_patch, patch_mask):
    """ColorLoss.forward (loss/loss.py:105-133)."""
    lb = pixel_l1(color_base, gt, pixel_mask) if color_base is not None else 0.0
    lc = pixel_l1(color, gt, pixel_mask) if color is not None else 0.0
    lp = pixel_l1(color_pixel, gt, patch_mask) if color_pixel is not None else 0.0
    lt = patch_loss(patch_colors, gt_patch, patch_mask, hps) if patch_colors is not None else 0.0
    total = (lb * w_base + lc * w_color + lp * w_pixel) / (w_base + w_color + w_pixel) + lt * w_patch
    return dict(loss=total, color_base_loss=lb, color_loss=lc, color_pixel_loss=lp, color_patch_loss=lt)
