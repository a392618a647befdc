"""Edge cases of the hot path on the GPU: single / ragged / empty ray batches, minimum and maximum sample counts,
one-quantile up-sampling, tile-boundary point counts of the fused MLP chains, and loud failures beyond the limits."""
import pytest
import torch

from common import build_modules, perturb_, state_dicts, oracle_nets
from oracle import udf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def nets(dev):
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0))
    sds = state_dicts(mods)
    for m in mods.values():
        m.to(dev)
    return mods, sds


def _renderer(mods, **kw):
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    return UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **kw)


def _rays(n, dev, seed=3):
    from neuraludf_amd import synth
    r = synth.make_rays(synth.make_scene("tiny"), 0, max(n, 1), seed=seed)
    return {k: v[:n].to(dev) for k, v in r.items()}


@pytest.mark.parametrize("n_rays", [1, 3, 5, 66])
def test_single_and_ragged_ray_batches_match_the_oracle(dev, nets, n_rays):
    """ray counts that are not multiples of the 4-rays-per-workgroup / 64-point tile sizes."""
    mods, sds = nets
    kw = dict(n_samples=16, n_importance=0, n_outside=0, up_sample_steps=1, perturb=0.0)
    r = _rays(n_rays, dev)
    out = _renderer(mods, **kw).render(r["rays_o"], r["rays_d"], r["near"], r["far"], cos_anneal_ratio=0.5,
                                       flip_saturation=0.5)
    cfg = O.RenderCfg(n_samples=16, n_importance=0, n_outside=0, up_sample_steps=1)
    with torch.no_grad():
        ref = O.render(oracle_nets(sds), cfg, r["rays_o"].cpu(), r["rays_d"].cpu(), r["near"].cpu(), r["far"].cpu(),
                       cos_anneal_ratio=0.5, flip_saturation=0.5)
    for k in ("color", "color_base", "weights", "udf", "depth"):
        a, b = out[k].detach().cpu().reshape(ref[k].shape), ref[k]
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max())), k
    ge, ge_ref = float(out["gradient_error"].detach()), float(ref["gradient_error"])
    assert abs(ge - ge_ref) < 1e-4 * max(1.0, ge_ref)


def test_empty_ray_batch(dev, nets):
    mods, _ = nets
    r = _rays(0, dev)
    out = _renderer(mods, n_samples=16, n_importance=8, n_outside=0, up_sample_steps=2, perturb=0.0).render(
        r["rays_o"], r["rays_d"], r["near"], r["far"], cos_anneal_ratio=1.0)
    assert out["color"].shape == (0, 3) and out["weights"].shape == (0, 24) and out["z_vals"].shape == (0, 24)


def test_minimum_and_maximum_sample_counts(dev, nets):
    from neuraludf_amd._lib import NudfError
    mods, _ = nets
    r = _rays(6, dev)
    # two samples per ray is the minimum the up-sampler accepts; one quantile per round
    out = _renderer(mods, n_samples=2, n_importance=3, n_outside=0, up_sample_steps=3, perturb=0.0).render(
        r["rays_o"], r["rays_d"], r["near"], r["far"], cos_anneal_ratio=1.0)
    z = out["z_vals"]
    assert z.shape == (6, 5) and bool((z[:, 1:] >= z[:, :-1]).all()) and bool(torch.isfinite(out["color"]).all())
    # 512 samples in total (inside + outside) is the composite kernel's maximum
    out = _renderer(mods, n_samples=480, n_importance=0, n_outside=32, up_sample_steps=1, perturb=0.0).render(
        r["rays_o"], r["rays_d"], r["near"], r["far"], cos_anneal_ratio=1.0)
    assert out["weights"].shape == (6, 512) and bool(torch.isfinite(out["color"]).all())
    assert float(out["weight_sum_fg_bg"].max()) <= 1.0 + 1e-4
    with pytest.raises(NudfError):     # beyond the limit: a loud error, not a silent truncation
        _renderer(mods, n_samples=513, n_importance=0, n_outside=0, up_sample_steps=1, perturb=0.0).render(
            r["rays_o"], r["rays_d"], r["near"], r["far"], cos_anneal_ratio=1.0)


@pytest.mark.parametrize("P", [1, 31, 32, 33, 63, 64, 65, 127, 129])
def test_udf_chains_at_tile_boundaries(dev, nets, P):
    """point counts around the 32- / 64-point tiles of the fused chains: value, gradient and parameter gradients."""
    mods, sds = nets
    g = torch.Generator().manual_seed(P)
    x = torch.randn(P, 3, generator=g) * 0.7
    wy, wg = torch.randn(P, 257, generator=g), torch.randn(P, 3, generator=g)
    on = oracle_nets(sds, requires_grad=True)
    y_ref = O.udf_forward(on.udf, x)
    g_ref = O.udf_gradient(on.udf, x, create_graph=True)
    ((y_ref * wy).sum() + (g_ref * wg).sum()).backward()
    net = mods["udf"]
    net.zero_grad()
    udf, feat, grad = net.evaluate(x.to(dev), want_grad=True)
    tol = lambda b: 1e-4 * max(1.0, float(b.abs().max()))
    assert float((udf.cpu() - y_ref[:, 0]).abs().max()) <= tol(y_ref[:, 0])
    assert float((feat[:, :256].cpu() - y_ref[:, 1:]).abs().max()) <= tol(y_ref[:, 1:])
    assert float((grad.cpu() - g_ref).abs().max()) <= tol(g_ref)
    ((udf * wy[:, 0].to(dev)).sum() + (feat[:, :256] * wy[:, 1:].to(dev)).sum() + (grad * wg.to(dev)).sum()).backward()
    for n, p in net.named_parameters():
        ref = on.udf[n].grad
        assert float((p.grad.cpu() - ref).abs().max()) <= 1e-3 * max(1e-6, float(ref.abs().max())), n


def test_kernel_argument_errors_are_loud(dev):
    from neuraludf_amd._lib import NudfError, Upsample, call, ptr
    a = Upsample()
    z = torch.zeros(4, 1, device=dev)
    a.rays_o = a.rays_d = ptr(torch.zeros(4, 3, device=dev))
    a.z = a.udf = ptr(z)
    a.N, a.M, a.K, a.mode = 4, 1, 2, 0           # M = 1: no section to sample from
    with pytest.raises(NudfError):
        call("nudf_upsample", a)
    with pytest.raises(NudfError):
        ptr(torch.zeros(3))                       # CPU tensor


@pytest.mark.parametrize("S", [128, 256, 512])
def test_composite_blocked_layout_equals_strided_layout(dev, S):
    """the FULL-case composite kernels exist in two lane layouts (sample i in lane i % 64 -- the default -- or lane l owns
    S/64 consecutive samples with 16-byte vector accesses, the bandwidth probe of DESIGN.md section 4.3): same
    arithmetic, different association of the two product scans.  Forward outputs and every gradient must agree to fp32
    rounding, with and without cosine annealing / normalised-gradient cosines."""
    from neuraludf_amd import _lib
    from neuraludf_amd.models.udf_renderer_blending import _CompositeFn
    g = torch.Generator().manual_seed(S)
    N = 67
    ro = torch.randn(N, 3, generator=g) * 0.3
    rd = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    z = torch.sort(torch.rand(N, S, generator=g) * 2 + 1.0, -1)[0]
    udf = (torch.rand(N, S, generator=g) * 0.2) ** 2
    udf[:, S // 2] = 1e-4
    grad = torch.randn(N, S, 3, generator=g) * 0.8
    col, cb = torch.rand(N, S, 3, generator=g), torch.rand(N, S, 3, generator=g)
    scal = torch.tensor([40.0, 70.0, 20.0])
    sd = torch.tensor([2.0 / 64])
    k = [torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g), torch.randn(N, S, generator=g) * 0.1,
         torch.randn(N, 1, generator=g), torch.randn(N, 3, generator=g), torch.randn(5, generator=g) * 1e-3]
    D = lambda t: t.to(dev)
    res = {}
    for anneal, use_norm in ((None, False), (0.6, True)):
        for blocked in (1, 0):
            _lib.lib().nudf_set_composite_blocked(blocked)
            try:
                leaves = [D(t).requires_grad_(True) for t in (udf, grad, col, cb, scal)]
                c = dict(s_nominal=S, cos_anneal=anneal, flip_saturation=0.9, use_norm_grad=use_norm, sparse_scale=25000.0,
                         diagnostics=False)
                out = _CompositeFn.apply(c, D(ro), D(rd), D(z), D(sd), None, leaves[0], leaves[1], leaves[2], leaves[3], None,
                                         None, None, leaves[4])
                color, cbo, w, depth, normals, wsum, wall, sums = out[:8]
                loss = ((color * D(k[0])).sum() + (cbo * D(k[1])).sum() + (w * D(k[2])).sum() + (depth * D(k[3])).sum()
                        + (normals * D(k[4])).sum() + (sums * D(k[5])).sum() + wsum.sum() * 0.3)
                loss.backward()
                res[blocked] = [t.detach().cpu() for t in (color, cbo, w, depth, normals, wsum, wall, sums)] + \
                               [t.grad.detach().cpu() for t in leaves]
            finally:
                _lib.lib().nudf_set_composite_blocked(0)
        names = ["color", "color_base", "weights", "depth", "normals", "wsum", "wsum_all", "sums", "d_udf", "d_grad",
                 "d_color", "d_color_base", "d_scal"]
        for nm, a, b in zip(names, res[1], res[0]):
            den = float(b.abs().max().clamp(min=1e-6))
            tol = 2e-4 if nm.startswith("d_") else 2e-5
            assert float((a - b).abs().max()) / den < tol, (S, anneal, nm, float((a - b).abs().max()) / den)
