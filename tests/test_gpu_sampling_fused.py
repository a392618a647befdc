"""The launches that start and end the hierarchical sampling, fused (the small-kernel tail):

  nudf_coarse_start   = torch `- 0.5` + nudf_coarse_z (coarse_z + sample_dist kernels) + nudf_ray_points(mode 0)
  nudf_upsample with merge_K > 0 = nudf_merge of the previous round + nudf_upsample
  nudf_merge_points   = nudf_merge (z only) + nudf_ray_points(mode 1) + nudf_copy_cols + the pad-column fill

Every fused launch is held BIT FOR BIT to the separate launches it replaces (same expressions, data movement otherwise),
and the renderer's sampling schedule built on them to the one built on the separate calls."""
import pytest
import torch

from neuraludf_amd._lib import Upsample, call, ptr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rays(N, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    o = (torch.randn(N, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.0, -2.5])).to(dev)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1).to(dev)
    near = (1.5 + 0.2 * torch.rand(N, 1, generator=g)).to(dev)
    far = (3.5 + 0.2 * torch.rand(N, 1, generator=g)).to(dev)
    return o.contiguous(), d.contiguous(), near.contiguous(), far.contiguous()


@pytest.mark.parametrize("N,S,jitter,per_ray", [(512, 64, True, True), (37, 64, True, True), (5, 33, False, True),
                                                (300, 16, True, False)])
def test_coarse_start_equals_the_separate_launches(dev, N, S, jitter, per_ray):
    o, d, near, far = _rays(N, dev, seed=N)
    if not per_ray:
        near, far = near[:1].contiguous(), far[:1].contiguous()
    stride = 1 if per_ray else 0
    raw = torch.rand(N, 1, device=dev) if jitter else None
    z_a = torch.empty(N, S, device=dev)
    sd_a = torch.empty(1, device=dev)
    centred = (raw - 0.5).contiguous() if jitter else None
    call("nudf_coarse_z", ptr(near), ptr(far), stride, ptr(centred), N, S, ptr(z_a), ptr(sd_a))
    p_a = torch.empty(N * S, 3, device=dev)
    call("nudf_ray_points", ptr(o), ptr(d), ptr(z_a), ptr(sd_a), N, S, 0, ptr(p_a))
    z_b = torch.empty(N, S, device=dev)
    sd_b = torch.empty(1, device=dev)
    p_b = torch.empty(N * S, 3, device=dev)
    call("nudf_coarse_start", ptr(near), ptr(far), stride, ptr(raw), 1, N, S, ptr(z_b), ptr(sd_b), ptr(o), ptr(d), ptr(p_b))
    assert torch.equal(z_a, z_b) and torch.equal(sd_a, sd_b) and torch.equal(p_a, p_b)
    # without the points / without the spacing
    z_c = torch.empty(N, S, device=dev)
    call("nudf_coarse_start", ptr(near), ptr(far), stride, ptr(centred), 0, N, S, ptr(z_c), None, None, None, None)
    assert torch.equal(z_a, z_c)


def _sorted(N, M, dev, lo=1.5, hi=3.5, seed=0, ties=False):
    g = torch.Generator().manual_seed(seed)
    z = torch.sort(lo + (hi - lo) * torch.rand(N, M, generator=g), dim=1)[0]
    if ties:
        z = (z * 16).round() / 16          # many equal values within and across the two lists
    return z.to(dev).contiguous()


@pytest.mark.parametrize("N,M0,K0,K,mode", [(512, 64, 16, 16, 0), (53, 64, 13, 13, 1), (7, 112, 16, 16, 0), (130, 300, 40, 24, 1),
                                            (64, 1, 1, 8, 0)])
def test_upsample_with_the_previous_merge_folded_in(dev, N, M0, K0, K, mode):
    o, d, _, _ = _rays(N, dev, seed=3)
    zp, za = _sorted(N, M0, dev, seed=1, ties=(N == 7)), _sorted(N, K0, dev, seed=2, ties=(N == 7))
    up = torch.rand(N, M0, device=dev) * 0.3
    ua = torch.rand(N, K0, device=dev) * 0.3
    M = M0 + K0
    zo = torch.empty(N, M, device=dev)
    uo = torch.empty(N, M, device=dev)
    call("nudf_merge", ptr(zp), ptr(up), ptr(za), ptr(ua), N, M0, K0, ptr(zo), ptr(uo))
    u = torch.linspace(0.5 / K, 1 - 0.5 / K, K, device=dev).contiguous()
    sd = torch.tensor([0.03], device=dev)
    gam = torch.tensor([20.0], device=dev)

    def args():
        a = Upsample()
        a.rays_o, a.rays_d, a.u, a.sample_dist, a.gamma_dev = ptr(o), ptr(d), ptr(u), ptr(sd), ptr(gam if mode == 1 else None)
        a.N, a.M, a.K, a.mode = N, M, K, mode | 512 | 1024 | 2048 | 4096
        a.inv_s, a.beta, a.gamma = 128.0, 256.0, 40.0
        return a
    a = args()
    a.z, a.udf = ptr(zo), ptr(uo)
    zn_a = torch.empty(N, K, device=dev)
    pn_a = torch.empty(N * K, 3, device=dev)
    a.z_new, a.pts_new = ptr(zn_a), ptr(pn_a)
    call("nudf_upsample", a)
    b = args()
    zm = torch.full((N, M), -1.0, device=dev)
    um = torch.full((N, M), -1.0, device=dev)
    b.prev_z, b.prev_udf, b.add_z, b.add_udf, b.z_merged, b.udf_merged, b.merge_K = (ptr(zp), ptr(up), ptr(za), ptr(ua),
                                                                                       ptr(zm), ptr(um), K0)
    zn_b = torch.empty(N, K, device=dev)
    pn_b = torch.empty(N * K, 3, device=dev)
    b.z_new, b.pts_new = ptr(zn_b), ptr(pn_b)
    call("nudf_upsample", b)
    assert torch.equal(zm, zo) and torch.equal(um, uo)
    assert torch.equal(zn_a, zn_b) and torch.equal(pn_a, pn_b)
    # the fast-arithmetic build (flags 0) shares the prologue
    a2, b2 = args(), args()
    a2.mode = b2.mode = mode
    a2.z, a2.udf, a2.z_new, a2.pts_new = ptr(zo), ptr(uo), ptr(zn_a), ptr(pn_a)
    b2.prev_z, b2.prev_udf, b2.add_z, b2.add_udf, b2.z_merged, b2.udf_merged, b2.merge_K = (ptr(zp), ptr(up), ptr(za), ptr(ua),
                                                                                           ptr(zm), ptr(um), K0)
    b2.z_new, b2.pts_new = ptr(zn_b), ptr(pn_b)
    call("nudf_upsample", a2)
    call("nudf_upsample", b2)
    assert torch.equal(zn_a, zn_b) and torch.equal(pn_a, pn_b)


def test_upsample_merge_arguments_are_checked(dev):
    from neuraludf_amd import _lib
    a = Upsample()
    z = torch.zeros(4, 8, device=dev)
    a.rays_o = a.rays_d = a.z = a.udf = a.u = a.sample_dist = ptr(z)
    a.N, a.M, a.K, a.mode, a.merge_K = 4, 8, 4, 0, 3          # merge requested, arrays missing
    a.z_new = ptr(z)
    with pytest.raises(RuntimeError):
        call("nudf_upsample", a)
    a.merge_K = 8                                            # nothing left of the previous list
    with pytest.raises(RuntimeError):
        call("nudf_upsample", a)


@pytest.mark.parametrize("N,M,K,ld,F", [(512, 112, 16, 288, 256), (33, 120, 8, 288, 256), (9, 64, 64, 40, 32), (3, 5, 0, 8, 4)])
def test_merge_points_equals_merge_ray_points_copy_and_fill(dev, N, M, K, ld, F):
    o, d, _, _ = _rays(N, dev, seed=5)
    za = _sorted(N, M, dev, seed=7, ties=(N == 9))
    zb = _sorted(N, max(K, 1), dev, seed=8, ties=(N == 9))[:, :K].contiguous()
    S = M + K
    sd = torch.tensor([0.0171], device=dev)
    zo = torch.empty(N, S, device=dev)
    if K:
        call("nudf_merge", ptr(za), None, ptr(zb), None, N, M, K, ptr(zo), None)
    else:
        zo.copy_(za)
    pts = torch.empty(N * S, 3, device=dev)
    call("nudf_ray_points", ptr(o), ptr(d), ptr(zo), ptr(sd), N, S, 1, ptr(pts))
    rows = (N * S + 63) // 64 * 64
    feat_a = torch.full((rows, ld), 7.0, device=dev)
    call("nudf_copy_cols", ptr(pts), 3, 1, ptr(feat_a) + 4 * F, ld, 3, N * S, 1.0)
    feat_a[:N * S, F + 3:] = 0
    zo_b = torch.empty(N, S, device=dev)
    pts_b = torch.empty(N * S, 3, device=dev)
    feat_b = torch.full((rows, ld), 7.0, device=dev)
    call("nudf_merge_points", ptr(za), ptr(zb), N, M, K, ptr(zo_b), ptr(o), ptr(d), ptr(sd), ptr(pts_b), ptr(feat_b) + 4 * F,
         ld, ld - F)
    assert torch.equal(zo, zo_b) and torch.equal(pts, pts_b) and torch.equal(feat_a, feat_b)
    zo_c = torch.empty(N, S, device=dev)
    pts_c = torch.empty(N * S, 3, device=dev)
    call("nudf_merge_points", ptr(za), ptr(zb), N, M, K, ptr(zo_c), ptr(o), ptr(d), ptr(sd), ptr(pts_c), None, 0, 0)
    assert torch.equal(zo, zo_c) and torch.equal(pts, pts_c)


@pytest.mark.parametrize("sched", ["classical", "mix"])
def test_renderer_schedule_on_the_fused_launches_equals_the_separate_ones(dev, sched):
    """UDFRendererBlending.render with the fused sampling launches against the same renderer driven through the separate
    entry points (the round-3 sequence, restated here): sample positions, colours and the UDF gradient bit for bit."""
    from common import build_modules, perturb_
    from neuraludf_amd.models import fields
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    mods = perturb_(build_modules(fields, seed=0))
    for m in mods.values():
        m.to(dev)
    N = 96
    o, d, near, far = _rays(N, dev, seed=11)
    kw = dict(n_samples=64, n_importance=64 if sched == "classical" else 60, n_outside=0, up_sample_steps=4, perturb=1.0,
              upsampling_type=sched)
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], **kw)
    torch.manual_seed(5)
    with torch.no_grad():
        out = rend.render(o, d, near, far, cos_anneal_ratio=0.7)
    # the separate launches
    torch.manual_seed(5)
    raw = torch.rand([N, 1], device=dev)
    z = torch.empty(N, 64, device=dev)
    sd = torch.empty(1, device=dev)
    call("nudf_coarse_z", ptr(near), ptr(far), 1, ptr((raw - 0.5).contiguous()), N, 64, ptr(z), ptr(sd))
    with torch.no_grad():
        udf = rend._udf_at(o, d, z, sd)
        steps = 4
        if sched == "classical":
            k = 64 // steps
            for i in range(steps):
                import numpy as np
                gamma = float(np.clip(20 * 2 ** (steps - i), 20, 320))
                z_new, p_new = rend._upsample(o, d, z, udf, sd, k, 0, 64 * 2 ** i, 64 * 2 ** (i + 1), gamma)
                u_new = None if i + 1 == steps else rend.udf_network.udf_only(p_new).reshape(N, k)
                z, udf = rend._merge(z, udf, z_new, u_new)
        else:
            k = 60 // (steps + 1)
            gd = rend.beta_network.get_gamma().clip(1e-6, 1e6).detach().reshape(1).contiguous()
            for i in range(steps):
                z_new, p_new = rend._upsample(o, d, z, udf, sd, k, 1, 64 * 2 ** i, 64 * 2 ** (i + 1), 0.0, gd)
                z, udf = rend._merge(z, udf, z_new, rend.udf_network.udf_only(p_new).reshape(N, k))
            i = steps - 1
            z_new, _ = rend._upsample(o, d, z, udf, sd, k, 0, 64 * 2 ** i, 64 * 2 ** (i + 1), 20 if i < 4 else 10)
            z, _ = rend._merge(z, udf, z_new, None)
        ref = rend.render(o, d, near, far, cos_anneal_ratio=0.7, z_vals_override=z)
    assert torch.equal(out["z_vals"], z)
    for k_ in ("color", "color_base", "weights", "gradients", "udf", "depth"):
        assert torch.equal(out[k_], ref[k_]), k_
