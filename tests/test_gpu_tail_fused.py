"""Small launches of a train step folded into their neighbours, each against the separate launches it
replaces, to the bit:

  * NudfGemmTNGroup.assign            -- the first weight-gradient launch ASSIGNS: no zero fill of the gradient buffers
  * NudfComposite.p_variance ...      -- inv_s / beta / gamma formed inside the composite launches (no nudf_scalars_fwd),
    NudfCompositeGrad.o_d_param          the backward's reduction + nudf_scalars_bwd as one launch
  * NudfComposite.defer_sums          -- the composite's batch sums reduced by the consumer's launch (nudf_step_loss_fwd)
  * nudf_col0_seed4's second output, the resident seed of loss.backward()

and the whole step with all of them against a trainer with every switch off."""
import pytest
import torch

from neuraludf_amd import synth
from neuraludf_amd._lib import call, ptr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("M", [333, 4096])
def test_weight_gradient_gemm_assign_equals_accumulate_into_zeros(dev, mode, M):
    from neuraludf_amd import mlp
    mlp.set_precision(mode)
    g = torch.Generator().manual_seed(9)
    shapes = [(256, 256), (217, 256), (256, 40), (3, 128), (129, 72), (1, 256)]
    ops = []
    for NA, NB in shapes:
        lda, ldb = (NA + 3) // 4 * 4, (NB + 3) // 4 * 4
        ops.append((torch.randn(M, lda, generator=g).to(dev), NA, torch.randn(M, ldb, generator=g).to(dev), NB))

    def jobs(fill):
        return [(A, NA, B, NB, torch.full((mlp.pad32(NA), B.shape[1]), fill, device=dev),
                 torch.full((mlp.pad32(NA),), fill, device=dev)) for A, NA, B, NB in ops]
    ja = jobs(0.0)
    mlp.gemm_tn_grouped(ja, M)
    jb = jobs(float("nan"))
    mlp.gemm_tn_grouped(jb, M, assign=True)
    for (A, NA, B, NB), a, b in zip(ops, ja, jb):
        assert torch.equal(a[4][:NA, :NB], b[4][:NA, :NB]) and torch.equal(a[5][:NA], b[5][:NA])
        assert bool(torch.isnan(b[4][NA:]).all()) and bool(torch.isnan(b[5][NA:]).all())      # untouched
    # a second launch into the same buffers accumulates as before
    mlp.gemm_tn_grouped(ja, M)
    mlp.gemm_tn_grouped(jb, M)
    for (A, NA, B, NB), a, b in zip(ops, ja, jb):
        assert torch.equal(a[4][:NA, :NB], b[4][:NA, :NB]) and torch.equal(a[5][:NA], b[5][:NA])


def test_assign_without_the_two_pass_reduction_is_refused(dev):
    from neuraludf_amd import mlp, _lib
    A, B = torch.randn(64, 32, device=dev), torch.randn(64, 32, device=dev)
    job = [(A, 32, B, 32, torch.zeros(32, 32, device=dev), None)]
    mlp.TN_DETERMINISTIC = False
    try:
        with pytest.raises(RuntimeError):
            mlp.gemm_tn_grouped(job, 64, assign=True)
    finally:
        mlp.TN_DETERMINISTIC = True


def _trainer(dev, rconf, seed=0):
    from neuraludf_amd.train import Trainer
    tr = Trainer(dev, rconf, seed=seed, fused_adam=True)
    tr.renderer.diagnostics = False
    return tr


RCONFS = {"classical": dict(n_samples=32, n_importance=32, n_outside=0, up_sample_steps=2, perturb=1.0),
          "background": dict(n_samples=32, n_importance=16, n_outside=8, up_sample_steps=2, perturb=1.0),
          "mix": dict(n_samples=32, n_importance=30, n_outside=0, up_sample_steps=2, perturb=1.0, upsampling_type="mix")}


@pytest.mark.parametrize("name", list(RCONFS))
def test_train_steps_with_and_without_the_folded_launches_are_bit_identical(dev, name):
    """three steps (render, loss, backward, Adam) with every fusion of this round on (the default) against a trainer with
    all of them switched off: loss, colours, and every parameter after every step."""
    from neuraludf_amd import mlp
    from neuraludf_amd.models import udf_renderer_blending as R
    scene = synth.make_scene("tiny")

    def run(fused):
        old = (R.FUSE_SCALARS, mlp.TN_ASSIGN)
        R.FUSE_SCALARS, mlp.TN_ASSIGN = fused, fused
        try:
            tr = _trainer(dev, RCONFS[name])
            tr.defer_sums_reduce = fused
            torch.manual_seed(77)
            hist = []
            for i in range(3):
                rays = synth.make_rays(scene, i % 3, 160, seed=50 + i)
                batch = {k: v.to(dev) for k, v in rays.items()}
                loss, out = tr.step(batch, cos_anneal_ratio=0.5 + 0.1 * i, flip_saturation=0.3 * i)
                hist.append((loss.clone(), out["color"].clone(), out["gradient_error"].clone(), out["variance"].clone(),
                             out["beta"].clone(), out["gamma"].clone(),
                             [p.detach().clone() for g in tr.param_groups for p in g]))
            torch.cuda.synchronize()
            return hist
        finally:
            R.FUSE_SCALARS, mlp.TN_ASSIGN = old
    a, b = run(True), run(False)
    for i, (x, y) in enumerate(zip(a, b)):
        for j in range(6):
            assert torch.equal(x[j], y[j]), (i, j)
        for k, (p, q) in enumerate(zip(x[6], y[6])):
            assert torch.equal(p, q), ("parameter", i, k)


def test_composite_scalars_inside_the_launch_and_deferred_sums(dev):
    """_CompositeFn with the three scalar parameters against the scal vector from nudf_scalars_fwd: every output, the
    scalar gradients through nudf_scalars_bwd; and the sums left as partials against nudf_partial_sums / the step loss."""
    from neuraludf_amd.models.udf_renderer_blending import _CompositeFn, _ScalarsFn, UDFRendererBlending
    g = torch.Generator().manual_seed(4)
    N, S = 37, 64
    D = lambda *s: torch.rand(*s, generator=g).to(dev)
    o = D(N, 3) * 0.2 - torch.tensor([0.0, 0.0, 2.5], device=dev)
    d = torch.nn.functional.normalize(D(N, 3) * 0.2 + torch.tensor([0.0, 0.0, 1.0], device=dev), dim=-1)
    z = torch.sort(1.5 + 2.0 * D(N, S), dim=1)[0].contiguous()
    sd = torch.tensor([0.03], device=dev)
    leaves = [(D(N, S) * 0.2).requires_grad_(), (D(N, S, 3) - 0.5).requires_grad_(), D(N, S, 3).requires_grad_(),
              D(N, S, 3).requires_grad_()]
    pv = torch.tensor([0.3], device=dev, requires_grad=True)
    pb = torch.tensor([0.45], device=dev, requires_grad=True)
    pg = torch.tensor([0.35], device=dev, requires_grad=True)
    c = dict(s_nominal=S, cos_anneal=0.6, flip_saturation=0.2, use_norm_grad=False, sparse_scale=25.0, diagnostics=False,
             alpha_type=0, sched=None, beta_hi=1.0 / 0.0005)

    def loss_of(outs):
        return (outs[0].sum() * 1.3 + outs[1].sum() + (outs[2] ** 2).sum() + outs[7] @ torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5],
                                                                                                      device=dev))
    scal, recip = _ScalarsFn.apply(pv, pb, pg, c["beta_hi"])
    ref = _CompositeFn.apply(c, o, d, z, sd, None, *leaves, None, None, None, scal)
    loss_of(ref).backward()
    gref = [t.grad.clone() for t in leaves + [pv, pb, pg]]
    for t in leaves + [pv, pb, pg]:
        t.grad = None
    out = _CompositeFn.apply(c, o, d, z, sd, None, *leaves, None, None, None, None, pv, pb, pg)
    assert len(out) == len(ref) + 2
    for a, b in zip(ref, out):
        assert torch.equal(a, b)
    assert torch.equal(out[-2], scal) and torch.equal(out[-1], recip)
    loss_of(out).backward()
    for a, t in zip(gref, leaves + [pv, pb, pg]):
        assert torch.equal(a, t.grad)
    # deferred second stage
    c2 = dict(c, defer_sums=True)
    with torch.no_grad():
        out2 = _CompositeFn.apply(c2, o, d, z, sd, None, *leaves, None, None, None, None, pv, pb, pg)
    sums = out2[7]
    assert hasattr(sums, "_nudf_ws")
    UDFRendererBlending.finish_sums(sums)
    assert not hasattr(sums, "_nudf_ws") and torch.equal(sums, ref[7])


def test_step_loss_reduces_the_partials_it_is_handed(dev):
    g = torch.Generator().manual_seed(6)
    n, nblk = 3 * 500, 129
    cb, c_, gt = (torch.rand(n, generator=g).to(dev) for _ in range(3))
    ws = torch.rand(nblk, 5, generator=g).to(dev).contiguous()
    sums_a = torch.empty(5, device=dev)
    call("nudf_partial_sums", ptr(ws), nblk, 5, ptr(sums_a))
    w = torch.zeros(16, device=dev)
    w[:9] = torch.tensor([0.5, 1.0, 0.0, 0.0, 0.1, 0.01, 0.02, 0.0, 1.5])
    out_a, den_a = torch.empty(8, device=dev), torch.empty(1, device=dev)
    call("nudf_step_loss_fwd", ptr(cb), ptr(c_), ptr(gt), n, None, 0, ptr(sums_a), 500.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, ptr(w),
         ptr(out_a), ptr(den_a), None, 0)
    sums_b = torch.full((5,), float("nan"), device=dev)
    out_b, den_b = torch.empty(8, device=dev), torch.empty(1, device=dev)
    call("nudf_step_loss_fwd", ptr(cb), ptr(c_), ptr(gt), n, None, 0, ptr(sums_b), 500.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, ptr(w),
         ptr(out_b), ptr(den_b), ptr(ws), nblk)
    assert torch.equal(sums_a, sums_b) and torch.equal(out_a, out_b) and torch.equal(den_a, den_b)
    assert torch.allclose(sums_a, ws.double().sum(0).float(), rtol=1e-5)


# ---------------------------------------------------------------------------------------------------------------------
# the colour network's epilogue takes the compositing sum (no-grad rendering)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,S,bg", [(64, 128, False), (37, 64, True), (5, 32, False), (130, 96, True)])
def test_no_grad_render_with_the_colour_sums_inside_the_colour_heads(dev, N, S, bg):
    """render() under no_grad: weights first, then the colour chain with row_w / row_sums and nudf_composite_colour_finish,
    against the same render with the per-sample colours through memory (NUDF_FUSE_COLOUR=0): everything that does not
    involve the colours bit for bit, the two colours to fp32 summation noise (32-point partial sums instead of the
    composite kernel's wave reduction)."""
    from common import build_modules, perturb_
    from neuraludf_amd.models import fields, udf_renderer_blending as R
    mods = perturb_(build_modules(fields, seed=0))
    for m in mods.values():
        m.to(dev)
    g = torch.Generator().manual_seed(N)
    o = (torch.randn(N, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, -2.5])).to(dev)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1).to(dev)
    near, far = torch.full((N, 1), 1.5, device=dev), torch.full((N, 1), 3.5, device=dev)
    rend = R.UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], n_samples=S // 2,
                                 n_importance=S // 2, n_outside=0, up_sample_steps=2, perturb=0.0)
    kw = dict(cos_anneal_ratio=0.8, background_rgb=(torch.tensor([1.0, 0.5, 0.25]) if bg else None))
    old = R.FUSE_COLOUR
    try:
        outs = []
        for fused in (True, False):
            R.FUSE_COLOUR = fused
            with torch.no_grad():
                outs.append(rend.render(o, d, near, far, **kw))
    finally:
        R.FUSE_COLOUR = old
    a, b = outs
    for k in ("weights", "z_vals", "depth", "normals", "weight_sum", "gradients", "udf", "gradient_error"):
        assert torch.equal(a[k], b[k]), k
    for k in ("color", "color_base"):
        err = float((a[k] - b[k]).abs().max())
        assert err <= 2e-6, (k, err)
        assert float(b[k].abs().max()) > 1e-3
    # with autograd on, the per-sample colours go through memory (the backward reads them): the unfused result, bit for bit
    out_g = rend.render(o, d, near, far, **kw)
    assert torch.equal(out_g["color"], b["color"]) and out_g["color"].requires_grad


def test_sigmoid_head_row_sums_are_the_weighted_block_sums(dev):
    """NudfChainStep.row_w / row_sums on the colour engine's chain: per-32-point sums of weight x colour against the same
    launch's per-point colours (float64 sums), and the launch is refused outside the SIGMOIDN steps' kernel."""
    from common import build_modules, perturb_
    from neuraludf_amd import mlp
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0))
    eng = mods["color"].to(dev).engine()
    P, S = 64 * 37 + 32, 32
    g = torch.Generator().manual_seed(1)
    CIN = torch.zeros(mlp.pad_rows(P), eng.cin_ld, device=dev)
    CIN[:P, :eng.F + 3] = torch.randn(P, eng.F + 3, generator=g).to(dev) * 0.5
    rays_d = torch.nn.functional.normalize(torch.randn(P // S, 3, generator=g), dim=-1).to(dev)
    w = torch.zeros(mlp.pad_rows(P), device=dev)
    w[:P] = torch.rand(P, generator=g).to(dev)
    with torch.no_grad():
        cb, col, _, _ = eng.forward(CIN[:P], rays_d, S, P, keep_state=False)
        sb, sc, _, _ = eng.forward(CIN[:P], rays_d, S, P, keep_state=False, row_w=w)
    for sums, colours in ((sb, cb), (sc, col)):
        ref = (colours.double() * w[:P, None].double()).reshape(P // 32, 32, 3).sum(1)
        assert float((sums[:P // 32, :3].double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


def test_colour_finish_arguments_are_checked(dev):
    s = torch.zeros(8, 4, device=dev)
    out = torch.zeros(2, 3, device=dev)
    with pytest.raises(RuntimeError):
        call("nudf_composite_colour_finish", ptr(s), ptr(s), 2, 48, None, None, ptr(out), ptr(out))      # S % 32 != 0
    with pytest.raises(RuntimeError):
        call("nudf_composite_colour_finish", ptr(s), None, 2, 64, None, None, ptr(out), ptr(out))        # one of a pair missing
