"""Round-6 mechanisms, each against the form it replaces:

  * the second-order term of the UDF double backward formed by the adjoint sweep from R and DA (NudfChainStep.X3, mlp.EX_FLY)
    against the stored form (the tangent sweep writes EX, the adjoint sweep reads it) -- every parameter gradient, in the
    emulated-fp32 mode and in the 16-bit mode;
  * the memoized chain / repack descriptors and the rest of the fast host path (NUDF_HOST_FAST) against freshly filled
    descriptors: eager train steps end in bit-identical parameters, and the memo is actually hit;
  * the packed seed of the reverse sweep (16-bit state) against the fp32-tile kernel that shares the code path (bit-identity is
    held by tests/test_gpu_chain_t16.py; here: the input gradient against the fp32 mode)."""
import pytest
import torch

from common import build_modules, perturb_

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _udf_backward(dev, P, seed=3, seed_scale=None):
    """every parameter gradient of the UDF network's double backward for random loss adjoints; seed_scale: a number or a [P]
    tensor the three adjoints (d udf, d feat, d grad) of every point are multiplied by"""
    from neuraludf_amd import mlp
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0))
    udf = mods["udf"].to(dev)
    eng = udf.engine()
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    d_udf = (torch.randn(P, generator=g) * 1e-4).to(dev)
    d_g = (torch.randn(P, 3, generator=g) * 1e-5).to(dev)
    d_feat = (torch.randn(P, 288, generator=g) * 1e-5).to(dev)
    if seed_scale is not None:
        sc = seed_scale.to(dev) if torch.is_tensor(seed_scale) else torch.full((P,), float(seed_scale), device=dev)
        d_udf, d_g, d_feat = d_udf * sc, d_g * sc[:, None], d_feat * sc[:, None]
    st = eng.forward(x, need_grad_state=True, feat_ld=288)
    gr, DA = eng.gradient(x, st)
    grads = eng.backward(x, st, DA, d_udf, d_feat, 288, d_g)
    torch.cuda.synchronize()
    return [t.detach().clone() for t in grads]


@pytest.mark.parametrize("mode,P", [("bf16x3", 8192), ("bf16x3", 19200), ("mixed16", 19200)])
def test_second_order_term_formed_in_the_adjoint_sweep_equals_the_stored_form(dev, mode, P):
    from neuraludf_amd import mlp
    mlp.set_precision(mode)
    old = mlp.EX_FLY
    try:
        mlp.EX_FLY = False
        stored = _udf_backward(dev, P)
        mlp.EX_FLY = True
        fly = _udf_backward(dev, P)
    finally:
        mlp.EX_FLY = old
    assert len(stored) == len(fly) >= 27
    worst = 0.0
    for a, b in zip(stored, fly):
        den = float(a.abs().max())
        if den == 0.0:
            assert float(b.abs().max()) == 0.0
            continue
        worst = max(worst, float((a - b).abs().max()) / den)
    # emulated fp32: the two forms differ by the roundings of R / (s scale); the 16-bit mode rounds R, DA and EX to bf16 in
    # different places (EX itself is never rounded in the new form): the mode's own resolution
    assert worst < (2e-5 if mode == "bf16x3" else 2e-2), worst
    print(f"{mode} P={P}: largest relative difference of a parameter gradient, formed vs stored second-order term: {worst:.2e}")


def test_fast_host_path_trains_bit_identically_and_hits_the_descriptor_memo(dev):
    from neuraludf_amd import _lib, mlp, synth
    from neuraludf_amd import train as T
    from neuraludf_amd.train import Trainer
    rconf = dict(n_samples=32, n_importance=16, n_outside=8, up_sample_steps=2, perturb=1.0)
    rays = synth.make_rays(synth.make_scene("tiny"), 0, 96, seed=5)
    batch = {k: v.to(dev) for k, v in rays.items()}

    def run(fast):
        old = (_lib.HOST_FAST, mlp.CHAIN_MEMO, T.SINGLE_THREAD_BACKWARD)
        _lib.HOST_FAST, mlp.CHAIN_MEMO, T.SINGLE_THREAD_BACKWARD = fast, fast, fast
        mlp._CHAIN_MEMO.clear()
        h0 = mlp.chain_memo_hits
        try:
            tr = Trainer(dev, rconf, seed=0, fused_adam=True)
            losses = []
            for _ in range(6):
                l, _ = tr.step(batch, cos_anneal_ratio=0.8, flip_saturation=0.9, perturb_overwrite=0)
                losses.append(float(l))
            torch.cuda.synchronize()
            params = [p.detach().clone() for m in tr.modules().values() for p in m.parameters()]
        finally:
            _lib.HOST_FAST, mlp.CHAIN_MEMO, T.SINGLE_THREAD_BACKWARD = old
        return losses, params, mlp.chain_memo_hits - h0

    l0, p0, h0 = run(False)
    l1, p1, h1 = run(True)
    assert h0 == 0 and h1 >= 20, (h0, h1)          # 5 repeat steps x >= 7 chain launches reuse their descriptors
    assert l0 == l1, (l0, l1)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M", [333, 8192])
def test_f16x2_weight_gradient_gemm_against_float64(dev, M):
    """NudfGemmTNGroup.prec 4 (gemm_tn2_group_kernel): three fp16 MFMA products per fp32 product, each side scaled by a power of
    two taken from the device scalar max |x| of its operands.  A = loss adjoints (1e-7, columns of mixed scale), B = activations:
    the error against the float64 contraction is no larger than 1.5x the exact fp32 kernel's (as held for bf16x3), with the
    maximum reported, and with B unscaled (amax_b = None); a maximum under-reported by 2^12 is clamped: finite and coarse, not
    inf."""
    from neuraludf_amd import mlp, _lib
    g = torch.Generator().manual_seed(11)
    shapes = [(256, 256), (217, 256), (256, 40), (3, 128), (129, 72), (1, 256)]
    ops = []
    for NA, NB in shapes:
        lda, ldb = (NA + 3) // 4 * 4, (NB + 3) // 4 * 4
        A = torch.randn(M, lda, generator=g) * torch.exp(torch.randn(1, lda, generator=g)) * 1e-7
        B = torch.randn(M, ldb, generator=g).abs() * 0.3
        ops.append((A, B, NA, NB))
    ref = [((A[:, :NA].double().t() @ B[:, :NB].double()), A[:, :NA].double().sum(0)) for A, B, NA, NB in ops]
    amax = torch.stack([A.abs().max() for A, _, _, _ in ops]).max().reshape(1).to(dev)

    def run(mode, **kw):
        mlp.set_precision(mode)
        jobs = [(A.to(dev), NA, B.to(dev), NB, torch.zeros(mlp.pad32(NA), B.shape[1], device=dev),
                 torch.zeros(mlp.pad32(NA), device=dev)) for A, B, NA, NB in ops]
        mlp.gemm_tn_grouped(jobs, M, **kw)
        torch.cuda.synchronize()
        return [(_err(j[4][:NA, :NB], r[0]), _err(j[5][:NA], r[1]), bool(torch.isfinite(j[4]).all()))
                for j, (A, B, NA, NB), r in zip(jobs, ops, ref)]

    e32 = run("fp32")
    e3 = run("bf16x3")
    e2 = run("bf16x3", f16x2=True, amax_a=amax)
    e2b = run("bf16x3", f16x2=True, amax_a=amax, amax_b=torch.tensor([0.3 * 4.0], device=dev))
    low = run("bf16x3", f16x2=True, amax_a=amax / 4096.0)
    print("C error vs float64 x 1e-7: exact fp32 / bf16x3 / f16x2 / f16x2 both scaled:",
          [(round(a[0] * 1e7, 2), round(b[0] * 1e7, 2), round(c[0] * 1e7, 2), round(d[0] * 1e7, 2)) for a, b, c, d in zip(e32, e3, e2, e2b)])
    for (c32, b32, _), (c3, _, _), (c2, b2, f2), (c2b, _, _), (cl, _, fl) in zip(e32, e3, e2, e2b, low):
        assert c2 <= 1.5 * c32 + 2e-7 and c2b <= 1.5 * c32 + 2e-7, (c32, c3, c2, c2b)
        assert b2 <= 3e-6 and f2
        assert fl and cl < 1.0            # clamped operands: wrong by construction, but finite


def _err(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    return float((a - ref).abs().max() / ref.abs().max().clamp(min=1e-300))


# ---- f16x2 backward sweeps (NudfChain.tile_scale, mlp.BWD_F16X2) -----------------------------------------------------------
def test_backward_sweeps_scale_exactly_with_a_power_of_two_of_the_loss(dev):
    """The tangent / adjoint sweeps run every tile of points multiplied by its own power of two (so that adjoints of the loss fit
    fp16's exponent range) and the f16x2 weight-gradient GEMMs scale their operands the same way: multiplying every loss adjoint
    by 2^k must multiply every parameter gradient by exactly 2^k -- bit for bit -- far below and above fp16's range."""
    from neuraludf_amd import mlp
    mlp.set_precision("bf16x3")
    assert mlp.BWD_F16X2 and mlp._sweep_dtype("bwd") == "f16x2"
    base = _udf_backward(dev, 8192)
    assert all(bool(torch.isfinite(t).all()) for t in base) and max(float(t.abs().max()) for t in base) > 0.0
    for k in (-60, -20, 14):
        got = _udf_backward(dev, 8192, seed_scale=2.0 ** k)
        for i, (a, b) in enumerate(zip(base, got)):
            assert torch.equal(a * 2.0 ** k, b), (k, i, float((a * 2.0 ** k - b).abs().max()), float(b.abs().max()))


def test_every_backward_sweep_scales_exactly_with_a_power_of_two_of_its_seeds(dev):
    """The same property over ALL reverse chains of a step (tests/chain_sweeps.py): UDF tangent + adjoint, the colour network's
    view- and base-branch ReLU backward (the second one joins the first one's d VIN through X2: ADDMASK), the NeRF's (rank-1
    operand on a later step) -- every parameter gradient and d CIN times exactly 2^-30, every forward value unchanged."""
    import re
    import chain_sweeps as CS
    from neuraludf_amd import mlp
    mlp.set_precision("bf16x3")
    assert mlp._sweep_dtype("bwd") == "f16x2"
    base = {k: v.detach().clone() for k, v in CS.sweeps(dev, 8192, 0, seed=3).items()}
    got = CS.sweeps(dev, 8192, 0, seed=3, seed_scale=2.0 ** -30)
    torch.cuda.synchronize()
    n_grad = 0
    for k, a in base.items():
        if re.fullmatch(r"(p|cg|n)\d+|dCIN", k):
            assert float(a.abs().max()) > 0.0, k
            assert torch.equal(a * 2.0 ** -30, got[k]), (k, float((a * 2.0 ** -30 - got[k]).abs().max()), float(got[k].abs().max()))
            n_grad += 1
        else:
            assert torch.equal(a, got[k]), k
    assert n_grad >= 60, n_grad


def test_f16x2_backward_sweeps_against_the_range_free_bf16x3_sweeps(dev):
    """Same gradients as the six-product sweeps, to the rounding of two fp32 emulations -- with the loss adjoints of the points
    spread over 40 binades (every tile of 64 points its own scale, many of them far outside fp16's range) and a block of
    points whose adjoints are all zero."""
    from neuraludf_amd import mlp
    mlp.set_precision("bf16x3")
    P = 19200
    g = torch.Generator().manual_seed(11)
    sc = torch.exp2(-torch.randint(0, 40, (P // 64,), generator=g).float()).repeat_interleave(64)
    sc[640:1280] = 0.0
    sc[5000:5003] = 2.0 ** -70          # single points 30 binades below their tile's largest: they keep 2^-30 of it
    old = mlp.BWD_F16X2
    try:
        mlp.BWD_F16X2 = False
        ref = _udf_backward(dev, P, seed_scale=sc)
        mlp.BWD_F16X2 = True
        got = _udf_backward(dev, P, seed_scale=sc)
    finally:
        mlp.BWD_F16X2 = old
    worst = 0.0
    for a, b in zip(ref, got):
        assert bool(torch.isfinite(b).all())
        den = float(a.abs().max())
        if den == 0.0:
            assert float(b.abs().max()) == 0.0
            continue
        worst = max(worst, float((a - b).abs().max()) / den)
    assert worst < 2e-5, worst
    print(f"largest relative difference of a parameter gradient, f16x2 vs bf16x3 backward sweeps: {worst:.2e}")


def test_tile_scale_is_refused_on_a_sweep_that_is_not_linear(dev):
    from neuraludf_amd import _lib, mlp
    from neuraludf_amd.models import fields
    mlp.set_precision("bf16x3")
    udf = perturb_(build_modules(fields, seed=0))["udf"].to(dev)
    eng = udf.engine()
    eng.forward(torch.zeros(64, 3, device=dev), need_grad_state=False, feat_ld=288)      # (packs the operand fragments)
    pl = eng.layers[1]
    x = torch.zeros(64, mlp.k8(pl.inp), device=dev)
    out = torch.empty(64, pl.out, device=dev)
    cb = mlp.ChainBuilder(64, "LOAD", mlp.k8(pl.inp))
    cb.init_load(x, x.shape[1])
    cb.tile_scale()
    cb.step("SOFTPLUS", pl.frag(mlp._kind("fwd", "fwd")), mlp.k8(pl.inp), pl.out, bias=pl.bias, C1=out)
    with pytest.raises(_lib.NudfError):
        cb.launch()

