"""The bf16x3 operand mode (mlp.PRECISION = "bf16x3", the library default; NudfChainStep.prec / NudfGemmTNGroup.prec = 3):
fp32 products EMULATED on the bf16 matrix pipe -- both operands split exactly into three bf16 parts, six partial products,
fp32 accumulation.  The claim to hold up is "fp32-level accuracy", so every quantity is measured against a FLOAT64
evaluation of the same network (the oracle run in double) next to the exact-fp32 MFMA kernels:

  * UDF value, features, d udf / d x at 4 096 points: the error of the bf16x3 chains against float64 is no larger than
    1.5x the error of the exact-fp32 chains (it is smaller on most quantities: the dropped partial products are bounded
    by 2^-23 |x| |y|, one fp32 rounding, and the accumulation is the same fp32);
  * every chain output and all parameter gradients of the full backward (second order included): bf16x3 against exact
    fp32 -- 3e-5 in the max norm on the smooth (softplus) network, the ReLU networks' gradients in the 2-norm;
  * the weight-gradient GEMM at ragged widths: split-image kernel (gemm_tn3_group_kernel), the generic kernel's
    split-on-the-way-out loop and the exact fp32 kernel against the float64 contraction;
  * the split itself: hi + mid + lo == x bit for bit, on the packed weight planes the chains read.

(The parity tests proper -- tests/test_gpu_fullsize_parity.py and friends, against fixtures the reference produced -- run
in this mode by default, with unchanged tolerances.)"""
import numpy as np
import pytest
import torch

from common import build_modules, perturb_, state_dicts, oracle_nets
from oracle import udf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _err(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    return float((a - ref).abs().max() / ref.abs().max().clamp(min=1e-30))


def test_chain_outputs_against_float64(dev):
    from neuraludf_amd import mlp
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0))
    sds = state_dicts(mods)
    udf = mods["udf"].to(dev)
    eng = udf.engine()
    P = 4096
    g = torch.Generator().manual_seed(0)
    x = torch.rand(P, 3, generator=g) * 2 - 1
    n64 = oracle_nets(sds, dtype=torch.float64)
    with torch.no_grad():
        ref = O.udf_forward(n64.udf, x.double())
    gref = O.udf_gradient(n64.udf, x.double(), create_graph=False)
    res = {}
    # "bf16x3": six bf16 products on every sweep (mlp.FWD_F16X2 = "0"); "f16x2": the library default -- THREE fp16 products
    # (NudfChainStep.prec 4) on every sweep, the backward ones with per-tile scaling (NudfChain.tile_scale)
    for mode, prec, fwd in (("fp32", "fp32", "1"), ("bf16x3", "bf16x3", "0"), ("f16x2", "bf16x3", "1")):
        mlp.set_precision(prec)
        mlp.set_fwd_split(fwd)
        eng.invalidate() if hasattr(eng, "invalidate") else None
        xd = x.to(dev)
        st = eng.forward(xd, need_grad_state=True, feat_ld=288)
        gr, _ = eng.gradient(xd, st)
        torch.cuda.synchronize()
        res[mode] = dict(udf=_err(st["udf"][:P].reshape(-1), ref[:, 0]), feat=_err(st["feat"][:P, :256], ref[:, 1:]),
                         grad=_err(gr[:P], gref))
    print("max error / max|ref| against float64:", res)
    for mode in ("bf16x3", "f16x2"):
        for k in ("udf", "feat", "grad"):
            assert res[mode][k] <= 1.5 * res["fp32"][k] + 1e-8, (mode, k, res)
            assert res[mode][k] < 2e-5, (mode, k, res)


def test_full_backward_parameter_gradients_match_exact_fp32(dev):
    """every chain launch of a train step's MLP work (tests/chain_sweeps.py): values and all parameter gradients of the
    bf16x3 mode against the exact-fp32 kernels."""
    import chain_sweeps as CS
    from neuraludf_amd import mlp
    outs = {}
    for mode, prec, fwd in (("fp32", "fp32", "1"), ("bf16x3", "bf16x3", "0"), ("f16x2", "bf16x3", "1")):
        mlp.set_precision(prec)
        mlp.set_fwd_split(fwd)
        outs[mode] = {k: v.detach().float().clone() for k, v in CS.sweeps(dev, 8192, 0, seed=3).items()}
        torch.cuda.synchronize()
    _compare_with_exact_fp32(outs["fp32"], outs["bf16x3"], "bf16x3 everywhere")
    _compare_with_exact_fp32(outs["fp32"], outs["f16x2"], "f16x2 on every sweep, the backward ones tile-scaled (the default)")


def _compare_with_exact_fp32(a, b, what):
    assert set(a) == set(b)
    # The UDF network is smooth (softplus): max-norm.  The colour net and the NeRF are ReLU networks: a pre-activation within
    # an ulp of zero takes the other branch in one of the two modes (a handful of the ~10^7 unit evaluations here), which
    # changes that point's gradient by a finite amount in EITHER direction -- fp32 summation orders differ from each other
    # in the same way -- so everything downstream of a ReLU mask is held in the 2-norm, plus a bound on how many elements
    # moved at all.
    smooth = {"udf", "sign", "feat", "X4", "X8", "g", "DA0", "DA3", "DA7", "uo"}
    worst = ("", 0.0)
    for k in a:
        if a[k].dtype != torch.float32 or a[k].numel() == 0:
            continue
        d = (a[k] - b[k]).double()
        if k in smooth:
            e = float(d.abs().max() / a[k].abs().max().clamp(min=1e-30))
        else:
            e = float(d.norm() / a[k].double().norm().clamp(min=1e-30))
            moved = int((d.abs() > 1e-3 * a[k].abs().max()).sum())
            assert moved <= max(2, int(2e-4 * d.numel())), (k, moved, d.numel())
        if e > worst[1]:
            worst = (k, e)
        # (3e-3: the bound tests/test_gpu_chain_rows.py holds the exact-fp32 kernels to AGAINST EACH OTHER on these tensors)
        assert e < (3e-5 if k in smooth else 3e-3), (k, e)
    print(f"largest relative difference {what} vs exact fp32:", worst)


@pytest.mark.parametrize("M", [333, 4096])
def test_weight_gradient_gemm_against_float64(dev, M):
    from neuraludf_amd import mlp, _lib
    g = torch.Generator().manual_seed(5)
    shapes = [(256, 256), (217, 256), (256, 40), (3, 128), (129, 72), (1, 256)]
    ops = []
    for NA, NB in shapes:
        lda, ldb = (NA + 3) // 4 * 4, (NB + 3) // 4 * 4
        A = torch.randn(M, lda, generator=g) * torch.exp(torch.randn(1, lda, generator=g))      # columns of mixed scale
        B = torch.randn(M, ldb, generator=g)
        ops.append((A, B, NA, NB))
    ref = [((A[:, :NA].double().t() @ B[:, :NB].double()), A[:, :NA].double().sum(0)) for A, B, NA, NB in ops]

    def run(mode, flags=0):
        mlp.set_precision(mode)
        old = _lib.lib().nudf_set_tn_flags(flags)
        try:
            jobs = [(A.to(dev), NA, B.to(dev), NB, torch.zeros(mlp.pad32(NA), B.shape[1], device=dev),
                     torch.zeros(mlp.pad32(NA), device=dev)) for A, B, NA, NB in ops]
            mlp.gemm_tn_grouped(jobs, M)
            torch.cuda.synchronize()
        finally:
            _lib.lib().nudf_set_tn_flags(old)
        return [(_err(j[4][:NA, :NB], r[0]), _err(j[5][:NA], r[1])) for j, (A, B, NA, NB), r in zip(jobs, ops, ref)]

    e32 = run("fp32")
    e3 = run("bf16x3")                 # split-image kernel
    e3g = run("bf16x3", flags=512)     # generic kernel, split on the way out of the fp32 image
    print("C error vs float64, exact fp32 / split image / generic split:", [(round(a[0] * 1e7, 2), round(b[0] * 1e7, 2), round(c[0] * 1e7, 2))
                                                                          for a, b, c in zip(e32, e3, e3g)], "x 1e-7")
    for (c32, b32), (c3, b3), (c3g, b3g) in zip(e32, e3, e3g):
        assert c3 <= 1.5 * c32 + 2e-7 and c3g <= 1.5 * c32 + 2e-7, (c32, c3, c3g)
        # bias sums: fp32 additions of the UNSPLIT values in every kernel, in each kernel's own association (the split-image
        # kernel: a tree over a k-step's 16 rows, a running sum over the steps) -- fp32 summation noise, nowhere near 2^-9
        assert b3 <= 3e-6 and b3g <= 3e-6 and b32 <= 3e-6, (b32, b3, b3g)


@pytest.mark.parametrize("M,rpb", [(8192, 1024), (5000, 640)])
def test_wide_weight_gradient_kernel_equals_the_128x128_kernel_bit_for_bit(dev, M, rpb):
    """gemm_tn3w_group_kernel (one 8-wave workgroup per pair of vertically adjacent 128 x 128 tiles, double-buffered split
    image, the two waves of a SIMD running the halves of a k-step in opposite order) against gemm_tn3_group_kernel
    (the default; the wide kernel is opt-in, NUDF_TN_FLAGS bit 1024) on the SAME row chunks: same k-steps, same MFMA order per accumulator, same workspace slots and
    reduce -> identical C and dbias, for paired tiles, unpaired ones (NA <= 128, a ragged second tile row) and a ragged last
    k-step."""
    from neuraludf_amd import mlp, _lib
    g = torch.Generator().manual_seed(7)
    shapes = [(256, 256), (217, 256), (256, 40), (3, 128), (129, 72), (1, 256), (256, 224)]
    ops = []
    for NA, NB in shapes:
        lda, ldb = (NA + 3) // 4 * 4, (NB + 3) // 4 * 4
        ops.append(((torch.randn(M, lda, generator=g) * torch.exp(torch.randn(1, lda, generator=g))).to(dev),
                    torch.randn(M, ldb, generator=g).to(dev), NA, NB))
    base = mlp.PRECISION

    def run(flags, assign):
        old = _lib.lib().nudf_set_tn_flags(flags)
        try:
            fill = torch.empty if assign else torch.zeros
            jobs = [(A, NA, B, NB, fill(mlp.pad32(NA), B.shape[1], device=dev), fill(mlp.pad32(NA), device=dev))
                    for A, B, NA, NB in ops]
            mlp.gemm_tn_grouped(jobs, M, assign=assign, rows_per_block=rpb)
            torch.cuda.synchronize()
        finally:
            _lib.lib().nudf_set_tn_flags(old)
        return [(j[4][:NA, :NB].clone(), j[5][:NA].clone()) for j, (A, B, NA, NB) in zip(jobs, ops)]

    try:
        mlp.set_precision("bf16x3")
        for assign in (False, True):
            wide, narrow = run(1024, assign), run(0, assign)
            for (cw, bw), (cn, bn), (A, B, NA, NB) in zip(wide, narrow, ops):
                assert torch.equal(cw, cn), (NA, NB, assign, float((cw - cn).abs().max()))
                assert torch.equal(bw, bn), (NA, NB, assign)
                ref = A[:, :NA].double().t() @ B[:, :NB].double()
                assert _err(cw, ref) < 2e-6
    finally:
        mlp.set_precision(base)


def test_packed_weight_planes_sum_to_the_weight_bit_for_bit(dev):
    """NudfPackFrag.dtype 3: the three bf16 planes of a fragment-ordered weight copy add up to the fp32 weight exactly."""
    from neuraludf_amd import mlp
    from neuraludf_amd.models import fields
    mods = build_modules(fields, seed=0)
    udf = mods["udf"].to(dev)
    mlp.set_precision("bf16x3")
    mlp.BWD_F16X2 = False      # (with the f16x2 backward sweeps no sweep of the default mode reads a three-plane copy of W^T)
    eng = udf.engine()
    x = torch.rand(64, 3, device=dev)
    eng.forward(x, need_grad_state=False, udf_only=True)            # packs the weights
    pl = eng.layers[2]
    f3 = pl._frags["fwd@bf16x3"]
    K, N = pl.inp, pl.out
    NT = (N + 31) // 32
    G16 = mlp.k8(K) // 16
    planes = f3.view(torch.int16).reshape(G16, NT, 3, 64, 8)
    w = (planes.to(torch.int32) << 16).view(torch.float32)           # bf16 -> fp32, exact
    total = (w[:, :, 0] + w[:, :, 1]) + w[:, :, 2]                   # hi + mid exact (<= 16 bits), + lo exact (24 bits)
    # element (g, T, lane, j) <-> B[16 g + 8 (lane >> 5) + j][32 T + (lane & 31)], B = W^T
    Wt = pl.Wt[:K, :N]
    ref = torch.zeros(G16 * 16, NT * 32, device=dev)
    ref[:K, :N] = Wt
    ref = ref.reshape(G16, 2, 8, NT, 32).permute(0, 3, 1, 4, 2).reshape(G16, NT, 64, 8)
    assert torch.equal(total, ref)
    assert float(w[:, :, 1].abs().max()) > 0 and float(w[:, :, 2].abs().max()) > 0


def test_packed_f16x2_planes_reproduce_the_weight(dev):
    """NudfPackFrag.dtype 4: the two fp16 planes of a fragment-ordered weight copy are hi = fp16(w) and lo = fp16((w - hi) 2^11)
    -- recomputed here with torch's fp16 rounding, bit for bit -- and hi + 2^-11 lo is within 2^-22 |w| of the fp32 weight."""
    from neuraludf_amd import mlp
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0))
    udf = mods["udf"].to(dev)
    mlp.set_precision("bf16x3")
    mlp.set_fwd_split("1")
    eng = udf.engine()
    x = torch.rand(64, 3, device=dev)
    eng.forward(x, need_grad_state=False, udf_only=True)            # packs the weights
    for li, kind, tr in ((2, "fwd@f16x2", True), (3, "bwd@f16x2", False)):
        pl = eng.layers[li]
        if kind not in pl._frags:
            st = eng.forward(x, need_grad_state=True, feat_ld=288)
            eng.gradient(x, st)
        f2 = pl._frags[kind]
        K, N = (pl.inp, pl.out) if tr else (pl.out, pl.inp)
        NT, G16 = (N + 31) // 32, mlp.k8(K) // 16
        planes = f2.view(torch.float16).reshape(G16, NT, 2, 64, 8)
        B = (pl.Wt[:K, :N] if tr else pl.W[:K, :N]).contiguous()
        ref = torch.zeros(G16 * 16, NT * 32, device=dev)
        ref[:K, :N] = B
        ref = ref.reshape(G16, 2, 8, NT, 32).permute(0, 3, 1, 4, 2).reshape(G16, NT, 64, 8)
        hi = ref.half()
        lo = ((ref - hi.float()) * 2048.0).half()
        assert torch.equal(planes[:, :, 0], hi), kind
        assert torch.equal(planes[:, :, 1], lo), kind
        back = planes[:, :, 0].double() + planes[:, :, 1].double() / 2048.0
        assert float((back - ref.double()).abs().max()) <= 2.0 ** -22 * float(ref.abs().max()), kind
        assert float(planes[:, :, 1].float().abs().max()) > 0


def test_non_finite_values_stay_non_finite(dev):
    """The split of +-inf is inf - inf = NaN in the remainder parts: an infinite activation comes out of a bf16x3 layer as
    NaN where the exact-fp32 kernels may propagate inf.  What must hold in BOTH modes is that a non-finite
    value is never laundered into a finite one: with the bias of a hidden layer at +inf every point's udf, features and
    input gradient are non-finite, and the renderer's status word (nudf_set_status_flag) would report it downstream."""
    from neuraludf_amd import mlp
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0))
    udf = mods["udf"].to(dev)
    with torch.no_grad():
        udf.lin3.bias.fill_(float("inf"))
    eng = udf.engine()
    x = (torch.rand(2048, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
    for mode in ("fp32", "bf16x3"):
        mlp.set_precision(mode)
        eng.invalidate()
        st = eng.forward(x, need_grad_state=True, feat_ld=288)
        gr, _ = eng.gradient(x, st)
        torch.cuda.synchronize()
        assert not bool(torch.isfinite(st["udf"][:2048]).any()), mode
        assert not bool(torch.isfinite(st["feat"][:2048, :256]).all(dim=1).any()), mode
        assert not bool(torch.isfinite(gr[:2048]).all(dim=1).any()), mode


def test_f16x2_operand_beyond_fp16_range_comes_out_non_finite_not_wrong(dev):
    """The forward-order sweeps' f16x2 split (NudfChainStep.prec 4) holds its operands in fp16's range: an activation above 65 504
    (here: a hidden layer's bias at 1e5, so softplus passes 1e5 on) has hi = fp16(x) = inf and comes out of the next layer as a
    NON-FINITE value -- which the renderer's status word reports -- where the exact-fp32 kernels and bf16x3 (fp32's exponent range)
    return a finite one.  The documented price of three products instead of six (include/nudf.h): never a finite
    wrong number."""
    from neuraludf_amd import mlp
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0))
    udf = mods["udf"].to(dev)
    with torch.no_grad():
        udf.lin2.bias.fill_(1.0e5)
    eng = udf.engine()
    x = (torch.rand(2048, 3, generator=torch.Generator().manual_seed(2)) * 2 - 1).to(dev)
    out = {}
    for mode, prec, fwd in (("fp32", "fp32", "1"), ("bf16x3", "bf16x3", "0"), ("f16x2", "bf16x3", "1")):
        mlp.set_precision(prec)
        mlp.set_fwd_split(fwd)
        eng.invalidate()
        out[mode] = eng.forward(x, need_grad_state=False, udf_only=True)["udf"][:2048].clone()
        torch.cuda.synchronize()
    assert bool(torch.isfinite(out["fp32"]).all()) and bool(torch.isfinite(out["bf16x3"]).all())
    assert not bool(torch.isfinite(out["f16x2"]).any())
