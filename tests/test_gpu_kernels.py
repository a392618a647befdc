"""Stage-wise parity of the HIP kernels (through the C ABI) against the CPU oracle on identical
inputs.  Tolerances: 1e-4 * max(1, |ref|_inf) on values (north_star), 1e-3 TRUE relative per tensor on gradients
(common.grel: no floor at 1)."""
import math

import pytest
import torch

from common import CONF, build_modules, perturb_, state_dicts, oracle_nets, grel, grel2
from oracle import udf_oracle as O

pytestmark = pytest.mark.gpu

VTOL = 1e-4
GTOL = 1e-3


def rel(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1.0))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def nets(dev):
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0))
    sds = state_dicts(mods)
    for m in mods.values():
        m.to(dev)
    return mods, sds


# ---------------------------------------------------------------------------------------------
def test_gemm_nn_plain_and_epilogues(dev):
    from neuraludf_amd import mlp
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in [(1000, 256, 256), (257, 217, 64), (4096, 257, 288), (130, 13, 160), (64, 1, 256)]:
        A = torch.randn(M, K, generator=g)
        B = torch.randn(K, mlp.pad32(N), generator=g) * 0.1
        bias = torch.randn(N, generator=g)
        ref = A.double() @ B[:, :N].double() + bias.double()
        Ad, Bd, bd = A.to(dev), B.to(dev), bias.to(dev)
        C = torch.full((M, N), float("nan"), device=dev)
        mlp.gemm_nn(Ad, Bd, M, N, K, "NONE", C1=C, bias=bd)
        assert rel(C, ref.float()) < 1e-5, (M, N, K)
        C2 = torch.empty(M, N, device=dev)
        mlp.gemm_nn(Ad, Bd, M, N, K, "SOFTPLUS", C1=C, C2=C2, bias=bd, scale=0.5)
        r = ref.float() * 0.05
        # compare against softplus of the kernel's own pre-activation scale: recompute reference directly
        sp = torch.nn.functional.softplus(ref.float(), beta=100.0, threshold=20.0) * 0.5
        assert rel(C, sp) < 1e-5
        sg = torch.where(ref * 100 > 20, torch.ones_like(ref), torch.sigmoid(100 * ref)).float()
        assert rel(C2, sg) < 2e-4       # sigmoid(100 a): fp32 rounding of a (~1e-6) is amplified 25x
        # softplus'-from-stored-activation epilogues: X1 = softplus(a)/xs  ->  s = sigmoid(100 a)
        apre = torch.randn(M, N, generator=g) * 0.05
        xs = 1.4142135
        X1 = (torch.nn.functional.softplus(apre.double(), beta=100.0, threshold=20.0) / xs).float().to(dev)
        sref = torch.where(apre.double() * 100 > 20, torch.ones_like(apre.double()), torch.sigmoid(100 * apre.double()))
        X2 = torch.randn(M, N, generator=g).to(dev)
        acc = A.double() @ B[:, :N].double()
        mlp.gemm_nn(Ad, Bd, M, N, K, "BWD", C1=C, X1=X1, X2=X2, scale=0.7, xscale=xs)
        assert rel(C, (acc * 0.7 * sref + X2.cpu().double()).float()) < 1e-5
        mlp.gemm_nn(Ad, Bd, M, N, K, "TANGENT", C1=C, C2=C2, X1=X1, X2=X2, scale=0.7, xscale=xs)
        assert rel(C, (acc * 0.7 * sref).float()) < 1e-5
        assert rel(C2, (acc * X2.cpu().double() * 100.0 * (1.0 - sref)).float()) < 1e-5


def test_gemm_tn(dev):
    from neuraludf_amd import mlp
    g = torch.Generator().manual_seed(1)
    for (M, NA, NB) in [(5000, 256, 256), (1234, 217, 64), (3000, 13, 160), (700, 257, 288)]:
        A = torch.randn(M, mlp.pad32(NA), generator=g)
        B = torch.randn(M, mlp.pad32(NB), generator=g)
        A2 = torch.randn(M, mlp.pad32(NA), generator=g)
        B2 = torch.randn(M, mlp.pad32(NB), generator=g)
        ref = A[:, :NA].double().t() @ B.double() + A2[:, :NA].double().t() @ B2.double()
        C = torch.zeros(mlp.pad32(NA), mlp.pad32(NB), device=dev)
        db = torch.zeros(NA, device=dev)
        mlp.gemm_tn(A.to(dev), NA, B.to(dev), C, NA, mlp.pad32(NB), M, dbias=db, A2=A2.to(dev), na2=NA, B2=B2.to(dev))
        assert rel(C[:NA], ref.float()) < 2e-5, (M, NA, NB)
        assert rel(db, A[:, :NA].double().sum(0).float()) < 2e-5


@pytest.mark.parametrize("M", [65, 1000, 20000])
def test_gemm_tn_grouped_ragged_layouts(dev, M):
    """the grouped weight-gradient launch on ragged layer widths: tiles whose live 32 x 32 sub-tiles are dealt along
    the rows (217 x 256), along the columns (256 x 40), a single live row (1 x 256), both ragged (300 x 100), M not a
    multiple of the k-step -- workspace (two-pass, run-to-run identical) and atomics paths against float64, every
    scheduling variant (cost-weighted and equal row chunks)."""
    from neuraludf_amd import _lib, mlp
    g = torch.Generator().manual_seed(11)
    shapes = [(256, 40), (217, 256), (256, 256), (1, 256), (300, 100), (33, 7), (129, 129)]
    jobs, refs = [], []
    for NA, NB in shapes:
        lda, ldb = max(4, (NA + 3) // 4 * 4), max(4, (NB + 3) // 4 * 4)
        A = torch.randn(M, lda, generator=g)
        B = torch.randn(M, ldb, generator=g)
        refs.append((A[:, :NA].double().t() @ B[:, :NB].double(), A[:, :NA].double().sum(0)))
        jobs.append((A.to(dev), NA, B.to(dev), NB, torch.zeros(mlp.pad32(NA), ldb, device=dev),
                     torch.zeros(mlp.pad32(NA), device=dev)))

    def run():
        for j in jobs:
            j[4].zero_(); j[5].zero_()
        mlp.gemm_tn_grouped(jobs, M)
        return [(j[4].clone(), j[5].clone()) for j in jobs]

    try:
        for flags, det in [(0, True), (0, False), (16, True), (16, False)]:
            _lib.lib().nudf_set_tn_flags(flags)
            mlp.TN_DETERMINISTIC = det
            out = run()
            for (NA, NB), (C, db), (rC, rb) in zip(shapes, out, refs):
                assert rel(C[:NA, :NB], rC.float()) < 2e-5, (flags, det, NA, NB)
                assert rel(db[:NA], rb.float()) < 2e-5, (flags, det, NA, NB)
                assert float(C[NA:].abs().max()) == 0.0 if C.shape[0] > NA else True
            if det:
                again = run()
                assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(out, again))
    finally:
        _lib.lib().nudf_set_tn_flags(0)
        mlp.TN_DETERMINISTIC = True


@pytest.mark.parametrize("M", [77, 5000])
def test_gemm_tn_grouped_bf16_operands(dev, M):
    """operands stored as bf16 (the 16-bit mode's saved state) in every combination with fp32 ones: the kernel widens
    them on the way into LDS, so with fp32 MFMAs the result equals the contraction of the ROUNDED operands to fp32
    rounding -- including the bias sums taken from a bf16 A operand and the ragged last k-step."""
    from neuraludf_amd import mlp
    g = torch.Generator().manual_seed(12)
    shapes = [(256, 256, True, True), (217, 256, True, False), (256, 40, False, True), (3, 128, True, True), (129, 72, True, True)]
    jobs, refs = [], []
    for NA, NB, a16, b16 in shapes:
        lda, ldb = (NA + 7) // 8 * 8, (NB + 7) // 8 * 8
        A = torch.randn(M, lda, generator=g)
        B = torch.randn(M, ldb, generator=g)
        Ad = A.to(dev).to(torch.bfloat16) if a16 else A.to(dev)
        Bd = B.to(dev).to(torch.bfloat16) if b16 else B.to(dev)
        Ar, Br = Ad.float().cpu().double(), Bd.float().cpu().double()
        refs.append((Ar[:, :NA].t() @ Br[:, :NB], Ar[:, :NA].sum(0)))
        jobs.append((Ad, NA, Bd, NB, torch.zeros(mlp.pad32(NA), ldb, device=dev), torch.zeros(mlp.pad32(NA), device=dev)))
    base = mlp.PRECISION          # the library default (bf16x3), or whatever NUDF_PRECISION selects
    mlp.gemm_tn_grouped(jobs, M)
    for (NA, NB, a16, b16), j, (rC, rb) in zip(shapes, jobs, refs):
        assert rel(j[4][:NA, :NB], rC.float()) < 2e-5, (NA, NB, a16, b16)
        assert rel(j[5][:NA], rb.float()) < 2e-5, (NA, NB, a16, b16)


@pytest.mark.parametrize("M", [77, 1000, 5000])
def test_gemm_tn_grouped_16bit_mfma_mode(dev, M):
    """config-5 mode (NudfGemmTNGroup.prec != 0): 16-bit MFMAs over bf16-ROUNDED operands with fp32 accumulation, from
    bf16- and fp32-stored operands in every combination, ragged widths, a ragged last k-step.  The packed k-pair LDS
    image (gemm_tn16_group_kernel) must give the contraction of the rounded operands to fp32 rounding, bit-identical C
    to the generic kernel's 16-bit loop (NUDF_TN_FLAGS bit 256), and the bias sums of the operands AS STORED."""
    from neuraludf_amd import mlp, _lib
    g = torch.Generator().manual_seed(14)
    shapes = [(256, 256, True, True), (217, 256, False, True), (256, 40, True, True), (3, 128, True, False), (129, 72, True, True),
              (1, 256, False, True)]
    jobs, refs = [], []
    for NA, NB, a16, b16 in shapes:
        lda, ldb = ((NA + 7) // 8 * 8, (NB + 7) // 8 * 8) if True else (NA, NB)
        if not a16:
            lda = (NA + 3) // 4 * 4          # an fp32 operand only needs a multiple of 4 (the two pieces clamp separately)
        A = torch.randn(M, lda, generator=g)
        B = torch.randn(M, ldb, generator=g)
        Ad = A.to(dev).to(torch.bfloat16) if a16 else A.to(dev)
        Bd = B.to(dev).to(torch.bfloat16) if b16 else B.to(dev)
        Ar, Br = Ad.to(torch.bfloat16).float().cpu().double(), Bd.to(torch.bfloat16).float().cpu().double()
        refs.append((Ar[:, :NA].t() @ Br[:, :NB], Ad.float().cpu().double()[:, :NA].sum(0)))
        jobs.append((Ad, NA, Bd, NB, torch.zeros(mlp.pad32(NA), ldb, device=dev), torch.zeros(mlp.pad32(NA), device=dev)))
    old = mlp.PRECISION
    mlp.PRECISION = "mixed16"
    try:
        mlp.gemm_tn_grouped(jobs, M)
        packed = [(j[4].clone(), j[5].clone()) for j in jobs]
        for j in jobs:
            j[4].zero_(); j[5].zero_()
        _lib.lib().nudf_set_tn_flags(256)
        mlp.gemm_tn_grouped(jobs, M)
    finally:
        _lib.lib().nudf_set_tn_flags(0)
        mlp.PRECISION = old
    for (NA, NB, a16, b16), j, (pC, pb), (rC, rb) in zip(shapes, jobs, packed, refs):
        assert rel(pC[:NA, :NB], rC.float()) < 2e-5, (NA, NB, a16, b16)
        assert rel(pb[:NA], rb.float()) < 2e-5, (NA, NB, a16, b16)
        assert torch.equal(pC, j[4]), (NA, NB, a16, b16)
        assert rel(pb[:NA], j[5][:NA]) < 2e-5


@pytest.mark.parametrize("M", [77, 1000, 5001])
def test_gemm_tn_grouped_16bit_packed_operands(dev, M):
    """the chains' 16-bit stored state is 4-POINT PACKED (include/nudf.h, NUDF_TN_A_P4 / _B_P4: element (r, c) at
    ((r // 4) ld + c) 4 + r % 4): the packed-image kernel copies its dwords straight into the MFMA operand image.  Same
    values, same MFMA order as the row-major bf16 operand: C is BIT-IDENTICAL, the bias sums agree to fp32 rounding (other
    summation order); ragged widths, a ragged last quad (M % 4 != 0), pad rows holding garbage, packed with fp32 partners;
    a packed operand outside the 16-bit mode / next to a row-major bf16 partner is refused."""
    from neuraludf_amd import mlp, _lib
    g = torch.Generator().manual_seed(15)
    shapes = [(256, 256, True, True), (217, 256, False, True), (256, 40, True, True), (3, 128, True, False), (129, 72, True, True),
              (1, 256, False, True)]
    Mp = mlp.pad_rows(M)
    jobs_rm, jobs_p4 = [], []
    for NA, NB, a16, b16 in shapes:
        lda, ldb = (NA + 7) // 8 * 8, (NB + 7) // 8 * 8
        if not a16:
            lda = (NA + 3) // 4 * 4
        A = torch.randn(Mp, lda, generator=g)       # rows >= M: garbage the kernel must mask
        B = torch.randn(Mp, ldb, generator=g)
        A[M:] = float("nan")
        B[M:] = float("nan")
        Ad, Bd = A.to(dev), B.to(dev)
        mk = lambda: (torch.zeros(mlp.pad32(NA), ldb, device=dev), torch.zeros(mlp.pad32(NA), device=dev))
        jobs_rm.append((Ad.to(torch.bfloat16) if a16 else Ad, NA, Bd.to(torch.bfloat16) if b16 else Bd, NB) + mk())
        Ap, Bp = (mlp.pack16(Ad) if a16 else Ad), (mlp.pack16(Bd) if b16 else Bd)
        if a16:
            assert torch.equal(mlp.unpack16(Ap)[:M], Ad.to(torch.bfloat16).float()[:M])
        jobs_p4.append((Ap, NA, Bp, NB) + mk())
    old = mlp.PRECISION
    mlp.PRECISION = "mixed16"
    try:
        mlp.gemm_tn_grouped(jobs_rm, M)
        mlp.gemm_tn_grouped(jobs_p4, M)
        again = [(j[0], j[1], j[2], j[3], torch.zeros_like(j[4]), torch.zeros_like(j[5])) for j in jobs_p4]
        mlp.gemm_tn_grouped(again, M)
        # refused combinations
        bad = jobs_p4[0]
        with pytest.raises(_lib.NudfError):     # packed next to a row-major bf16 partner
            mlp.gemm_tn_grouped([(bad[0], 256, jobs_rm[0][2], 256, torch.zeros_like(bad[4]), None)], M)
        mlp.PRECISION = "fp32"
        with pytest.raises(_lib.NudfError):     # packed operand, fp32 MFMA mode
            mlp.gemm_tn_grouped([(bad[0], 256, bad[2], 256, torch.zeros_like(bad[4]), None)], M)
    finally:
        mlp.PRECISION = old
    for (NA, NB, a16, b16), r, p, q in zip(shapes, jobs_rm, jobs_p4, again):
        assert torch.isfinite(p[4]).all() and torch.isfinite(p[5]).all(), (NA, NB)
        assert torch.equal(r[4], p[4]), (NA, NB, a16, b16)
        assert rel(p[5][:NA], r[5][:NA]) < 2e-5, (NA, NB, a16, b16)
        assert torch.equal(p[4], q[4]) and torch.equal(p[5], q[5])


@pytest.mark.parametrize("M", [77, 5000])
def test_gemm_tn_grouped_blocked_operands(dev, M):
    """fp32 operands in the BLOCKED layout of nudf.h (what the transposed-product chain kernel stores), alone and mixed
    with row-major ones, ragged widths and a ragged last k-step: same contraction and bias sums as the row-major call."""
    from neuraludf_amd import mlp
    g = torch.Generator().manual_seed(13)
    shapes = [(256, 256, True, True), (217, 256, True, True), (256, 40, True, False), (3, 128, False, True), (129, 72, True, True)]
    Mp = (M + 63) // 64 * 64
    jobs, refs = [], []
    for NA, NB, ablk, bblk in shapes:
        lda, ldb = (NA + 3) // 4 * 4, (NB + 3) // 4 * 4
        A = torch.zeros(Mp, lda); A[:M] = torch.randn(M, lda, generator=g)
        B = torch.zeros(Mp, ldb); B[:M] = torch.randn(M, ldb, generator=g)
        A[M:] = float("nan"); B[M:] = float("nan")          # rows past M must never reach a result
        refs.append((A[:M, :NA].double().t() @ B[:M, :NB].double(), A[:M, :NA].double().sum(0)))
        Ad, Bd = A.to(dev), B.to(dev)
        jobs.append((mlp.block(Ad) if ablk else Ad, NA, mlp.block(Bd) if bblk else Bd, NB,
                     torch.zeros(mlp.pad32(NA), ldb, device=dev), torch.zeros(mlp.pad32(NA), device=dev)))
        assert torch.equal(mlp.unblock(mlp.block(Ad)).nan_to_num(7.0), Ad.nan_to_num(7.0))
    mlp.gemm_tn_grouped(jobs, M)
    for (NA, NB, ablk, bblk), j, (rC, rb) in zip(shapes, jobs, refs):
        assert rel(j[4][:NA, :NB], rC.float()) < 2e-5, (NA, NB, ablk, bblk)
        assert rel(j[5][:NA], rb.float()) < 2e-5, (NA, NB, ablk, bblk)


def test_posenc_and_vjp(dev):
    from neuraludf_amd._lib import call, ptr
    g = torch.Generator().manual_seed(2)
    x = torch.randn(999, 3, generator=g) * 0.8
    e = O.posenc(x, 6)
    out = torch.zeros(999, 64, device=dev)
    xd = x.to(dev)
    call("nudf_posenc", ptr(xd), 3, 1, None, 3, 6, 1.0, 999, ptr(out), 64, 1.0, None, 0, 0.0)
    assert rel(out[:, :39], e) < 1e-5
    d = torch.randn(999, 64, generator=g)
    xg = x.clone().requires_grad_(True)
    (O.posenc(xg, 6) * d[:, :39]).sum().backward()
    gg = torch.empty(999, 3, device=dev)
    dd = d.to(dev)
    call("nudf_posenc_vjp", ptr(xd), 3, 3, 6, 1.0, 999, ptr(dd), 64, 1.0, None, 0, 0.0, ptr(gg))
    assert grel(gg, xg.grad) < 1e-5


# ---------------------------------------------------------------------------------------------
def test_udf_forward_gradient_and_param_grads(dev, nets):
    mods, sds = nets
    g = torch.Generator().manual_seed(3)
    P = 777
    x = torch.randn(P, 3, generator=g) * 0.7
    wy = torch.randn(P, 257, generator=g)
    wg = torch.randn(P, 3, generator=g)
    on = oracle_nets(sds, requires_grad=True)
    y_ref = O.udf_forward(on.udf, x)
    g_ref = O.udf_gradient(on.udf, x, create_graph=True)
    ((y_ref * wy).sum() + (g_ref * wg).sum()).backward()

    net = mods["udf"]
    net.zero_grad()
    udf, feat, grad = net.evaluate(x.to(dev), want_grad=True)
    assert rel(udf, y_ref[:, 0]) < VTOL
    assert rel(feat[:, :256], y_ref[:, 1:]) < VTOL
    assert rel(grad, g_ref) < VTOL
    wyd, wgd = wy.to(dev), wg.to(dev)
    ((udf * wyd[:, 0]).sum() + (feat[:, :256] * wyd[:, 1:]).sum() + (grad * wgd).sum()).backward()
    for n, p in net.named_parameters():
        assert p.grad is not None, n
        assert grel(p.grad, on.udf[n].grad) < GTOL, n
    # reference call surface
    with torch.no_grad():
        y = net(x.to(dev))
        assert rel(y, y_ref) < VTOL
        assert rel(net.udf_only(x.to(dev)), y_ref[:, 0]) < VTOL


@pytest.mark.parametrize("path", ["chain", "layers"])
def test_color_network_with_normals(dev, path):
    """ResidualRenderingNetwork in any mode but 'no_normal' (fields.py:456-461): base input [pts, n, -n, feat] with the
    normals detached -- reference call surface against the oracle, values and every parameter gradient."""
    from neuraludf_amd import mlp
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0, color_mode="idr"))
    sds = state_dicts(mods)
    net = mods["color"].to(dev)
    g = torch.Generator().manual_seed(6)
    P = 900
    pts, dirs = torch.randn(P, 3, generator=g) * 0.6, torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    nrm = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    feat = torch.randn(P, 256, generator=g) * 0.3
    w = [torch.randn(P, 3, generator=g), torch.randn(P, 3, generator=g), torch.randn(P, 10, generator=g)]
    on = oracle_nets(sds, requires_grad=True)
    fr = feat.clone().requires_grad_(True)
    ref = O.color_forward(on.color, pts, nrm, dirs, fr, O.ColorCfg(mode="idr", d_in=12))
    sum((a * b).sum() for a, b in zip(ref, w)).backward()
    try:
        if path == "layers":
            mlp.USE_CHAIN = False
        net.zero_grad()
        fd = feat.to(dev).requires_grad_(True)
        out = net(pts.to(dev), nrm.to(dev), dirs.to(dev), fd)
        for a, b in zip(out, ref):
            assert rel(a, b) < VTOL
        sum((a * b.to(dev)).sum() for a, b in zip(out, w)).backward()
    finally:
        mlp.USE_CHAIN = True
    assert grel(fd.grad, fr.grad) < GTOL
    for n, p in net.named_parameters():
        assert p.grad is not None, n
        assert grel(p.grad, on.color[n].grad) < GTOL, n


@pytest.mark.parametrize("udf_type", ["square", "sdf"])
@pytest.mark.parametrize("path", ["chain", "chain_tq", "layers"])
def test_udf_type_variants(dev, udf_type, path):
    """UDFNetwork.udf_out's other branches (fields.py:184-190; the confs' comment offers 'square'): values, d udf / dx and
    every parameter gradient of a loss on udf, features AND the gradient -- for 'square' that includes the term through
    f'' = 2 -- against the oracle, on the chain kernels (32-point tiles; transposed-product kernel) and the per-layer path."""
    from neuraludf_amd import mlp
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0, udf_type=udf_type))
    sds = state_dicts(mods)
    net = mods["udf"].to(dev)
    g = torch.Generator().manual_seed(4)
    P = 1500
    x = torch.randn(P, 3, generator=g) * 0.7
    wy = torch.randn(P, 257, generator=g)
    wg = torch.randn(P, 3, generator=g)
    cfg = O.UDFCfg(udf_type=udf_type)
    on = oracle_nets(sds, requires_grad=True)
    y_ref = O.udf_forward(on.udf, x, cfg)
    g_ref = O.udf_gradient(on.udf, x, cfg, create_graph=True)
    assert rel(O.udf_gradient_analytic(on.udf, x, cfg), g_ref) < 1e-5
    ((y_ref * wy).sum() + (g_ref * wg).sum()).backward()
    try:
        if path == "chain_tq":
            mlp.CHAIN_TILE = 66
        elif path == "layers":
            mlp.USE_CHAIN = False
        net.zero_grad()
        udf, feat, grad = net.evaluate(x.to(dev), want_grad=True)
        assert rel(udf, y_ref[:, 0]) < VTOL
        assert rel(feat[:, :256], y_ref[:, 1:]) < VTOL
        assert rel(grad, g_ref) < VTOL
        wyd, wgd = wy.to(dev), wg.to(dev)
        ((udf * wyd[:, 0]).sum() + (feat[:, :256] * wyd[:, 1:]).sum() + (grad * wgd).sum()).backward()
        with torch.no_grad():
            assert rel(net.udf_only(x.to(dev)), y_ref[:, 0]) < VTOL
    finally:
        mlp.CHAIN_TILE, mlp.USE_CHAIN = 0, True
    for n, p in net.named_parameters():
        assert p.grad is not None, n
        assert grel(p.grad, on.udf[n].grad) < GTOL, n


def test_sdf_network_class_on_the_hip_chains(dev):
    """the reference's SDFNetwork surface (sdf / sdf_hidden_appearance / gradient / forward, fields.py:84-112) served by the
    same chains as UDFNetwork with the identity head, against the oracle's forward with udf_type 'sdf'."""
    from neuraludf_amd.models import fields
    torch.manual_seed(5)
    net = fields.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1.0,
                            geometric_init=True, weight_norm=True, inside_outside=True)
    gp = torch.Generator().manual_seed(1)
    with torch.no_grad():          # trained-like weights (the init zeroes the encoding columns), as common.perturb_ does
        for p in net.parameters():
            p.add_(torch.randn(p.shape, generator=gp) * 0.02 * (p.abs().mean() + 0.05))
    sd = {n: t.detach().clone() for n, t in net.state_dict().items()}
    net = net.to(dev)
    x = torch.randn(700, 3, generator=torch.Generator().manual_seed(6)) * 0.6
    cfg = O.UDFCfg(udf_type="sdf")
    y_ref = O.udf_forward(sd, x, cfg)
    g_ref = O.udf_gradient(sd, x, cfg, create_graph=False)
    assert float(y_ref[:, 0].min()) < 0 < float(y_ref[:, 0].max())          # signed: the head is not the abs one
    xd = x.to(dev)
    assert rel(net.sdf(xd), y_ref[:, :1]) < VTOL
    assert rel(net.sdf_hidden_appearance(xd), y_ref) < VTOL and rel(net(xd), y_ref) < VTOL
    assert rel(net.gradient(xd)[:, 0], g_ref) < VTOL


@pytest.mark.parametrize("mode", ["chain64", "layers"])
def test_udf_other_paths_match_the_default(dev, nets, mode):
    """the 64-point-tile chain kernel (used for P > 16 k) and the per-layer GEMM path against the default
    (32-point tiles at this size, itself checked against the oracle above)."""
    from neuraludf_amd import mlp
    mods, _ = nets
    eng = mods["udf"].engine()
    g = torch.Generator().manual_seed(9)
    P = 1000
    x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    d_udf, d_feat, d_g = (torch.randn(P, generator=g).to(dev), torch.randn(P, 288, generator=g).to(dev),
                          torch.randn(P, 3, generator=g).to(dev))

    def run():
        st = eng.forward(x, True, 288)
        gr, DA = eng.gradient(x, st)
        grads = eng.backward(x, st, DA, d_udf, d_feat, 288, d_g)
        return [st["udf"], st["feat"][:, :256], gr] + list(grads)
    ref = run()
    try:
        if mode == "chain64":
            mlp.CHAIN_TILE = 64
        else:
            mlp.USE_CHAIN = False
        got = run()
    finally:
        mlp.CHAIN_TILE, mlp.USE_CHAIN = 0, True
    for a, b in zip(got, ref):
        assert rel(a, b) < 2e-5


@pytest.mark.parametrize("mode", ["chain64", "layers"])
def test_nerf_paths_agree(dev, nets, mode):
    """the fused background-NeRF chain (32-point tiles) against its 64-point-tile variant and the per-layer GEMM
    path: values and all 24 parameter gradients, per-ray view directions (x_div = samples per ray)."""
    from neuraludf_amd import mlp
    mods, _ = nets
    net = mods["nerf"]
    g = torch.Generator().manual_seed(9)
    S, R = 6, 83
    P = S * R
    p3 = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    pts4 = torch.cat([p3, torch.rand(P, 1, generator=g)], -1).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    w1, w2 = torch.randn(P, 1, generator=g).to(dev), torch.randn(P, 3, generator=g).to(dev)

    def run():
        net.zero_grad()
        s, rgb = net.evaluate(pts4, dirs, S)
        ((s * w1).sum() + (rgb * w2).sum()).backward()
        return [s.detach(), rgb.detach()] + [p.grad.clone() for p in net.parameters()]
    ref = run()
    try:
        if mode == "chain64":
            mlp.CHAIN_TILE = 64
        else:
            mlp.USE_CHAIN = False
        got = run()
    finally:
        mlp.CHAIN_TILE, mlp.USE_CHAIN = 0, True
    for a, b in zip(got, ref):
        assert rel(a, b) < 2e-5


def test_color_network(dev, nets):
    mods, sds = nets
    g = torch.Generator().manual_seed(4)
    P = 515
    pts = torch.randn(P, 3, generator=g) * 0.6
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    feat = torch.randn(P, 256, generator=g)
    w = [torch.randn(P, k, generator=g) for k in (3, 3, 10)]
    on = oracle_nets(sds, requires_grad=True)
    fr = feat.clone().requires_grad_(True)
    cb, col, lg = O.color_forward(on.color, pts, None, dirs, fr)
    ((cb * w[0]).sum() + (col * w[1]).sum() + (lg * w[2]).sum()).backward()

    net = mods["color"]
    net.zero_grad()
    fd = feat.to(dev).requires_grad_(True)
    cb2, col2, lg2 = net(pts.to(dev), None, dirs.to(dev), fd)
    assert rel(cb2, cb) < VTOL and rel(col2, col) < VTOL and rel(lg2, lg) < VTOL
    ((cb2 * w[0].to(dev)).sum() + (col2 * w[1].to(dev)).sum() + (lg2 * w[2].to(dev)).sum()).backward()
    assert grel(fd.grad, fr.grad) < GTOL
    for n, p in net.named_parameters():
        assert grel(p.grad, on.color[n].grad) < GTOL, n


def test_nerf(dev, nets):
    mods, sds = nets
    g = torch.Generator().manual_seed(5)
    P = 300
    p3 = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    inv = torch.rand(P, 1, generator=g)
    pts4 = torch.cat([p3, inv], -1)
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    w1, w2 = torch.randn(P, 1, generator=g), torch.randn(P, 3, generator=g)
    on = oracle_nets(sds, requires_grad=True)
    s, rgb = O.nerf_forward(on.nerf, pts4, dirs)
    ((s * w1).sum() + (rgb * w2).sum()).backward()
    net = mods["nerf"]
    net.zero_grad()
    s2, rgb2 = net(pts4.to(dev), dirs.to(dev))
    assert rel(s2, s) < VTOL and rel(rgb2, rgb) < VTOL
    ((s2 * w1.to(dev)).sum() + (rgb2 * w2.to(dev)).sum()).backward()
    for n, p in net.named_parameters():
        assert grel(p.grad, on.nerf[n].grad) < GTOL, n
    # density-only call (fields.py:614-617: input_views=None returns alpha) and its gradients into the pts / alpha layers
    net.zero_grad()
    s3 = net(pts4.to(dev), None)
    assert torch.equal(s3, s2.detach())
    (s3 * w1.to(dev)).sum().backward()
    for o in on.nerf.values():
        o.grad = None
    (O.nerf_forward(on.nerf, pts4, dirs)[0] * w1).sum().backward()
    for n, p in net.named_parameters():
        if n.startswith("pts_linears") or n.startswith("alpha_linear"):
            assert grel(p.grad, on.nerf[n].grad) < GTOL, n
    # use_viewdirs=False: constructible (output_linear in the state dict), forward asserts like the reference (:629-630)
    from neuraludf_amd.models import fields
    nv = fields.NeRF(D=2, W=64, d_in=4, d_in_view=3, multires=2, multires_view=2, use_viewdirs=False).to(dev)
    assert "output_linear.weight" in nv.state_dict()
    with pytest.raises(AssertionError):
        nv(pts4.to(dev), dirs.to(dev))


# ---------------------------------------------------------------------------------------------
def _rays(n, seed=7):
    from neuraludf_amd import synth
    scene = synth.make_scene("tiny")
    return synth.make_rays(scene, 0, n, seed=seed)


@pytest.mark.parametrize("case", ["plain", "bg_anneal", "normgrad", "theorical", "theorical_bg_anneal"])
def test_composite_stagewise(dev, case):
    """identical (z, udf, grad, colours) into the oracle's composite math and into the kernel."""
    from neuraludf_amd.models.udf_renderer_blending import _CompositeFn
    g = torch.Generator().manual_seed(11)
    kind = "theorical" if case.startswith("theorical") else "numerical"      # sdf2alpha_type (:308 / :321)
    case = {"theorical": "plain", "theorical_bg_anneal": "bg_anneal"}.get(case, case)
    N, S = 37, (70 if case != "bg_anneal" else 130)
    n_out = 9 if case == "bg_anneal" else 0
    r = _rays(N)
    z = torch.sort(r["near"] + (r["far"] - r["near"]) * torch.rand(N, S, generator=g), -1)[0]
    udf = (torch.rand(N, S, generator=g) * 0.2) ** 2
    udf[:, S // 2] = 1e-4                      # a surface hit
    grad = torch.randn(N, S, 3, generator=g) * 0.8
    col = torch.rand(N, S, 3, generator=g)
    cb = torch.rand(N, S, 3, generator=g)
    inv_s, beta, gamma = torch.tensor([30.0]), torch.tensor([60.0]), torch.tensor([20.0])
    sdist = 2.0 / 64
    anneal = 0.6 if case == "bg_anneal" else None
    fs = 0.9
    use_norm = case == "normgrad"
    bg_z = bg_sigma = bg_col = None
    if n_out:
        bg_z = torch.sort(r["far"] + 0.1 + torch.rand(N, n_out, generator=g) * 3, -1)[0]
        bg_sigma = torch.randn(N, n_out, generator=g) * 2
        bg_col = torch.rand(N, n_out, 3, generator=g)

    leaves = [t.clone().requires_grad_(True) for t in (udf, grad, col, cb, inv_s, beta, gamma)]
    if n_out:
        leaves += [bg_sigma.clone().requires_grad_(True), bg_col.clone().requires_grad_(True)]

    def oracle_composite(u, gr, c, b, s_, be, ga, bs=None, bc=None):
        dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((N, 1), sdist)], -1)
        mid = z + dists * 0.5
        dirs = r["rays_d"][:, None, :].expand(N, S, 3)
        pts = r["rays_o"][:, None, :] + dirs * mid[..., None]
        gm = torch.linalg.norm(gr, ord=2, dim=-1, keepdim=True)
        gn = gr / (gm + 1e-5)
        tc = (dirs * (gn if use_norm else gr)).sum(-1)
        flip = -torch.sign((dirs * gn).sum(-1))
        flip[flip == 0] = 1
        raw = O.udf2logistic(u, be, 1.0, 1.0)
        aocc = 1.0 - torch.exp(-torch.relu(raw) * ga * dists)
        vm = torch.cat([(tc < 0.01).float()[:, 1:], torch.ones(N, 1)], -1)
        vis = O.excl_cumprod((1.0 - aocc + fs * vm).clip(0, 1) + 1e-7).clip(0, 1)
        ap = O.sdf2alpha(u, -tc.abs(), dists, s_, anneal, kind=kind)
        am = O.sdf2alpha(-u, -tc.abs(), dists, s_, anneal, kind=kind)
        alpha = ap * vis + am * (1 - vis)
        cc, bb = c, b
        if bs is not None:
            dz = torch.cat([bg_z[:, 1:] - bg_z[:, :-1], torch.full((N, 1), sdist)], -1)
            alpha = torch.cat([alpha, 1.0 - torch.exp(-torch.relu(bs) * dz)], -1)
            cc = torch.cat([c, bc], 1)
            bb = torch.cat([b, bc], 1)
        w = alpha * O.excl_cumprod(1.0 - alpha + 1e-7)
        pn = torch.linalg.norm(pts, ord=2, dim=-1)
        ge = (gm[..., 0] - 1.0) ** 2
        relax, near = (pn < 1.2).float(), (u < 0.05).float().detach()
        sums = torch.stack([(relax * ge).sum(), relax.sum(), (near * ge).sum(), near.sum(),
                            torch.exp(-25000.0 * u).sum()])
        return dict(color=(cc * w[..., None]).sum(1), color_base=(bb * w[..., None]).sum(1), weights=w,
                    depth=(mid * w[:, :S]).sum(1, keepdim=True), normals=(flip[..., None] * gr * w[:, :S, None]).sum(1),
                    sums=sums, vis=vis, alpha=alpha[:, :S], wsum=w[:, :S].sum(-1, keepdim=True))

    ref = oracle_composite(*leaves)
    wts = dict(color=torch.randn(N, 3, generator=g), color_base=torch.randn(N, 3, generator=g),
               weights=torch.randn(N, S + n_out, generator=g) * 0.1, depth=torch.randn(N, 1, generator=g),
               normals=torch.randn(N, 3, generator=g), wsum=torch.randn(N, 1, generator=g))
    sw = torch.tensor([0.7, 0.0, 0.3, 0.0, 1e-3])

    def loss(o, to=lambda t: t):
        return sum((o[k] * to(wts[k])).sum() for k in wts) + (o["sums"] * to(sw)).sum()

    loss(ref).backward()

    D = lambda t: None if t is None else t.to(dev)
    dl = [t.detach().clone().to(dev).requires_grad_(True) for t in leaves]
    scal = torch.cat([dl[4], dl[5], dl[6]])
    c = dict(s_nominal=S, cos_anneal=anneal, flip_saturation=fs, use_norm_grad=use_norm, sparse_scale=25000.0,
             diagnostics=True, alpha_type=1 if kind == "theorical" else 0)
    outs = _CompositeFn.apply(c, D(r["rays_o"]), D(r["rays_d"]), D(z), torch.tensor([sdist], device=dev), None,
                              dl[0], dl[1], dl[2], dl[3], D(bg_z), dl[7] if n_out else None, dl[8] if n_out else None,
                              scal)
    names = ["color", "color_base", "weights", "depth", "normals", "wsum", "wsum_all", "sums"]
    o = dict(zip(names, outs[:8]))
    diag = dict(zip(["alpha", "alpha_plus", "alpha_minus", "vis_prob"], outs[8:12]))
    for k in ["color", "color_base", "weights", "depth", "normals", "wsum", "sums"]:
        assert rel(o[k], ref[k]) < VTOL, k
    assert rel(diag["vis_prob"], ref["vis"]) < VTOL
    assert rel(diag["alpha"], ref["alpha"]) < VTOL
    loss(o, lambda t: t.to(dev)).backward()
    for i, nm in enumerate(["udf", "grad", "color", "color_base", "inv_s", "beta", "gamma", "bg_sigma", "bg_color"][:len(dl)]):
        assert dl[i].grad is not None, nm
        assert grel(dl[i].grad, leaves[i].grad) < GTOL, nm


@pytest.mark.parametrize("kind", ["unbias", "noocc", "unbias_theorical"])
def test_upsample_and_merge_stagewise(dev, nets, kind):
    """each up-sampling round of the oracle (given its z, udf) against nudf_upsample / nudf_merge."""
    mods, sds = nets
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    r = _rays(53, seed=21)
    a_kind = "theorical" if kind.endswith("theorical") else "numerical"
    kind = kind.split("_")[0]
    cfg = O.RenderCfg(n_samples=64, n_importance=60 if kind == "unbias" else 66, n_outside=0, up_sample_steps=5,
                      upsampling_type="classical" if kind == "unbias" else "mix", sdf2alpha_type=a_kind)
    trace = []
    on = oracle_nets(sds)
    z0, _, sd = O.coarse_z(cfg, r["near"], r["far"], 53)
    O.importance_sample(on, cfg, r["rays_o"], r["rays_d"], z0, sd, trace)
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], n_samples=64,
                               n_importance=cfg.n_importance, n_outside=0, up_sample_steps=5, perturb=0.0,
                               upsampling_type=cfg.upsampling_type, sdf2alpha_type=a_kind)
    ro, rd = r["rays_o"].to(dev), r["rays_d"].to(dev)
    sdd = torch.tensor([sd], device=dev)
    n_bad = 0
    for t in trace:
        k = t["z_new"].shape[1]
        mode = 0 if t["kind"] == "unbias" else 1
        z_new, pts_new = rend._upsample(ro, rd, t["z"].to(dev), t["udf"].to(dev), sdd, k, mode, t["inv_s"], t["beta"],
                                        t["gamma"])
        err = (z_new.cpu() - t["z_new"]).abs().max(dim=1)[0]
        n_bad += int((err > 1e-4).sum())
        assert float(err.median()) < 1e-5
        assert rel(pts_new.reshape(53, k, 3), r["rays_o"][:, None] + r["rays_d"][:, None] * z_new.cpu()[..., None]) < 1e-5
        # merge against torch.sort on the oracle's own z_new
        u_new = torch.rand(53, k)
        zs, us = O.merge_sorted(t["z"], t["z_new"], t["udf"], u_new)
        zo, uo = rend._merge(t["z"].to(dev), t["udf"].to(dev), t["z_new"].to(dev), u_new.to(dev))
        assert torch.equal(zo.cpu(), zs)
        # torch.sort is not stable: compare the (z, udf) pairs as multisets per ray
        def canon(zz, uu):
            i1 = torch.argsort(uu, dim=1, stable=True)
            z1, u1 = torch.gather(zz, 1, i1), torch.gather(uu, 1, i1)
            return torch.gather(u1, 1, torch.argsort(z1, dim=1, stable=True))
        assert rel(canon(zo.cpu(), uo.cpu()), canon(zs, us)) < 1e-6
    # quantile bins may flip on ~ulp-level CDF differences (hard part 2)
    # ('theorical' takes 1 - sigmoid(sdf inv_s) with inv_s up to 1024: where the sigmoid is within a few ulp of 1 the
    # difference is an ulp-quantised 1e-7-ish number, so one ulp of the sigmoid moves those small weights by tens of
    # percent -- in the fp32 reference just the same -- and a few more bins flip)
    print(f"upsample stagewise [{kind}, {a_kind}]: {n_bad} of {len(trace) * 53} ray-rounds moved by > 1e-4")
    # bound: 1 % of the ray-rounds (round 2 allowed 5 %, round 3 2.5 %: with the reference's transcendentals -- Sleef
    # sigmoid, correctly rounded exp -- on top of the serial double scans the kernel reproduces the CPU reference's bins;
    # 'theorical' 2.5 %)
    assert n_bad <= max(1, len(trace) * 53 // (40 if a_kind == "theorical" else 100)), n_bad


@pytest.mark.parametrize("case", ["cfg1_flat", "classical_bg", "mix", "theorical_bg", "square_bg", "idr_bg"])
def test_render_end_to_end_and_param_grads(dev, nets, case):
    mods, sds = nets
    if case == "idr_bg":             # colour network with the (detached) unit normals in its input (fields.py:456-461)
        from neuraludf_amd.models import fields
        mods = perturb_(build_modules(fields, seed=0, color_mode="idr"))
        sds = state_dicts(mods)
        for m in mods.values():
            m.to(dev)
    if case == "square_bg":          # udf_type 'square': same weights, another head (own modules: the engine reads the type)
        from neuraludf_amd.models import fields
        mods = perturb_(build_modules(fields, seed=0, udf_type="square"))
        for m in mods.values():
            m.to(dev)
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    n = 64
    r = _rays(n, seed=31)
    if case == "cfg1_flat":          # BASELINE config 1: 64 rays x 32 samples, no importance / outside
        kw = dict(n_samples=32, n_importance=0, n_outside=0, up_sample_steps=1)
    elif case == "classical_bg":
        kw = dict(n_samples=64, n_importance=50, n_outside=32, up_sample_steps=5)
    elif case == "theorical_bg":     # the reference's other sdf2alpha branch, core and up-sampling
        kw = dict(n_samples=64, n_importance=50, n_outside=32, up_sample_steps=5, sdf2alpha_type="theorical")
    elif case in ("square_bg", "idr_bg"):
        kw = dict(n_samples=64, n_importance=50, n_outside=32, up_sample_steps=5)
    else:
        kw = dict(n_samples=64, n_importance=78, n_outside=0, up_sample_steps=5, upsampling_type="mix",
                  use_norm_grad_for_cosine=True)
    cfg = O.RenderCfg(**{k: v for k, v in kw.items()})
    if case == "square_bg":
        cfg.udf = O.UDFCfg(udf_type="square")
    if case == "idr_bg":
        cfg.color = O.ColorCfg(mode="idr", d_in=12)
    on = oracle_nets(sds, requires_grad=True)
    ref = O.render(on, cfg, r["rays_o"], r["rays_d"], r["near"], r["far"], cos_anneal_ratio=0.8, flip_saturation=0.9)
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], perturb=1.0, **kw)
    for m in mods.values():
        m.zero_grad()
    D = lambda t: t.to(dev)
    out = rend.render(D(r["rays_o"]), D(r["rays_d"]), D(r["near"]), D(r["far"]), cos_anneal_ratio=0.8,
                      perturb_overwrite=0, flip_saturation=0.9)
    assert set(["color_base", "color", "color_pixel", "patch_colors", "patch_mask", "weight_sum", "weight_sum_fg_bg",
                "depth", "variance", "beta", "gamma", "normals", "gradients", "gradients_flip", "weights",
                "gradient_error", "gradient_error_near_surface", "inside_sphere", "udf", "z_vals", "gradient_mag",
                "true_cos", "vis_prob", "alpha", "alpha_plus", "alpha_minus", "mid_z_vals", "dists", "sparse_error",
                "alpha_occ", "raw_occ", "sparse_random_error"]) == set(out.keys())
    # samples: identical up to quantile-bin flips on a few rays
    zerr = (out["z_vals"].cpu() - ref["z_vals"]).abs().max(dim=1)[0]
    good = zerr < 1e-4
    # up-sampling is discontinuous (searchsorted bins, thresholds): ulp-level differences in the MLP
    # output move the new samples of some rays by a bin (the reference itself moves ~9 % of the rays
    # under 1e-6 relative noise); those rays are compared statistically (PSNR)
    # ('theorical' weights come from 1 - sigmoid(x) near sigmoid = 1, see test_upsample_and_merge_stagewise: more rays
    # change a bin in one of the five rounds; everything downstream is still checked exactly on the oracle's own samples)
    # ('square': udf = h0^2 is flat around the surface, so the sharp late rounds see relative ulp noise of h0 doubled)
    print(f"render end to end [{case}]: {int(good.sum())} / {n} rays keep the oracle's samples")
    # measured (round 5): 64 / 62 / 64 / 62 of 64 rays for the default settings, 46 ('theorical'), 51 ('square')
    assert good.float().mean() > {"theorical_bg": 0.55, "square_bg": 0.6}.get(case, 0.85)
    for k in ["color", "color_base", "depth", "weight_sum"]:
        assert rel(out[k][good.to(dev)], ref[k][good]) < 1e-4, k
    mse = ((out["color"].cpu() - ref["color"]) ** 2).mean()
    psnr = 20.0 * math.log10(1.0 / math.sqrt(float(mse) + 1e-20))
    assert psnr > 70.0, psnr
    # ---- same samples in both: everything downstream of the sampling must agree, incl. d loss / d theta ----
    z_ref = ref["z_vals"]
    _, z_out_ref, sd = O.coarse_z(cfg, r["near"], r["far"], n)
    sdd = torch.tensor([sd], device=dev)
    bg_sigma = bg_color = z_out_d = None
    if cfg.n_outside:
        z_out_d = D(z_out_ref.contiguous())
        bg_sigma, bg_color, _ = rend.render_core_outside(D(r["rays_o"]), D(r["rays_d"]), z_out_d, sdd)
    out2 = rend.render_core(D(r["rays_o"]), D(r["rays_d"]), D(z_ref), sdd, 0.8, None, z_out_d, bg_sigma, bg_color,
                            0.9, s_nominal=cfg.n_samples + cfg.n_importance)
    for k in ["color", "color_base", "depth", "weights", "udf", "gradients", "normals", "vis_prob", "alpha",
              "gradient_error", "gradient_error_near_surface", "sparse_error", "true_cos", "weight_sum",
              "weight_sum_fg_bg", "alpha_occ", "inside_sphere", "mid_z_vals", "dists", "gradient_mag"]:
        # sparse_error = mean sum exp(-25000 udf) amplifies an fp32 ulp of udf (1e-8) to 2.5e-4 relative
        assert rel(out2[k], ref[k]) < (2e-3 if k == "sparse_error" else VTOL), k

    def loss_of(o, rgb):
        return ((o["color"] - rgb).abs().mean() + 0.5 * (o["color_base"] - rgb).abs().mean()
                + 0.1 * o["gradient_error"] + 0.01 * o["gradient_error_near_surface"])
        # sparse_error = mean sum exp(-25000 udf) is left out of THIS loss on purpose: its gradient
        # -25000 exp(-25000 u) turns an fp32 ulp-level difference of u (1e-6 abs) into percents, for the
        # fp32 reference just as much; its backward is checked on identical udf inputs in
        # test_composite_stagewise instead

    loss_of(ref, r["true_rgb"]).backward()
    loss_of(out2, D(r["true_rgb"])).backward()

    # fp64 run of the oracle on the same samples = ground truth for the gradients; the fp32 oracle's own
    # distance from it calibrates the tolerance (these are second-order quantities with cancellation)
    torch.set_default_dtype(torch.float64)
    try:
        on64 = oracle_nets(sds, requires_grad=True, dtype=torch.float64)
        ro, rd = r["rays_o"].double(), r["rays_d"].double()
        ba = bc = None
        if cfg.n_outside:
            ba, bc = O.render_core_outside(on64, cfg, ro, rd, torch.cat([z_ref.double(), z_out_ref.double()], -1), sd)
        o64 = O.render_core(on64, cfg, ro, rd, z_ref.double(), sd, 0.8, None, ba, bc, 0.9)
        loss_of(o64, r["true_rgb"].double()).backward()
    finally:
        torch.set_default_dtype(torch.float32)
    for net, key in [("udf", "udf"), ("color", "color"), ("var", "var"), ("beta", "beta"), ("nerf", "nerf")]:
        for nme, p in mods[net].named_parameters():
            g32, g64 = getattr(on, key)[nme].grad, getattr(on64, key)[nme].grad
            if g32 is None or g64 is None or not p.requires_grad:
                continue
            assert p.grad is not None, (net, nme)
            e_oracle32 = grel(g32, g64.float())
            if net in ("color", "nerf"):
                # ReLU networks on 64 rays (512 NeRF points): ONE unit whose pre-activation sits within fp32 rounding of zero
                # takes the other branch than in the oracle and moves a whole row of a weight gradient by that point's share,
                # 1 / 512 -- measured 1.4e-3 in the max norm on nerf.pts_linears.1.weight with the exact-fp32 kernels as with
                # the split ones.  Held in the 2-norm at the bar, in the max norm at 5x; the full-size fixtures
                # (16 384 NeRF points, test_gpu_fullsize_parity.py) hold the max norm at 1e-3.
                assert grel2(p.grad, g64.float()) < max(GTOL, 3.0 * e_oracle32), (net, nme, e_oracle32)
                assert grel(p.grad, g64.float()) < max(5 * GTOL, 3.0 * e_oracle32), (net, nme, e_oracle32)
            else:
                assert grel(p.grad, g64.float()) < max(GTOL, 3.0 * e_oracle32), (net, nme, e_oracle32)


@pytest.mark.parametrize("with_mask", [False, True])
def test_color_loss_two_term_kernel(dev, with_mask):
    """ColorLoss with only the L1 terms active (one fused launch) against the oracle's loss/loss.py restatement."""
    from neuraludf_amd.loss.loss import ColorLoss
    g = torch.Generator().manual_seed(17)
    n = 333
    cb, c, gt = (torch.rand(n, 3, generator=g) for _ in range(3))
    mask = (torch.rand(n, 1, generator=g) > 0.4).float() if with_mask else None
    cbr, cr = cb.clone().requires_grad_(True), c.clone().requires_grad_(True)
    ref = O.color_loss(0.2, 1.0, 0.5, 0.0, 3, cbr, cr, gt, None, mask, None, None, None)
    (ref["loss"] + 0.3 * ref["color_base_loss"] - 0.1 * ref["color_loss"]).backward()
    cbd, cd = cb.to(dev).requires_grad_(True), c.to(dev).requires_grad_(True)
    out = ColorLoss(0.2, 1.0, 0.5, 0.0)(cbd, cd, gt.to(dev), None, mask.to(dev) if with_mask else None, None, None, None)
    for k in ("loss", "color_base_loss", "color_loss"):
        assert abs(float(out[k].detach()) - float(ref[k].detach())) < 1e-6 * max(1.0, abs(float(ref[k].detach()))), k
    (out["loss"] + 0.3 * out["color_base_loss"] - 0.1 * out["color_loss"]).backward()
    assert grel(cbd.grad, cbr.grad) < 1e-5 and grel(cd.grad, cr.grad) < 1e-5


@pytest.mark.parametrize("with_mask", [False, True])
def test_color_loss_sharded_split_matches_fused(dev, with_mask):
    """the ray-sharded form (local sums -> all-reduce -> finish) of the fused ColorLoss: one shard reproduces the fused
    launch bit for bit, two shards summed (what the all-reduce does) agree to fp32 rounding."""
    from neuraludf_amd._lib import call, ptr
    g = torch.Generator().manual_seed(23)
    n = 512
    cb, c, gt = (torch.rand(n, 3, generator=g).to(dev) for _ in range(3))
    mask = (torch.rand(n, 1, generator=g) > 0.4).float().to(dev) if with_mask else None
    w = (0.2, 1.0, 0.5)

    def fused():
        out, den = torch.empty(3, device=dev), torch.empty(1, device=dev)
        call("nudf_color_loss_fwd", ptr(cb), ptr(c), ptr(gt), cb.numel(), ptr(mask), mask.numel() if with_mask else 0, *w,
             None, ptr(out), ptr(den))
        return out, den

    def fused_wdev():                   # the same weights from a device vector (NUDF_LW_*) override garbage by-value ones
        from neuraludf_amd._lib import LW_COUNT
        out, den = torch.empty(3, device=dev), torch.empty(1, device=dev)
        wd = torch.zeros(LW_COUNT, device=dev)
        wd[:3] = torch.tensor(w)
        call("nudf_color_loss_fwd", ptr(cb), ptr(c), ptr(gt), cb.numel(), ptr(mask), mask.numel() if with_mask else 0,
             7.0, 8.0, 9.0, ptr(wd), ptr(out), ptr(den))
        return out, den

    def sums_of(lo, hi):
        s = torch.empty(3, device=dev)
        m = mask[lo:hi].contiguous() if with_mask else None
        call("nudf_color_loss_sums", ptr(cb[lo:hi].contiguous()), ptr(c[lo:hi].contiguous()), ptr(gt[lo:hi].contiguous()),
             (hi - lo) * 3, ptr(m), (hi - lo) if with_mask else 0, ptr(s))
        return s

    def finish(s):
        out, den = torch.empty(3, device=dev), torch.empty(1, device=dev)
        call("nudf_color_loss_finish", ptr(s), 1 if with_mask else 0, *w, None, ptr(out), ptr(den))
        return out, den
    o0, d0 = fused()
    ow, dw = fused_wdev()
    assert torch.equal(o0, ow) and torch.equal(d0, dw)
    o1, d1 = finish(sums_of(0, n))
    assert torch.equal(o0, o1) and torch.equal(d0, d1)
    o2, d2 = finish(sums_of(0, n // 2) + sums_of(n // 2, n))
    assert rel(o2, o0) < 1e-6 and rel(d2, d0) < 1e-6


@pytest.mark.parametrize("case", [dict(mode="idr", d_in=12, multires_view=4, squeeze_out=True, blending_cand_views=0),
                                  dict(mode="no_normal", d_in=6, multires_view=4, squeeze_out=True, blending_cand_views=10),
                                  dict(mode="no_view_dir", d_in=9, multires_view=0, squeeze_out=False, blending_cand_views=0)],
                         ids=["idr", "no_normal", "no_view_dir"])
def test_plain_rendering_network(dev, case):
    """RenderingNetwork (fields.py:325-397, the non-residual colour MLP; importable name of the call surface) on the
    per-layer HIP GEMMs against the oracle: values, d feature, parameter gradients, all three modes."""
    from neuraludf_amd.models import fields as nf
    # seed 6: with seed 5 one of the 96 k hidden pre-activations sits within fp32 rounding of the ReLU kink, so the two
    # implementations take different sides there and one unit's whole adjoint differs (scripts/debug_plain.py)
    torch.manual_seed(6)
    net = nf.RenderingNetwork(d_feature=256, d_out=3, d_hidden=96, n_layers=3, weight_norm=True, **case)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    P = 333
    pts, nrm, dirs = (torch.randn(P, 3, generator=g) for _ in range(3))
    feat = torch.randn(P, 256, generator=g)
    fr = feat.clone().requires_grad_(True)
    color, extra = O.rendering_forward(sd, pts, nrm, dirs, fr, mode=case["mode"], multires_view=case["multires_view"],
                                       squeeze_out=case["squeeze_out"])
    w1, w2 = torch.randn(P, 3, generator=g), torch.randn(P, max(extra.shape[1], 1), generator=g)
    loss = (color * w1).sum() + ((extra * w2).sum() if extra.shape[1] else 0.0)
    loss.backward()
    net.to(dev)
    fd = feat.to(dev).requires_grad_(True)
    out = net(pts.to(dev), nrm.to(dev), dirs.to(dev), fd)
    if case["blending_cand_views"] > 0:
        c2, e2 = out
        assert rel(e2, extra) < VTOL
        l2 = (c2 * w1.to(dev)).sum() + (e2 * w2.to(dev)).sum()
    else:
        c2 = out
        l2 = (c2 * w1.to(dev)).sum()
    assert rel(c2, color) < VTOL
    l2.backward()
    assert grel(fd.grad, fr.grad) < GTOL
    for n, p in net.named_parameters():
        assert grel(p.grad, sd[n].grad) < GTOL, n
