"""the weight-gradient GEMM group under the round-5 workgroup order (tn_decode_block: every row chunk's tiles on one XCD,
holes in the grid, per-group lane rotation): random groups -- ragged tile counts, row counts that leave a narrower last
column of chunks, single-tile problems between multi-tile ones -- against a float64 contraction, in the exact fp32 mode and
the bf16x3 mode, accumulate and assign forms; and the holes leave the workspace / outputs of other problems untouched."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _jobs(shapes, M, dev, g):
    jobs, ref = [], []
    for NA, NB in shapes:
        lda, ldb = max(4, (NA + 3) // 4 * 4), max(4, (NB + 3) // 4 * 4)
        A = torch.randn(M, lda, generator=g).to(dev)
        B = (torch.randn(M, ldb, generator=g) * 0.1).to(dev)
        C0 = torch.randn((NA + 31) // 32 * 32, NB, generator=g).to(dev)
        d0 = torch.randn((NA + 31) // 32 * 32, generator=g).to(dev)
        jobs.append((A, NA, B, NB, C0.clone(), d0.clone()))
        ref.append((C0, d0, A[:, :NA].double().t() @ B[:, :NB].double(), A[:, :NA].double().sum(0)))
    return jobs, ref


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("M", [777, 5000, 40000, 65536])
def test_random_groups_against_float64(precision, M):
    from neuraludf_amd import mlp
    dev = torch.device("cuda:0")
    old = mlp.PRECISION
    mlp.set_precision(precision)
    try:
        gen = torch.Generator().manual_seed(M)
        cases = [[(256, 40), (256, 256), (217, 256), (1, 256)],
                 [(129, 72), (3, 128), (300, 260), (64, 64), (256, 129)],
                 [(384, 384)],
                 [(128, 128), (256, 256), (128, 128), (257, 31)]]
        for shapes in cases:
            for assign in (False, True):
                jobs, ref = _jobs(shapes, M, dev, gen)
                mlp.gemm_tn_grouped(jobs, M, assign=assign)
                torch.cuda.synchronize()
                for (A, NA, B, NB, C, db), (C0, d0, Cr, dr) in zip(jobs, ref):
                    scale = float(Cr.abs().max())
                    want = Cr + (0.0 if assign else C0[:NA].double())
                    tol = 3e-6 * scale * max(1.0, (M / 4096.0) ** 0.5) + 1e-6
                    assert float((C[:NA].double() - want).abs().max()) <= tol, (precision, M, shapes, NA, NB, assign)
                    assert torch.equal(C[NA:], C0[NA:])                      # padding rows are not the kernel's to write
                    wantb = dr + (0.0 if assign else d0[:NA].double())
                    assert float((db[:NA].double() - wantb).abs().max()) <= 3e-6 * float(dr.abs().max() + 1.0) * max(1.0, (M / 4096.0) ** 0.5)
                    assert torch.equal(db[NA:], d0[NA:])
    finally:
        mlp.set_precision(old)


def test_launch_is_deterministic_and_independent_of_the_xcd_order():
    """the same group twice, and with the plain tile-major order (NUDF_TN_FLAGS bit 64: no holes): identical bits -- the order of
    the workgroups changes where they run, not what they compute (workspace slots are tile-major, the reduce order is fixed)."""
    from neuraludf_amd import _lib, mlp
    dev = torch.device("cuda:0")
    old = mlp.PRECISION
    mlp.set_precision("bf16x3")
    try:
        M = 30000
        shapes = [(256, 40), (256, 256), (217, 256), (1, 256), (256, 256)]
        outs = []
        for flags in (0, 0, 64):
            prev = _lib.lib().nudf_set_tn_flags(flags)
            try:
                jobs, _ = _jobs(shapes, M, dev, torch.Generator().manual_seed(5))
                mlp.gemm_tn_grouped(jobs, M, assign=True, rows_per_block=1024)   # 30 chunks per tile: a last column of 6 -> holes
                torch.cuda.synchronize()
                outs.append([(j[4].clone(), j[5].clone()) for j in jobs])
            finally:
                _lib.lib().nudf_set_tn_flags(prev)
        for other in outs[1:]:
            for (c0, d0), (c1, d1) in zip(outs[0], other):
                assert torch.equal(c0, c1) and torch.equal(d0, d1)
    finally:
        mlp.set_precision(old)


def test_operands_past_2_gib_take_the_flat_path_for_the_early_chunks():
    """the steady state of the bf16x3 kernel reads through BUFFER descriptors (32-bit byte counts): a row chunk whose operand
    tail is >= 2 GiB must stay on the flat-address staging (same values), later chunks of the same launch use the descriptors."""
    from neuraludf_amd import mlp
    dev = torch.device("cuda:0")
    old = mlp.PRECISION
    mlp.set_precision("bf16x3")
    try:
        M, N = 2_200_000, 256                      # 2 200 000 x 256 x 4 B = 2.25 GB per operand
        g = torch.Generator(device=dev).manual_seed(3)
        A = torch.randn(M, N, device=dev, generator=g)
        B = torch.randn(M, N, device=dev, generator=g) * 0.05
        C = torch.zeros(N, N, device=dev)
        db = torch.zeros(N, device=dev)
        mlp.gemm_tn_grouped([(A, N, B, N, C, db)], M, assign=True)
        torch.cuda.synchronize()
        ref = torch.zeros(N, N, device=dev, dtype=torch.float64)
        refb = torch.zeros(N, device=dev, dtype=torch.float64)
        for r0 in range(0, M, 200_000):
            a = A[r0:r0 + 200_000].double()
            ref += a.t() @ B[r0:r0 + 200_000].double()
            refb += a.sum(0)
        scale = float(ref.abs().max())
        assert float((C.double() - ref).abs().max()) <= 2e-5 * scale
        assert float((db.double() - refb).abs().max()) <= 2e-5 * float(refb.abs().max() + 1.0)
    finally:
        mlp.set_precision(old)
