"""The wave-private fused-chain kernel (csrc/mlp_chain_rows.hip, tile_rows = 128) against the workgroup-shared
kernel (csrc/mlp_chain.hip) -- the path every oracle test of test_gpu_kernels.py pins -- on the same inputs, for
every sweep of the three networks: point counts around the 32-point wave tiles and the 128-point workgroups (ragged
last wave, whole waves without points), and one size with two full rounds of workgroups.

The two kernels run the same products in the same k order; the wave-private one starts its accumulators from the
bias (bias-first), so values agree to fp32 rounding (2e-5 of the tensor's magnitude), not to the bit.  Gradients of
the ReLU networks can additionally move by whole elements where a pre-activation lies within an ulp of zero; those
tensors are held to a relative L2 bound (chain_sweeps.compare)."""
import pytest
import torch

from chain_sweeps import compare, sweeps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _exact_fp32_mode():
    """these are statements about the exact-fp32 MFMA kernels (mlp.PRECISION = "fp32": bit-identity between the three
    fp32 chain kernels, the blocked state layout, the paired tiles); the library default is the bf16x3 mode
    (tests/test_gpu_bf16x3.py).  conftest restores the mode after every test."""
    from neuraludf_amd import mlp
    mlp.set_precision("fp32")
    yield


@pytest.mark.parametrize("P", [1, 31, 33, 97, 128, 129, 1000, 4133])
def test_rows_kernel_matches_shared_kernel_ragged_sizes(dev, P):
    a = sweeps(dev, P, 64, seed=P)
    b = sweeps(dev, P, 128, seed=P)
    assert set(a) == set(b)
    bad, nbit, worst, worst_l2 = compare(a, b, f"P={P}", verbose=False, l2_tol=max(3e-3, 10.0 / P))
    assert not bad, bad
    for k in a:
        assert bool(torch.isfinite(b[k]).all()), k


def test_rows_kernel_two_rounds_and_rerun_is_deterministic(dev):
    P = 128 * 300 + 77
    a = sweeps(dev, P, 64, seed=3)
    b = sweeps(dev, P, 128, seed=3)
    c = sweeps(dev, P, 128, seed=3)
    bad, _, worst, worst_l2 = compare(a, b, f"P={P} shared vs rows", verbose=True)
    assert not bad, bad
    # the chains themselves are run-to-run identical (only the weight-gradient GEMMs use fp32 atomics)
    for k in ("udf", "feat", "X8", "g", "DA0", "uo", "cb", "cc", "dCIN", "nsig", "nrgb"):
        assert torch.equal(b[k], c[k]), k


def test_rows_kernel_is_the_one_that_ran(dev):
    """tile_rows = 128 must reach the wave-private kernel: its bias-first summation differs from the shared kernel's
    in the last bits, so bit-identical results would mean a silent fall-back."""
    a = sweeps(dev, 4096, 64, seed=9)
    b = sweeps(dev, 4096, 128, seed=9)
    assert not torch.equal(a["feat"], b["feat"])
    assert float((a["feat"] - b["feat"]).abs().max()) < 2e-5 * float(a["feat"].abs().max())


# ---- the workgroup-shared 64-point tile with the transposed product (tile_rows = 66, mlp_chain_tq_kernel) ---------------
@pytest.mark.parametrize("P", [1, 63, 65, 129, 1000, 4133])
def test_tq_kernel_matches_shared_kernel_ragged_sizes(dev, P):
    a = sweeps(dev, P, 64, seed=P)
    b = sweeps(dev, P, 66, seed=P)
    assert set(a) == set(b)
    bad, nbit, worst, worst_l2 = compare(a, b, f"P={P}", verbose=False, l2_tol=max(3e-3, 10.0 / P))
    assert not bad, bad
    for k in a:
        assert bool(torch.isfinite(b[k]).all()), k


def test_tq_kernel_full_size_is_the_one_that_ran_and_deterministic(dev):
    P = 64 * 700 + 13
    a = sweeps(dev, P, 64, seed=5)
    b = sweeps(dev, P, 66, seed=5)
    c = sweeps(dev, P, 66, seed=5)
    bad, _, worst, worst_l2 = compare(a, b, f"P={P} shared vs tq", verbose=True)
    assert not bad, bad
    # same products in the same order, bias added after them in both kernels: the UDF sweeps agree to the BIT (what makes
    # chunked and unchunked renders identical although their launches pick different kernels)
    for k in ("udf", "sign", "feat", "X4", "X8", "g", "DA0", "DA3", "DA7", "uo"):
        assert torch.equal(a[k], b[k]), k
    for k in ("udf", "feat", "X8", "g", "DA0", "uo", "cb", "cc", "dCIN", "nsig", "nrgb"):
        assert torch.equal(b[k], c[k]), k


def mlp_state_probe(dev, P):
    from chain_sweeps import engines
    x = (torch.rand(P, 3) * 2 - 1).to(dev)
    return engines(dev)["eng"].forward(x, need_grad_state=True)


def test_blocked_state_layout_matches_row_major_path(dev):
    """large fp32 launches keep the UDF engine's saved state in the BLOCKED layout (transposed-product kernel + grouped
    weight-gradient GEMM address it): every value, stored array (un-blocked for the comparison) and parameter gradient
    against the row-major path on the default kernel, at a size with a ragged last tile."""
    from neuraludf_amd import mlp
    P = 64 * 300 + 21
    assert mlp.BLOCKED_STATE and mlp._state_blocked(P)
    a = sweeps(dev, P, 64, seed=8)          # tile 64: row-major state, mlp_chain_kernel
    b = sweeps(dev, P, 0, seed=8)           # auto: blocked state
    bad, _, worst, worst_l2 = compare(a, b, f"P={P} row-major vs blocked", verbose=True)
    assert not bad, bad
    for k in ("udf", "sign", "feat", "X4", "X8", "g", "DA0", "DA3", "DA7", "uo"):     # bit-identical sweeps
        assert torch.equal(a[k], b[k]), k
    st = mlp_state_probe(dev, P)
    assert mlp._isblk(st["X"][4]) and not mlp._isblk(st["X"][0])      # the blocked path really ran


# ---- two 64-point tiles per workgroup in anti-phase (tile_rows = 130, mlp_chain_pair_kernel) -----------------------------
@pytest.mark.parametrize("P", [1, 63, 65, 129, 191, 1000, 4133])
def test_pair_kernel_equals_tq_kernel_to_the_bit_ragged_sizes(dev, P):
    """the paired-tile kernel runs the transposed-product kernel's own K loops and epilogues, only interleaved in time:
    every sweep output, stored array and parameter gradient of all three networks is bit-identical -- including sizes whose
    last workgroup holds one live tile and one dead one (P = 1, 63, 65: 1-2 tiles; 129, 191: 3 tiles)."""
    a = sweeps(dev, P, 66, seed=P)
    b = sweeps(dev, P, 130, seed=P)
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))


def test_pair_kernel_full_size_default_dispatch_and_determinism(dev):
    """at >= 32 768 points the automatic choice IS the paired kernel (blocked state for the UDF sweeps, row-major for the
    colour / NeRF chains and forward-only launches): bit-identical to the explicit pair request, to the transposed-product
    kernel and -- on the UDF sweeps -- to the default shared-tile kernel; re-runs are identical."""
    from neuraludf_amd import mlp
    P = 64 * 701 + 13                       # 702 tiles = 351 pairs, ragged last tile
    assert mlp._state_blocked(P)
    ref = sweeps(dev, P, 64, seed=11)       # row-major state, mlp_chain_kernel<64>
    tq = sweeps(dev, P, 66, seed=11)
    auto = sweeps(dev, P, 0, seed=11)
    pair = sweeps(dev, P, 130, seed=11)
    again = sweeps(dev, P, 0, seed=11)
    for k in auto:
        assert torch.equal(auto[k], pair[k]), k
        assert torch.equal(auto[k], again[k]), k
        assert torch.equal(auto[k], tq[k]), k
    for k in ("udf", "sign", "feat", "X4", "X8", "g", "DA0", "DA3", "DA7", "uo"):
        assert torch.equal(ref[k], auto[k]), k
    bad, _, worst, worst_l2 = compare(ref, auto, f"P={P} shared vs paired", verbose=True)
    assert not bad, bad
