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


def _udf_backward(dev, P, seed=3):
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
