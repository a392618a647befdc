"""BASELINE config 5 mode: 16-bit MFMA operands (fp16 forward sweeps, bf16 backward sweeps, fp32 accumulate and
fp32 everything else) against the exact-fp32 path on identical rays / weights.  Per SURVEY section 8(c) the bar for
this mode is PSNR of the colours (runner formula, exp_runner_blending.py:341-342, mask = 1), not 1e-4."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 20.0 * math.log10(1.0 / math.sqrt(max(mse, 1e-20)))


def _cos(a, b):
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    return float((a * b).sum() / (a.norm() * b.norm()).clamp(min=1e-30))


def test_mixed16_render_and_gradients_track_fp32():
    from neuraludf_amd import mlp, synth
    from neuraludf_amd.train import Trainer
    dev = torch.device("cuda:0")
    rconf = dict(n_samples=64, n_importance=0, n_outside=0, up_sample_steps=1, perturb=0.0)   # fixed samples:
    tr = Trainer(dev, rconf, seed=0)                                                           # no discontinuities
    rays = synth.make_rays(synth.make_scene("dtu"), 0, 256, seed=11)
    batch = {k: v.to(dev) for k, v in rays.items()}

    def run():
        for m in tr.modules().values():
            m.zero_grad()
        loss, out = tr.loss(batch, cos_anneal_ratio=1.0, flip_saturation=1.0, perturb_overwrite=0)
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in tr.udf.named_parameters()}
        grads.update({"c." + n: p.grad.detach().clone() for n, p in tr.color.named_parameters()})
        return float(loss), out["color"].detach().clone(), out["weights"].detach().clone(), grads

    base = mlp.PRECISION          # the library default (bf16x3), or whatever NUDF_PRECISION selects
    l32, c32, w32, g32 = run()
    try:
        mlp.set_precision("mixed16")
        l16, c16, w16, g16 = run()
    finally:
        mlp.set_precision(base)
    l32b, c32b, _, _ = run()                      # switching back restores the exact path
    assert torch.equal(c32, c32b) and l32 == l32b

    assert not torch.equal(c16, c32), "mixed16 must actually use the 16-bit kernels"
    psnr = _psnr(c16, c32)
    assert psnr > 55.0, psnr
    assert float((w16 - w32).abs().max()) < 2e-2
    assert abs(l16 - l32) < 5e-3 * max(1.0, abs(l32))
    worst = min(_cos(g16[k], g32[k]) for k in g32 if float(g32[k].abs().max()) > 0)
    assert worst > 0.98, worst
    print(f"mixed16 vs fp32: colour PSNR {psnr:.1f} dB, min gradient cosine {worst:.5f}")


@pytest.mark.parametrize("P", [1000, 64 * 37 + 1])
def test_mixed16_stored_state_is_4_point_packed(P):
    """the 16-bit mode's saved arrays (X, DA; R / EX / ABAR through the parameter gradients) are bf16 in the 4-point
    packed layout of include/nudf.h (NUDF_CH_STATE16): unpacked (mlp.unpack16) they are the fp32 path's arrays to 16-bit
    operand accuracy at every row -- a layout slip would move whole rows by O(1) -- ragged P (pad rows, a partial last
    quad for the weight-gradient GEMM) included; the parameter gradients of the three-sweep backward keep their direction."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from chain_sweeps import engines
    from neuraludf_amd import mlp
    dev = torch.device("cuda:0")
    st_ = engines(dev)
    eng, ceng = st_["eng"], st_["ceng"]
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    d_udf = torch.randn(P, generator=g).to(dev)
    d_g = torch.randn(P, 3, generator=g).to(dev)
    d_feat = (torch.randn(P, ceng.cin_ld, generator=g) * 0.1).to(dev)

    def run():
        st = eng.forward(x, need_grad_state=True, feat_ld=ceng.cin_ld)
        gr, DA = eng.gradient(x, st)
        grads = eng.backward(x, st, DA, d_udf, d_feat, ceng.cin_ld, d_g)
        return st, gr, DA, [t.clone() for t in grads]

    base = mlp.PRECISION          # the library default (bf16x3), or whatever NUDF_PRECISION selects
    st32, g32, DA32, p32 = run()
    try:
        mlp.set_precision("mixed16")
        st16, g16, DA16, p16 = run()
    finally:
        mlp.set_precision(base)
    L = len(DA32)
    for l in range(1, L + 1):
        a = st16["X"][l]
        assert a.dtype == torch.bfloat16 and mlp._isp4(a)
        w = eng.layers[l].inp
        ref = mlp.unblock(st32["X"][l])[:P, :w] if mlp._isblk(st32["X"][l]) else st32["X"][l][:P, :w]
        d = (mlp.unpack16(a)[:P, :w] - ref).abs()
        assert float(d.max()) < 2e-2 * (float(ref.abs().max()) + 1e-6), (l, float(d.max()), float(ref.abs().max()))
    for l in range(L):
        w = eng.layers[l].out
        ref = DA32[l][:P, :w]
        got = mlp.unpack16(DA16[l])[:P, :w]
        rel_l2 = float((got - ref).norm() / (ref.norm() + 1e-30))
        assert rel_l2 < 3e-2, (l, rel_l2)
        # per row: no row moved by O(1)
        row = (got - ref).norm(dim=1) / (ref.norm(dim=1) + 1e-3 * float(ref.norm(dim=1).max()))
        assert float(row.max()) < 0.25, (l, float(row.max()))
    assert float((g16 - g32).abs().max()) < 5e-2 * float(g32.abs().max())
    for a, b in zip(p16, p32):
        if float(b.abs().max()) > 0:
            assert _cos(a, b) > 0.98, _cos(a, b)


@pytest.mark.parametrize("P", [960, 64 * 41])
def test_mixed16_colour_net_state_is_packed_bf16(P):
    """the colour net's saved hidden activations and hidden adjoints are bf16 (4-point packed) in the 16-bit mode while
    its interface buffers (CIN, VIN, d VIN, d CIN, the head adjoints) stay fp32: unpacked hidden state = the fp32 path's to
    16-bit accuracy at every row, outputs within the mode's bar, parameter gradients and d CIN keep their direction."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from chain_sweeps import engines
    from neuraludf_amd import mlp
    dev = torch.device("cuda:0")
    st_ = engines(dev)
    eng, ceng = st_["eng"], st_["ceng"]
    S = 64
    g = torch.Generator().manual_seed(6)
    x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    rays_d = torch.nn.functional.normalize(torch.randn(P // S, 3, generator=g), dim=-1).to(dev)
    d_cb = torch.randn(P, 3, generator=g).to(dev)
    d_cc = torch.randn(P, 3, generator=g).to(dev)
    feat = eng.forward(x, need_grad_state=False, feat_ld=ceng.cin_ld)["feat"]      # fp32 features for both runs

    def run():
        cb, cc, logits, cst = ceng.forward(feat, rays_d, S, P)
        d_lg = torch.randn(P, logits.shape[1], generator=torch.Generator().manual_seed(7)).to(dev) if logits is not None else None
        grads, dCIN = ceng.backward(cst, cb, cc, d_cb, d_cc, d_lg)
        return cb.clone(), cc.clone(), cst, [t.clone() for t in grads], dCIN.clone()

    base = mlp.PRECISION          # the library default (bf16x3), or whatever NUDF_PRECISION selects
    cb32, cc32, st32, g32, d32 = run()
    try:
        mlp.set_precision("mixed16")
        cb16, cc16, st16, g16, d16 = run()
    finally:
        mlp.set_precision(base)
    assert _psnr(cc16, cc32) > 55.0 and _psnr(cb16, cb32) > 55.0
    for name in ("HB", "HV"):
        assert st16[name][0].dtype == torch.float32          # CIN / VIN
        for l in range(1, len(st32[name])):
            a, ref = st16[name][l], st32[name][l][:P]
            assert a.dtype == torch.bfloat16 and mlp._isp4(a)
            got = mlp.unpack16(a)[:P]
            row = (got - ref).norm(dim=1) / (ref.norm(dim=1) + 1e-3 * float(ref.norm(dim=1).max()))
            assert float(row.max()) < 0.1, (name, l, float(row.max()))
            assert bool(((got > 0) == (ref > 0)).float().mean() > 0.995)      # the ReLU masks of the reverse sweep
    assert _cos(d16[:, :256], d32[:, :256]) > 0.99
    for a, b in zip(g16, g32):
        if float(b.abs().max()) > 0:
            assert _cos(a, b) > 0.98, _cos(a, b)
