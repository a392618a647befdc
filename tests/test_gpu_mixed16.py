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

    assert mlp.PRECISION == "fp32"
    l32, c32, w32, g32 = run()
    try:
        mlp.set_precision("mixed16")
        l16, c16, w16, g16 = run()
    finally:
        mlp.set_precision("fp32")
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
