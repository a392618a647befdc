"""The non-finite status word (include/nudf.h: nudf_set_status_flag) -- what replaces the reference's host-side NaN stops
(/root/reference/models/udf_renderer_blending.py:97-101, 265-269, 543-544, 860-864): the up-sampling, composite and step-loss
kernels OR a bit into ONE int32 of device memory; nothing waits for it, the caller reads it when it likes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(dev, n_importance=16, fused=True):
    from neuraludf_amd.train import Trainer
    rconf = dict(n_samples=32, n_importance=n_importance, n_outside=0, up_sample_steps=2, perturb=1.0)
    return Trainer(dev, rconf, seed=0, fused_adam=fused)


def _batch(dev, n=64):
    from neuraludf_amd import synth
    rays = synth.make_rays(synth.make_scene("tiny"), 0, n, seed=3)
    return {k: v.to(dev) for k, v in rays.items()}


def test_clean_step_leaves_the_word_zero_and_planted_nans_set_the_bits():
    from neuraludf_amd import _lib
    dev = torch.device("cuda:0")
    tr = _trainer(dev)
    batch = _batch(dev)
    tr.renderer.clear_status()
    for _ in range(2):
        tr.step(batch)
    assert tr.renderer.status() == 0
    tr.renderer.check_finite()                      # nothing to report
    # (a) a NaN renderer scalar: the clips of :373-377 are hardware min / max here and would swallow it (inv_s = 1e-6,
    # finite weights, a finite loss) -- the composite launch reports the parameter itself
    good = tr.var.variance.detach().clone()
    with torch.no_grad():
        tr.var.variance.fill_(float("nan"))
    tr.loss(batch)
    assert tr.renderer.status(clear=True) & _lib.STATUS_NONFINITE_RENDER
    with torch.no_grad():
        tr.var.variance.copy_(good)
    tr.loss(batch)
    assert tr.renderer.status() == 0
    # (b) a NaN in the colour head (a NaN in a HIDDEN ReLU layer is swallowed by relu = max(x, 0), a hardware max): NaN base
    # colours -> the rays' composited colours and the loss
    with torch.no_grad():
        tr.color.lin_base4.bias.fill_(float("nan"))
    loss, out = tr.loss(batch)
    assert not bool(torch.isfinite(out["color_base"]).any()) and not bool(torch.isfinite(loss))
    bits = tr.renderer.status()
    assert bits & _lib.STATUS_NONFINITE_RENDER, bits
    assert bits & _lib.STATUS_NONFINITE_LOSS, bits
    assert not bits & _lib.STATUS_NONFINITE_SAMPLES, bits      # the sampling does not see the colour network
    with pytest.raises(FloatingPointError) as e:
        tr.renderer.check_finite()
    assert "composited" in str(e.value) and "loss" in str(e.value)
    assert tr.renderer.status() == 0                # check_finite cleared it


def test_upsample_kernel_alone_reports_a_nan_sample():
    """the kernel-level contract: nudf_upsample whose new samples come out non-finite (here: a NaN sample position on one
    ray) ORs bit 2 and nothing else; a NaN udf VALUE is swallowed by the min / max clips of the alphas and reports nothing"""
    from neuraludf_amd import _lib
    dev = torch.device("cuda:0")
    tr = _trainer(dev)
    r = tr.renderer
    b = _batch(dev, 16)
    z = (b["near"] + (b["far"] - b["near"]) * torch.linspace(0, 1, 32, device=dev)[None]).contiguous()
    udf = torch.rand(16, 32, device=dev) * 0.2
    sd = torch.tensor([float(((b["far"] - b["near"]) / 32).mean())], device=dev)
    r.clear_status()
    with torch.no_grad():
        r._upsample(b["rays_o"], b["rays_d"], z, udf, sd, 8, 0, 64.0, 0.1, 20.0)
    assert r.status() == 0
    z[3, :] = float("nan")
    with torch.no_grad():
        zn, _ = r._upsample(b["rays_o"], b["rays_d"], z, udf, sd, 8, 0, 64.0, 0.1, 20.0)
    assert not bool(torch.isfinite(zn[3]).any()) and bool(torch.isfinite(zn[:3]).all())
    assert r.status(clear=True) == _lib.STATUS_NONFINITE_SAMPLES


def test_a_replayed_graph_keeps_reporting():
    """the word's address is a kernel argument: a captured step ORs into the word it was captured with"""
    from neuraludf_amd import _lib
    from neuraludf_amd.train import GraphedStep
    dev = torch.device("cuda:0")
    tr = _trainer(dev)
    batch = _batch(dev)
    g = GraphedStep(tr, eager_steps=1)
    tr.renderer.clear_status()
    for _ in range(4):
        g(batch)
    assert g.replays >= 2
    assert tr.renderer.status() == 0
    with torch.no_grad():
        tr.color.lin_base4.bias.fill_(float("nan"))
    g(batch)
    bits = tr.renderer.status(clear=True)
    assert bits & _lib.STATUS_NONFINITE_RENDER and bits & _lib.STATUS_NONFINITE_LOSS, bits
