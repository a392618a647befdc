"""Training-trajectory parity: k optimiser steps of the HIP path (render + ColorLoss + eikonal +
backward + fused Adam) against the CPU oracle trained with torch.optim.Adam from the same initial weights on the same
rays.  Fixed sample positions (no importance sampling) keep the comparison free of quantile-bin flips."""
import math

import pytest
import torch

from common import state_dicts
from oracle import udf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_outside", [0, 8])
def test_five_training_steps_follow_the_oracle(n_outside):
    from neuraludf_amd import synth
    from neuraludf_amd.train import Trainer
    dev = torch.device("cuda:0")
    rconf = dict(n_samples=32, n_importance=0, n_outside=n_outside, up_sample_steps=1, perturb=0.0)
    tr = Trainer(dev, rconf, seed=0, fused_adam=True)
    tr.renderer.diagnostics = False
    rays = synth.make_rays(synth.make_scene("tiny"), 0, 64, seed=21)
    batch = {k: v.to(dev) for k, v in rays.items()}

    # oracle twin: same initial weights, same parameter groups / learning rates as exp_runner_blending.py:136-139
    sds = {k: {n: t.detach().cpu().clone() for n, t in m.state_dict().items()} for k, m in tr.modules().items()}
    nets = O.Nets(**{k: {n: t.clone().requires_grad_(True) for n, t in sds[k].items()}
                     for k in ("udf", "color", "var", "beta", "nerf")})
    nets.beta["gamma"].requires_grad_(False)
    nets.beta["zeta"].requires_grad_(False)
    geo = list(nets.udf.values())
    other = list(nets.var.values()) + list(nets.color.values()) + [nets.beta["beta"]]
    nerf = list(nets.nerf.values())
    opt = torch.optim.Adam([{"params": geo, "lr": 1e-4}, {"params": other}, {"params": nerf}], lr=5e-4)
    cfg = O.RenderCfg(n_samples=32, n_importance=0, n_outside=n_outside, up_sample_steps=1)

    ref_losses, gpu_losses = [], []
    for _ in range(5):
        out = O.render(nets, cfg, rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=1.0,
                       flip_saturation=1.0)
        cl = O.color_loss(0.01, 1.0, 0.0, 0.0, 3, out["color_base"], out["color"], rays["true_rgb"], None, None,
                          None, None, None)
        loss = cl["loss"] + 0.1 * out["gradient_error"]
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
        l, _ = tr.step(batch, cos_anneal_ratio=1.0, flip_saturation=1.0, perturb_overwrite=0)
        gpu_losses.append(float(l))
    for a, b in zip(gpu_losses, ref_losses):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (gpu_losses, ref_losses)
    # the parameters themselves after 5 steps.  Adam normalises the update -- every weight moves by ~lr per step whatever the
    # size of its gradient -- so an element whose gradient is three orders below its tensor's largest follows fp32 noise:
    # from step 2 on such elements' gradients differ by tens of percent between ANY two fp32 evaluations (measured,
    # scripts/debug_train_parity.py: the exact-fp32 kernels land 2.3e-4 from the oracle on color.lin1.weight_g[80], the bf16x3
    # kernels 2.3e-4, the f16x2 forward sweeps 2.9e-4 on another element of the same ReLU network with outside samples and
    # 3.3e-6 -- thirty times closer than either -- without).  Two statements are held: no element is further from the oracle
    # than one step of one sign flip (lr = 5e-4; a systematic error would show as 5 steps x lr = 2.5e-3), and per network the
    # 5-step movement agrees in the 2-norm to 2 % (measured: colour 0.8 %, UDF 0.05 %, NeRF 0.02 %).
    worst = 0.0
    for k, m in tr.modules().items():
        if k == "nerf" and n_outside == 0:
            continue
        num = den = 0.0
        for n, p in m.state_dict().items():
            ref = getattr(nets, k)[n].detach()
            worst = max(worst, float((p.detach().cpu() - ref).abs().max()))
            mv_h, mv_r = p.detach().cpu().double() - sds[k][n].double(), ref.double() - sds[k][n].double()
            num += float((mv_h - mv_r).pow(2).sum())
            den += float(mv_r.pow(2).sum())
        assert den > 0 and (num / den) ** 0.5 < 2e-2, (k, (num / max(den, 1e-300)) ** 0.5)
    assert worst < 4e-4, worst
    # and the colours rendered by the two trained models
    with torch.no_grad():
        _, o_gpu = tr.loss(batch, cos_anneal_ratio=1.0, flip_saturation=1.0, perturb_overwrite=0)
        o_ref = O.render(nets, cfg, rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=1.0,
                         flip_saturation=1.0)
    mse = float(((o_gpu["color"].cpu() - o_ref["color"]) ** 2).mean())
    assert 20.0 * math.log10(1.0 / math.sqrt(mse + 1e-20)) > 60.0
