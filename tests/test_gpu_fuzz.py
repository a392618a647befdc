"""Randomised parity sweep: render() with fixed (un-resampled) sample positions over random ray counts, sample counts
(full and ragged 64-lane chunks, 1..4 chunks), outside-sample counts, cosine / annealing / flip-saturation settings,
against the oracle.  Complements the fixtures: the composite kernels have one instantiation per chunk count and a
specialised one for full chunks, the chains one per tile shape."""
import random

import pytest
import torch

from common import build_modules, perturb_, state_dicts, oracle_nets
from oracle import udf_oracle as O

pytestmark = pytest.mark.gpu


def _cases(n=14, seed=20260926):
    rng = random.Random(seed)
    out = [dict(n_rays=3, n_samples=64, n_outside=0, norm=False, anneal=1.0, flip=1.0),      # exactly one full chunk
           dict(n_rays=2, n_samples=128, n_outside=0, norm=True, anneal=None, flip=0.0),     # two full chunks
           dict(n_rays=5, n_samples=192, n_outside=0, norm=False, anneal=0.5, flip=0.9)]     # three full chunks
    while len(out) < n:
        out.append(dict(n_rays=rng.randint(1, 9), n_samples=rng.choice([2, 3, 17, 63, 65, 100, 127, 129, 200, 250]),
                        n_outside=rng.choice([0, 0, 1, 7, 33]), norm=rng.random() < 0.5,
                        anneal=rng.choice([None, 0.0, 0.3, 1.0]), flip=rng.choice([0.0, 0.5, 0.9, 1.0])))
    return out


@pytest.fixture(scope="module")
def nets():
    from neuraludf_amd.models import fields
    dev = torch.device("cuda:0")
    mods = perturb_(build_modules(fields, seed=0))
    sds = state_dicts(mods)
    for m in mods.values():
        m.to(dev)
    return mods, sds, dev


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "r%d_s%d_o%d_%s_a%s_f%s" % (
    c["n_rays"], c["n_samples"], c["n_outside"], "n" if c["norm"] else "u", c["anneal"], c["flip"]))
def test_render_matches_oracle(nets, case):
    from neuraludf_amd import synth
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    mods, sds, dev = nets
    kw = dict(n_samples=case["n_samples"], n_importance=0, n_outside=case["n_outside"], up_sample_steps=1,
              use_norm_grad_for_cosine=case["norm"])
    r = synth.make_rays(synth.make_scene("tiny"), 0, case["n_rays"], seed=case["n_samples"] * 31 + case["n_rays"])
    rend = UDFRendererBlending(mods["nerf"], mods["udf"], mods["var"], mods["color"], mods["beta"], perturb=0.0, **kw)
    with torch.no_grad():
        out = rend.render(r["rays_o"].to(dev), r["rays_d"].to(dev), r["near"].to(dev), r["far"].to(dev),
                          cos_anneal_ratio=case["anneal"], flip_saturation=case["flip"], perturb_overwrite=0)
        ref = O.render(oracle_nets(sds), O.RenderCfg(**kw), r["rays_o"], r["rays_d"], r["near"], r["far"],
                       cos_anneal_ratio=case["anneal"], flip_saturation=case["flip"])
    for k in ("z_vals", "udf", "weights", "color", "color_base", "depth", "normals", "weight_sum", "weight_sum_fg_bg",
              "vis_prob", "alpha", "true_cos", "gradient_error", "gradient_error_near_surface"):
        a, b = out[k].detach().cpu().float().reshape(-1), ref[k].detach().float().reshape(-1)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max())), k
