"""The fused loss assembly of the single-process step (loss._StepLossFn: nudf_step_loss_fwd / _bwd -- ColorLoss's L1 terms,
the three regularisers and the runner's weighted total in one launch each way) against the unfused chain it replaces
(_ColorLossFn + _ErrorsFn + scalar torch ops), and the column-0 seed kernel of the UDF head's adjoint."""
import pytest
import torch

from neuraludf_amd import synth

pytestmark = pytest.mark.gpu

RCONF = dict(n_samples=32, n_importance=32, n_outside=0, up_sample_steps=2, perturb=1.0)


def _one(fused, tconf, mask_weight=0.0):
    from neuraludf_amd.train import Trainer
    dev = torch.device("cuda:0")
    tr = Trainer(dev, RCONF, seed=0, fused_adam=True, train_conf=dict(tconf, mask_weight=mask_weight))
    tr.fuse_loss = fused
    batch = {k: v.to(dev) for k, v in synth.make_rays(synth.make_scene("tiny"), 0, 160, seed=4).items()}
    batch["mask"] = (torch.rand(160, 1, generator=torch.Generator().manual_seed(1)) > 0.3).float().to(dev)
    loss, out = tr.loss(batch, cos_anneal_ratio=0.6, flip_saturation=0.9, perturb_overwrite=0)
    loss.backward()
    grads = {n: p.grad.clone() for m in tr.modules().values() for n, p in m.named_parameters() if p.grad is not None}
    terms = {k: float(out[k]) for k in ("gradient_error", "gradient_error_near_surface", "sparse_error")}
    return float(loss), terms, grads


@pytest.mark.parametrize("tconf", [dict(igr_weight=0.1, igr_ns_weight=0.0, sparse_weight=0.0),
                                   dict(igr_weight=0.1, igr_ns_weight=0.05, sparse_weight=0.01)],
                         ids=["shipped_weights", "all_regularisers_on"])
def test_fused_step_loss_equals_the_unfused_chain(tconf):
    la, ta, ga = _one(False, tconf)
    lb, tb, gb = _one(True, tconf)
    assert la == lb, (la, lb)                      # same reductions, the total's products / sums rounded one by one
    assert ta == tb
    assert set(ga) == set(gb)
    for n in ga:
        assert torch.equal(ga[n], gb[n]), n


def test_mask_loss_keeps_the_generic_path():
    """with the mask BCE term on the fused launch does not apply: both settings take the generic assembly"""
    tconf = dict(igr_weight=0.1, igr_ns_weight=0.0, sparse_weight=0.0)
    la, _, ga = _one(False, tconf, mask_weight=0.1)
    lb, _, gb = _one(True, tconf, mask_weight=0.1)
    assert abs(la - lb) <= 1e-7 * max(1.0, abs(la))
    for n in ga:
        assert float((ga[n] - gb[n]).abs().max()) <= 1e-6 * float(ga[n].abs().max().clamp(min=1e-12)), n


def test_col0_seed4_kernel():
    from neuraludf_amd._lib import call, ptr
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    P, Pp = 1000, 1024
    sign = torch.sign(torch.randn(P, generator=g)).to(dev)
    d = torch.randn(P, generator=g).to(dev)
    out = torch.full((Pp, 4), 7.0, device=dev)
    both = torch.full((Pp, 4), 7.0, device=dev)
    call("nudf_col0_seed4", ptr(sign), ptr(d), 0.37, P, Pp, ptr(out), ptr(both))
    ref = torch.zeros(Pp, 4, device=dev)
    ref[:P, 0] = sign * d * 0.37
    assert torch.equal(out, ref)
    call("nudf_col0_seed4", ptr(sign), None, 0.37, P, Pp, ptr(out), None)
    ref[:P, 0] = sign * 0.37
    assert torch.equal(out, ref)
    assert torch.equal(both, ref)          # the second output of the first launch: sign * scale
