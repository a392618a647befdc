"""`FusedAdam` (one nudf_adam_step launch) against torch.optim.Adam on the same parameters / gradients,
with the runner's three parameter groups and a per-iteration learning-rate rewrite
(exp_runner_blending.py:136-139, :167-191), including a parameter that starts frozen."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_adam_matches_torch_adam():
    from neuraludf_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    shapes = [(256, 39), (256, 1), (256,), (1,), (217, 256), (3, 128), (5000,), (1025,), (1024,), (7, 3, 5)]
    ref_p = [torch.randn(s, generator=g).to(dev).requires_grad_(True) for s in shapes]
    new_p = [p.detach().clone().requires_grad_(True) for p in ref_p]

    def groups(ps):
        return [{"params": ps[:3], "lr": 1e-4}, {"params": ps[3:7]}, {"params": ps[7:]}]

    ref = torch.optim.Adam(groups(ref_p), lr=5e-4)
    new = FusedAdam(groups(new_p), lr=5e-4)
    for it in range(12):
        lr = 5e-4 * (1.0 - it / 20.0)
        for o in (ref, new):
            for gi, grp in enumerate(o.param_groups):
                grp["lr"] = lr * (0.2 if gi == 0 else 1.0)
        for i, (a, b) in enumerate(zip(ref_p, new_p)):
            if i == 3 and it < 4:          # frozen at first (like the variance network, :353-359)
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g).to(dev) * (10.0 ** ((i % 5) - 3))
            a.grad, b.grad = gr.clone(), gr.clone()
        ref.step()
        new.step()
    torch.cuda.synchronize()
    for a, b in zip(ref_p, new_p):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), a.shape
    # repeat steps reuse the filled launch descriptors (round 5): every step but the first and the one the frozen parameter joined
    assert new.plan_hits == 10, new.plan_hits
    # ... and a state reload (new state tensors) must drop them: one more step after load_state_dict on both sides
    new.load_state_dict(new.state_dict())
    ref.load_state_dict(ref.state_dict())
    for a, b in zip(ref_p, new_p):
        gr = torch.randn(a.shape, generator=g).to(dev) * 1e-2
        a.grad, b.grad = gr.clone(), gr.clone()
    ref.step()
    new.step()
    assert new.plan_hits == 10
    new.step()
    ref.step()
    assert new.plan_hits == 11
    torch.cuda.synchronize()
    for a, b in zip(ref_p, new_p):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), a.shape
    # state interchange with torch.optim.Adam checkpoints
    sd = new.state_dict()
    ref2 = torch.optim.Adam(groups([p.detach().clone().requires_grad_(True) for p in new_p]), lr=5e-4)
    ref2.load_state_dict(sd)
    assert int(ref2.state[ref2.param_groups[0]["params"][0]]["step"]) == 14
    assert int(ref2.state[ref2.param_groups[1]["params"][0]]["step"]) == 10


def test_fused_adam_invalidates_packed_weight_cache():
    """the optimizer writes parameters through raw pointers; the engines' packed-weight caches are keyed on the
    tensors' version counters, so a stale cache would silently freeze the network."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from common import build_modules, perturb_
    from neuraludf_amd.models import fields
    from neuraludf_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    mods = perturb_(build_modules(fields, seed=0))
    udf = mods["udf"].to(dev)
    x = (torch.rand(300, 3, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    opt = FusedAdam(udf.parameters(), lr=1e-2)
    y0 = udf(x).detach().clone()
    for p in udf.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    y1 = udf(x).detach()
    assert float((y1 - y0).abs().max()) > 1e-4, "forward did not see the updated parameters"
    fresh = build_modules(fields, seed=0)["udf"].to(dev)
    fresh.load_state_dict(udf.state_dict())
    y2 = fresh(x).detach()
    assert float((y1 - y2).abs().max()) <= 1e-6 * max(1.0, float(y2.abs().max()))
