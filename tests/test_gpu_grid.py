"""GPU: dense-grid field queries for meshing: extract_fields / extract_gradient_fields of the drop-in
renderer module against the oracle's UDF / analytic gradient on the same grid (reference semantics:
models/udf_renderer_blending.py:16-49, caller exp_runner_blending.py:749-771)."""
import numpy as np
import pytest
import torch

from common import build_modules, perturb_, state_dicts, oracle_nets
from oracle import udf_oracle as O

pytestmark = pytest.mark.gpu


def test_grid_queries_match_oracle(monkeypatch):
    from neuraludf_amd.models import fields
    from neuraludf_amd.models import udf_renderer_blending as rb
    dev = torch.device("cuda:0")
    mods = perturb_(build_modules(fields, seed=0))
    sds = state_dicts(mods)
    udf = mods["udf"].to(dev)
    bmin, bmax = torch.tensor([-1.01, -0.9, -1.0]), torch.tensor([1.01, 1.1, 0.8])
    R = 37
    monkeypatch.setattr(rb, "GRID_CHUNK_POINTS", 5 * R * R)            # 8 slabs, ragged last one (37 = 7*5 + 2)
    u = rb.extract_fields(bmin, bmax, R, lambda p: -udf.udf(p), device=dev)
    g = rb.extract_gradient_fields(bmin, bmax, R, lambda p: udf.gradient(p).squeeze(), device=dev)
    assert u.shape == (R, R, R) and u.dtype == np.float32 and g.shape == (R, R, R, 3)
    # one big chunk gives the same volume (row tiles do not interact)
    monkeypatch.setattr(rb, "GRID_CHUNK_POINTS", 1 << 21)
    u1 = rb.extract_fields(bmin, bmax, R, lambda p: -udf.udf(p), device=dev)
    np.testing.assert_array_equal(u, u1)
    # oracle on the reference's grid: x-major meshgrid of the three linspaces
    ax = [torch.linspace(float(bmin[k]), float(bmax[k]), R) for k in range(3)]
    pts = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    nets = oracle_nets(sds)
    with torch.no_grad():
        ou = O.udf_forward(nets.udf, pts)[:, 0]
    og = O.udf_gradient(nets.udf, pts, create_graph=False).detach().reshape(-1, 3)
    np.testing.assert_allclose(u.reshape(-1), -ou.numpy(), atol=1e-4)
    gm = float(og.abs().max())
    np.testing.assert_allclose(g.reshape(-1, 3), og.numpy(), atol=2e-4 * max(1.0, gm))
