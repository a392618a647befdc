"""CPU check (fp64) of the hand-derived backward that mlp.UDFEngine sequences on the GPU: forward sweep,
reverse sweep for d udf/dx, tangent sweep (forward-over-reverse) and adjoint sweep, written with plain
matrix products exactly as the kernels do them, against torch.autograd double-backward."""
import math

import torch

from common import build_modules, perturb_, state_dicts, oracle_nets
from oracle import udf_oracle as O


def _sp(a):
    return torch.nn.functional.softplus(a, beta=100.0, threshold=20.0)


def test_second_order_udf_backward_matches_autograd():
    torch.set_default_dtype(torch.float64)
    try:
        from neuraludf_amd.models import fields
        sds = state_dicts(perturb_(build_modules(fields, seed=0)))
        on = oracle_nets(sds, requires_grad=True, dtype=torch.float64)
        g = torch.Generator().manual_seed(0)
        P = 64
        x = torch.randn(P, 3, generator=g) * 0.7
        wy = torch.randn(P, 257, generator=g)
        wg = torch.randn(P, 3, generator=g)
        y = O.udf_forward(on.udf, x)
        gr = O.udf_gradient(on.udf, x, create_graph=True)
        ((y * wy).sum() + (gr * wg).sum()).backward()

        # ---- the engine's algebra ----
        sd = {k: v.detach() for k, v in on.udf.items()}
        L, E, skip = 8, 39, 4
        W = [O.wn_weight(sd, f"lin{l}") for l in range(L + 1)]
        b = [sd[f"lin{l}.bias"] for l in range(L + 1)]
        r2 = 1.0 / math.sqrt(2.0)
        emb = O.posenc(x, 6)
        X = [None] * (L + 1)
        S = [None] * L
        X[0] = emb
        for l in range(L):
            a = X[l] @ W[l].t() + b[l]
            S[l] = torch.where(100 * a > 20, torch.ones_like(a), torch.sigmoid(100 * a))
            h = _sp(a)
            X[l + 1] = torch.cat([h, emb], 1) * r2 if (l + 1) == skip else h
        aL = X[L] @ W[L].t() + b[L]
        sign = torch.sign(aL[:, :1])
        # reverse sweep
        DA = [None] * L
        DA[L - 1] = sign * W[L][0:1, :] * S[L - 1]
        for l in range(L - 1, 0, -1):
            d = DA[l] @ W[l]
            if l == skip:
                d = d * r2
                d = d[:, :W[l - 1].shape[0]]
            DA[l - 1] = d * S[l - 1]
        # tangent sweep, direction wg
        xg = x.clone().requires_grad_(True)
        (R0,) = torch.autograd.functional.jvp(lambda t: O.posenc(t, 6), (x,), (wg,))[1:] or (None,)
        R = [None] * (L + 1)
        EX = [None] * L
        R[0] = R0
        for l in range(L):
            t = R[l] @ W[l].t()
            EX[l] = t * DA[l] * 100.0 * (1.0 - S[l])
            EX[l] = torch.where(S[l] >= 1.0, torch.zeros_like(EX[l]), EX[l])
            r = t * S[l]
            R[l + 1] = torch.cat([r, R0], 1) * r2 if (l + 1) == skip else r
        # adjoint sweep
        AB = [None] * (L + 1)
        AB[L] = torch.cat([sign * wy[:, :1], wy[:, 1:]], 1)
        for l in range(L, 0, -1):
            d = AB[l] @ W[l]
            if l == skip:
                d = d[:, :W[l - 1].shape[0]] * r2
            AB[l - 1] = d * S[l - 1] + EX[l - 1]
        for l in range(L + 1):
            dW = AB[l].t() @ X[l]
            if l < L:
                dW = dW + DA[l].t() @ R[l]
            else:
                dW[0] = dW[0] + (sign * R[L]).sum(0)
            db = AB[l].sum(0)
            # weight_norm chain rule
            v, gg = sd[f"lin{l}.weight_v"], sd[f"lin{l}.weight_g"]
            inv = 1.0 / v.norm(dim=1, keepdim=True)
            dg = (dW * v).sum(1, keepdim=True) * inv
            dv = gg * inv * dW - v * (gg * inv ** 3 * (dW * v).sum(1, keepdim=True))
            for name, mine in [("weight_v", dv), ("weight_g", dg), ("bias", db)]:
                ref = on.udf[f"lin{l}.{name}"].grad
                err = float((mine - ref).abs().max() / ref.abs().max().clamp(min=1e-12))
                assert err < 1e-9, (l, name, err)
    finally:
        torch.set_default_dtype(torch.float32)
