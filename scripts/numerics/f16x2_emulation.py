"""CPU emulation of the operand splits of the fused chains against float64, BEFORE any kernel is written:
  fp32     -- exact fp32 products, fp32 accumulate (torch matmul)
  bf16x3   -- x = hi + mid + lo (three bf16 parts), six of nine products, fp32 accumulate   (mlp_chain.hip ch_mma16x3)
  f16x2    -- x = hi + 2^-11 lo (two fp16 parts), acc0 += hi hi', acc1 += hi lo' + lo hi', result acc0 + 2^-11 acc1
on the UDF network's forward sweep (value + features) and on its input-gradient reverse sweep (models/fields.py:192-231), seed-0
perturbed weights (tests/common.py), 4096 points in the unit ball.  Prints max error / max |ref| per quantity."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import build_modules, perturb_, state_dicts  # noqa: E402
from oracle import udf_oracle as O  # noqa: E402


def split_bf16x3(x):
    hi = x.bfloat16().float()
    r = x - hi
    mid = r.bfloat16().float()
    lo = (r - mid).bfloat16().float()
    return hi, mid, lo


def mm_fp32(a, w):          # a [P,K] @ w[K,N]
    return a @ w


def mm_bf16x3(a, w):
    a0, a1, a2 = split_bf16x3(a)
    w0, w1, w2 = split_bf16x3(w)
    # smallest first, as the kernel
    return (((((a0 @ w2) + (a2 @ w0)) + (a1 @ w1)) + (a0 @ w1)) + (a1 @ w0)) + (a0 @ w0)


def split_f16x2(x, s=2048.0):
    hi = x.half().float()
    lo = ((x - hi) * s).half().float()
    return hi, lo


def mm_f16x2(a, w):
    a0, a1 = split_f16x2(a)
    w0, w1 = split_f16x2(w)
    acc1 = (a0 @ w1) + (a1 @ w0)
    acc0 = a0 @ w0
    return acc0 + acc1 * (1.0 / 2048.0)


def forward(sd, x, mm, dt):
    cfg = O.UDFCfg()
    emb = O.posenc(x.to(dt), cfg.multires)
    h = emb
    acts, Ws = [], []
    for l in range(cfg.n_lin):
        if l in cfg.skip_in:
            h = torch.cat([h, emb], 1) / np.sqrt(2)
        W = O.wn_weight({k: v.to(dt) for k, v in sd.items()}, f"lin{l}")
        Ws.append(W)
        a = mm(h, W.t().contiguous()) + sd[f"lin{l}.bias"].to(dt)
        acts.append(a)
        h = O.softplus100(a) if l < cfg.n_lin - 1 else a
    return h, acts, Ws, emb


def gradient(sd, x, mm, dt):
    cfg = O.UDFCfg()
    h, acts, Ws, emb = forward(sd, x, mm, dt)
    sig = [torch.where(a * 100 > 20, torch.ones_like(a), torch.sigmoid(100 * a)) for a in acts[:-1]]
    delta = torch.sign(h[:, :1]) * Ws[-1][0:1, :]
    d_emb = torch.zeros_like(emb)
    for l in range(cfg.n_lin - 2, -1, -1):
        da = delta * sig[l]
        delta = mm(da, Ws[l].contiguous())
        if l in cfg.skip_in:
            delta = delta / np.sqrt(2)
            d_emb = d_emb + delta[:, -emb.shape[1]:]
            delta = delta[:, :-emb.shape[1]]
    d_emb = d_emb + delta
    g = d_emb[:, :3].clone()
    xx = x.to(dt)
    for k in range(cfg.multires):
        f = 2.0 ** k
        g = g + f * (d_emb[:, 3 + 6 * k: 6 + 6 * k] * torch.cos(xx * f) - d_emb[:, 6 + 6 * k: 9 + 6 * k] * torch.sin(xx * f))
    return h, g


def main():
    from neuraludf_amd.models import fields
    mods = perturb_(build_modules(fields, seed=0))
    sd = state_dicts(mods)["udf"]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4096, 3, generator=g)
    x = x / x.norm(dim=1, keepdim=True) * torch.rand(4096, 1, generator=g) ** (1 / 3)
    h64, g64 = gradient(sd, x, mm_fp32, torch.float64)
    print("max error / max |float64 value|:  udf, features, d udf / d x")
    for name, mm in (("fp32", mm_fp32), ("bf16x3", mm_bf16x3), ("f16x2", mm_f16x2)):
        h, gg = gradient(sd, x, mm, torch.float32)
        e = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        print(f"{name:8s} {e(h[:, :1].abs(), h64[:, :1].abs()):.3e} {e(h[:, 1:], h64[:, 1:]):.3e} {e(gg, g64):.3e}")


if __name__ == "__main__":
    main()
