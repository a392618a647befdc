"""All fused-chain launches of one train step's MLP work (UDF value + state, input gradient, second-order backward,
colour net forward / backward, background NeRF forward / backward) on seeded inputs -> dict of result tensors; and the
comparison used to A/B the two chain kernels (tests/test_gpu_chain_rows.py, scripts/chain_ab.py)."""
import torch

from common import build_modules, perturb_
from neuraludf_amd import mlp
from neuraludf_amd.models import fields

_state = {}


def engines(dev):
    if "eng" not in _state:
        mods = perturb_(build_modules(fields, seed=0))
        udf, col, nerf = mods["udf"].to(dev), mods["color"].to(dev), mods["nerf"].to(dev)
        _state.update(udf=udf, col=col, nerf=nerf, eng=udf.engine(), ceng=col.engine(), neng=nerf.engine())
    return _state


def sweeps(dev, P, tile, seed=0, S=64, seed_scale=1.0):
    """tile: 0 = automatic, 32 / 64 = workgroup-shared tiles, 128 = wave-private tiles.  seed_scale multiplies every loss adjoint
    the backward sweeps start from (d udf, d grad, d colour, d logits, the NeRF's d sigma / d rgb)."""
    st_ = engines(dev)
    eng, ceng, nerf = st_["eng"], st_["ceng"], st_["nerf"]
    old_tile = mlp.CHAIN_TILE
    mlp.CHAIN_TILE = tile
    try:
        g = torch.Generator().manual_seed(seed)
        x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
        d_udf = torch.randn(P, generator=g).to(dev) * seed_scale
        d_g = torch.randn(P, 3, generator=g).to(dev) * seed_scale
        rays_d = torch.nn.functional.normalize(torch.randn((P + S - 1) // S, 3, generator=g), dim=-1).to(dev)
        out = {}
        st = eng.forward(x, need_grad_state=True, feat_ld=ceng.cin_ld)
        ub = mlp.unblock
        out.update(udf=st["udf"], sign=st["sign"], feat=st["feat"][:, :256], X4=ub(st["X"][4])[:P, :256], X8=ub(st["X"][8])[:P])
        gr, DA = eng.gradient(x, st)
        out.update(g=gr, DA0=ub(DA[0])[:P], DA3=ub(DA[3])[:P, :217], DA7=ub(DA[7])[:P])
        out["uo"] = eng.forward(x, need_grad_state=False, udf_only=True)["udf"]
        # colour net on the UDF features
        Pc = (P // S) * S
        d_feat = torch.zeros(P, ceng.cin_ld, device=dev)
        if Pc > 0:
            cb, cc, logits, cst = ceng.forward(st["feat"], rays_d, S, Pc)
            out.update(cb=cb, cc=cc)
            if logits is not None:
                out["logits"] = logits
            d_cb = torch.randn(Pc, 3, generator=g).to(dev) * seed_scale
            d_cc = torch.randn(Pc, 3, generator=g).to(dev) * seed_scale
            d_lg = torch.randn(Pc, logits.shape[1], generator=g).to(dev) * seed_scale if logits is not None else None
            cgr, dCIN = ceng.backward(cst, cb, cc, d_cb, d_cc, d_lg)
            out["dCIN"] = dCIN[:Pc, :256]
            for i, t in enumerate(cgr):
                out[f"cg{i}"] = t
            d_feat[:Pc] = dCIN[:Pc]
        grads = eng.backward(x, st, DA, d_udf, d_feat, ceng.cin_ld, d_g)
        for i, t in enumerate(grads):
            out[f"p{i}"] = t
        # background NeRF
        Pn = (min(P, 32768) // S) * S
        if Pn > 0:
            pts4 = torch.randn(Pn, 4, generator=g).to(dev) * 0.5
            sig, rgb = nerf.evaluate(pts4, rays_d[:Pn // S].contiguous(), S)
            out.update(nsig=sig.detach(), nrgb=rgb.detach())
            (sig.sum() + (rgb * torch.randn(rgb.shape, generator=g).to(dev)).sum()).backward(
                gradient=torch.tensor(float(seed_scale), device=dev))
            for i, prm in enumerate(nerf.parameters()):
                if prm.grad is not None:
                    out[f"n{i}"] = prm.grad.detach().clone()
                    prm.grad = None
    finally:
        mlp.CHAIN_TILE = old_tile
    return out


def compare(a, b, tag, verbose=True, l2_tol=3e-3):
    """Value tensors must agree to fp32 rounding.  ReLU-net gradients additionally flip whole elements where a
    pre-activation sits within an ulp of 0 (the two kernels add the bias in a different order, so a ReLU mask can
    differ), which moves single rows by O(1): they are judged by their relative L2 difference and by the fraction of
    elements that moved (one flipped element of P points moves a weight gradient by O(1 / P), hence `l2_tol`).
    -> (bad list, bit-identical count, worst rel diff of the smooth tensors, worst rel L2)."""
    worst, worst_l2, nbit, bad = 0.0, 0.0, 0, []
    for k in a:
        d = (a[k] - b[k]).abs()
        ref = float(a[k].abs().max()) + 1e-30
        l2 = float(d.double().pow(2).sum().sqrt() / (a[k].double().pow(2).sum().sqrt() + 1e-30))
        if torch.equal(a[k], b[k]):
            nbit += 1
        smooth = k[0] not in "cn" and k != "dCIN" or k in ("cb", "cc", "nsig", "nrgb")   # colour / NeRF gradients: ReLU kinks
        frac = float((d > 1e-5 * ref).float().mean())
        worst = max(worst, float(d.max()) / ref if smooth else 0.0)
        worst_l2 = max(worst_l2, l2)
        if (smooth and float(d.max()) / ref > 2e-5) or l2 > l2_tol:
            bad.append((k, float(d.max()) / ref, l2, frac))
    if verbose:
        print(f"check {tag}: {len(a)} tensors, {nbit} bit-identical, worst rel diff (smooth tensors) {worst:.3e}, "
              f"worst rel L2 {worst_l2:.3e}", "MISMATCH " + str(bad) if bad else "OK", flush=True)
    return bad, nbit, worst, worst_l2
