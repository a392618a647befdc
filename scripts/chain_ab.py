"""A/B of the two fused-chain kernels on the GPU: workgroup-shared tiles (tile 64) vs wave-private tiles (tile 128).

  python scripts/chain_ab.py [check] [time] [timeline]

check    : every sweep of the UDF net (value + state, input gradient, second-order backward), the colour net
           (forward, backward) and the NeRF with both kernels on the same inputs -- same fp32 summation order, so the
           results must agree to the last bit (reported: max abs difference, bit-equality);
time     : per-launch HIP-event times / TFLOP/s of every chain launch at 32 768 / 65 536 / 262 144 points;
timeline : s_memtime stamps of the wave-private kernel (K loop / epilogue cycles per step).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
from common import build_modules, perturb_  # noqa: E402
from neuraludf_amd import mlp  # noqa: E402
from neuraludf_amd.models import fields  # noqa: E402

dev = torch.device("cuda:0")
mods = perturb_(build_modules(fields, seed=0))
udf = mods["udf"].to(dev)
col = mods["color"].to(dev)
nerf = mods["nerf"].to(dev)
eng = udf.engine()
ceng = col.engine()
neng = nerf.engine()
what = set(sys.argv[1:]) or {"check", "time"}


def sweeps(P, tile, seed=0, S=64):
    """all chain launches of one train step's MLP work at P points -> dict of result tensors."""
    mlp.CHAIN_TILE = tile
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    d_udf = torch.randn(P, generator=g).to(dev)
    d_g = torch.randn(P, 3, generator=g).to(dev)
    rays_d = torch.nn.functional.normalize(torch.randn((P + S - 1) // S, 3, generator=g), dim=-1).to(dev)
    out = {}
    st = eng.forward(x, need_grad_state=True, feat_ld=ceng.cin_ld)
    out.update(udf=st["udf"], sign=st["sign"], feat=st["feat"][:, :256], X4=st["X"][4][:P, :256], X8=st["X"][8][:P])
    gr, DA = eng.gradient(x, st)
    out.update(g=gr, DA0=DA[0][:P], DA3=DA[3][:P, :217], DA7=DA[7][:P])
    out["uo"] = eng.forward(x, need_grad_state=False, udf_only=True)["udf"]
    # colour net on the UDF features
    CIN = st["feat"]
    Pc = (P // S) * S
    cb, cc, logits, cst = ceng.forward(CIN, rays_d, S, Pc)
    out.update(cb=cb, cc=cc)
    if logits is not None:
        out["logits"] = logits
    d_cb = torch.randn(Pc, 3, generator=g).to(dev)
    d_cc = torch.randn(Pc, 3, generator=g).to(dev)
    d_lg = torch.randn(Pc, logits.shape[1], generator=g).to(dev) if logits is not None else None
    cgr, dCIN = ceng.backward(cst, cb, cc, d_cb, d_cc, d_lg)
    out["dCIN"] = dCIN[:, :256]
    for i, t in enumerate(cgr):
        out[f"cg{i}"] = t
    d_feat = torch.zeros(P, ceng.cin_ld, device=dev)
    d_feat[:Pc] = dCIN[:Pc]
    grads = eng.backward(x, st, DA, d_udf, d_feat, ceng.cin_ld, d_g)
    for i, t in enumerate(grads):
        out[f"p{i}"] = t
    # background NeRF
    Pn = (min(P, 32768) // S) * S
    pts4 = torch.randn(Pn, 4, generator=g).to(dev) * 0.5
    sig, rgb = nerf.evaluate(pts4, rays_d[:Pn // S].contiguous(), S)
    out.update(nsig=sig.detach(), nrgb=rgb.detach())
    (sig.sum() + (rgb * torch.randn(rgb.shape, generator=g).to(dev)).sum()).backward()
    for i, prm in enumerate(nerf.parameters()):
        if prm.grad is not None:
            out[f"n{i}"] = prm.grad.detach().clone()
            prm.grad = None
    return out


def compare(a, b, tag):
    """value tensors must agree to fp32 rounding; ReLU-net gradients additionally flip whole elements where a
    pre-activation sits within an ulp of 0 (the two kernels add the bias in a different order), so they are judged by
    their relative L2 difference and the fraction of elements that moved."""
    worst, worst_l2, nbit, bad = 0.0, 0.0, 0, []
    for k in a:
        d = (a[k] - b[k]).abs()
        ref = float(a[k].abs().max()) + 1e-30
        l2 = float(d.double().pow(2).sum().sqrt() / (a[k].double().pow(2).sum().sqrt() + 1e-30))
        if torch.equal(a[k], b[k]):
            nbit += 1
        smooth = k[0] not in "cn" or k in ("cb", "cc", "nsig", "nrgb")       # colour / NeRF gradients have ReLU kinks
        frac = float((d > 1e-5 * ref).float().mean())
        worst = max(worst, float(d.max()) / ref if smooth else 0.0)
        worst_l2 = max(worst_l2, l2)
        if (smooth and float(d.max()) / ref > 2e-5) or l2 > 2e-3 or frac > 2e-2:
            bad.append((k, float(d.max()) / ref, l2, frac))
    print(f"check {tag}: {len(a)} tensors, {nbit} bit-identical, worst rel diff (smooth tensors) {worst:.3e}, "
          f"worst rel L2 {worst_l2:.3e}", "MISMATCH " + str(bad) if bad else "OK", flush=True)


if "check" in what:
    for P in (1000, 128 * 300 + 77, 65536):
        a = sweeps(P, 64)
        b = sweeps(P, 128)
        c = sweeps(P, 128)
        compare(a, b, f"P={P} shared-vs-rows")
        compare(b, c, f"P={P} rows-vs-rows(rerun)")

TILES = tuple(int(t) for t in os.environ.get("AB_TILES", "64,128").split(","))

if "time" in what:
    for P in (32768, 65536, 262144):
        for tile in TILES:
            sweeps(P, tile)           # warm-up (packing, allocator)
            torch.cuda.synchronize()
            mlp.PROFILE = []
            for _ in range(3):
                sweeps(P, tile)
            torch.cuda.synchronize()
            rec = mlp.PROFILE
            mlp.PROFILE = None
            n = len(rec) // 3
            rows = []
            for i in range(n):
                name, fl = rec[i][0], rec[i][1]
                us = min(rec[i + k * n][2].elapsed_time(rec[i + k * n][3]) for k in range(3)) * 1e3
                rows.append((name, fl, us))
            tot_fl = sum(f for nm, f, u in rows if nm == "mlp_chain")
            tot_us = sum(u for nm, f, u in rows if nm == "mlp_chain")
            tn_fl = sum(f for nm, f, u in rows if nm == "gemm_tn")
            tn_us = sum(u for nm, f, u in rows if nm == "gemm_tn")
            print(f"time P={P} tile={tile}: chains {tot_us:.0f} us {tot_fl / tot_us / 1e6:.1f} TF | gemm_tn {tn_us:.0f} us "
                  f"{tn_fl / max(tn_us, 1e-9) / 1e6:.1f} TF")
            print("    " + " ".join(f"{fl / 1e9:.1f}G/{us:.0f}us={fl / us / 1e6:.0f}T" for nm, fl, us in rows if nm == "mlp_chain"),
                  flush=True)

if "timeline" in what:
    P = 65536
    x = (torch.rand(P, 3) * 2 - 1).to(dev)
    for tile, nwb in ((128, 4), (64, 4)):
        mlp.CHAIN_TILE = tile
        rows = 32 if tile == 128 else 64
        nb = (P + (rows * (4 if tile == 128 else 1)) - 1) // (rows * (4 if tile == 128 else 1))
        eng.forward(x, True, 288)
        dbg = torch.zeros(nb * 4, 32, dtype=torch.int64, device=dev)
        mlp.CHAIN_DEBUG = dbg
        eng.forward(x, True, 288)
        torch.cuda.synchronize()
        mlp.CHAIN_DEBUG = None
        d = dbg.cpu().numpy()
        d = d[d[:, 1] > 0]
        t0 = d[:, 1].min()
        import numpy as np
        kl = np.stack([d[:, 2 + 2 * s] - (d[:, 1] if s == 0 else d[:, 1 + 2 * s]) for s in range(10)], 1)
        ep = np.stack([d[:, 3 + 2 * s] - d[:, 2 + 2 * s] for s in range(10)], 1)
        tot = d[:, 21] - d[:, 1]
        print(f"timeline tile={tile}: waves {d.shape[0]}, kernel span {int(d[:, 21].max() - t0)} ticks, per-wave total median "
              f"{int(np.median(tot))} (min {int(tot.min())} max {int(tot.max())})")
        print("   K loop (+wait) median per step:", [int(v) for v in np.median(kl, 0)])
        print("   epilogue       median per step:", [int(v) for v in np.median(ep, 0)], flush=True)
