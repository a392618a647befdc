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
from chain_sweeps import compare, engines, sweeps as _sweeps  # noqa: E402
from neuraludf_amd import mlp  # noqa: E402

dev = torch.device("cuda:0")
what = set(sys.argv[1:]) or {"check", "time"}
eng = engines(dev)["eng"]


def sweeps(P, tile, seed=0):
    return _sweeps(dev, P, tile, seed)


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
