"""per-wave timelines of the fused chain kernel (s_memtime stamps: K loop end, after barrier 1, epilogue end, after
barrier 2 of every step) for the UDF sweeps of a train step at P points: where do the cycles of a wave go, and how busy
is each SIMD's matrix pipe (two waves per SIMD: one of each of the CU's two workgroups)?"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from common import build_modules, perturb_
from neuraludf_amd import mlp
from neuraludf_amd.models import fields

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mods = perturb_(build_modules(fields, seed=0))
udf, col = mods["udf"].to(dev), mods["color"].to(dev)
eng, ceng = udf.engine(), col.engine()
g = torch.Generator().manual_seed(0)
x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
d_udf = torch.randn(P, generator=g).to(dev)
d_g = torch.randn(P, 3, generator=g).to(dev)
d_feat = torch.randn(P, ceng.cin_ld, generator=g).to(dev)

launches = []
orig_launch = mlp.ChainBuilder.launch


def launch(self):
    if mlp.CHAIN_DEBUG is not None:
        nb = (P + 31) // 32
        dbg = torch.zeros(nb * 4, 64, dtype=torch.int64, device=dev)
        mlp.CHAIN_DEBUG = dbg
        orig_launch(self)            # (fills the descriptor: the builder only records until launch)
        steps = [(int(self.c.step[i].epi), int(self.c.step[i].K), int(self.c.step[i].N)) for i in range(self.n)]
        launches.append((dbg, steps, self.flops, [int(self.c.step[i].prec) for i in range(self.n)]))
        return
    orig_launch(self)


mlp.ChainBuilder.launch = launch


S = 128
rays_d = torch.nn.functional.normalize(torch.randn(P // S, 3, generator=g), dim=-1).to(dev)
d_cb = torch.randn(P, 3, generator=g).to(dev)
d_cc = torch.randn(P, 3, generator=g).to(dev)
COLOUR = os.environ.get("TIMELINE_COLOUR", "0") == "1"


def run():
    st = eng.forward(x, need_grad_state=True, feat_ld=ceng.cin_ld)
    if COLOUR:      # the colour net's forward (both branches) and its two reverse sweeps instead of the UDF backward
        cb, cc, logits, cst = ceng.forward(st["feat"], rays_d, S, P)
        d_lg = torch.zeros_like(logits) if logits is not None else None
        ceng.backward(cst, cb, cc, d_cb, d_cc, d_lg)
        return
    gr, DA = eng.gradient(x, st)
    eng.backward(x, st, DA, d_udf, d_feat, ceng.cin_ld, d_g)


WARM = int(os.environ.get("TIMELINE_WARM", "12"))     # back-to-back sweeps before the stamped one: steady-state clocks
for _ in range(WARM):
    run()
mlp.CHAIN_DEBUG = torch.zeros(1, dtype=torch.int64, device=dev)
run()                                                   # enqueued behind the warm-up without a host sync in between
torch.cuda.synchronize()
mlp.CHAIN_DEBUG = None
from neuraludf_amd._lib import CH as _CH
EPI = {v: k for k, v in _CH.items()}

for li, (dbg, steps, flops, precs) in enumerate(launches):
    d = dbg.cpu().numpy()
    d = d[d[:, 1] > 0]
    n = len(steps)
    t0 = d[:, 1].min()
    tend = d[:, 5 + 4 * (n - 1)].max()
    span = float(tend - t0)
    # phases per wave
    k = np.zeros(len(d)); w1 = np.zeros(len(d)); ep = np.zeros(len(d)); w2 = np.zeros(len(d))
    prev = d[:, 1].astype(np.float64)
    per_step = []
    for s in range(n):
        a, b, c, e = (d[:, 2 + 4 * s + j].astype(np.float64) for j in range(4))
        k += a - prev; w1 += b - a; ep += c - b; w2 += e - c
        per_step.append(((a - prev).mean(), (b - a).mean(), (c - b).mean(), (e - c).mean()))
        prev = e
    tot = k + w1 + ep + w2
    # matrix-pipe cycles per wave of a 64-point tile: algorithmic flops per point x 64 / 4 waves, per step on the pipe it runs on --
    # fp32: v_mfma_f32_32x32x2_f32 = 4 096 flops in 64 cycles; 16-bit: v_mfma_f32_32x32x16 = 32 768 flops in 32 cycles, times the
    # products of the step's operand mode (1: plain 16-bit, 6: bf16x3, 3: f16x2)
    mfma_cycles = 0.0
    for (epi_, K_, N_), pr in zip(steps, precs):
        fl = 2.0 * K_ * N_ * 64 / 4            # (padded K, N: what the tile loop executes)
        mfma_cycles += fl / 4096 * 64 if pr == 0 else fl * mlp.MFMA_PRODUCTS[pr] / 32768 * 32
    wall = (d[:, 63] - d[:, 62]).astype(np.float64)            # 100 MHz ticks
    mem = (d[:, 5 + 4 * (n - 1)] - d[:, 1]).astype(np.float64)
    ok = wall > 0
    mhz = float((mem[ok] / wall[ok]).mean() * 100.0) if ok.any() else float("nan")
    launch_us = float(d[:, 63].max() - d[:, 62].min()) / 100.0 if ok.any() else float("nan")
    print(f"launch {li}: {n} steps, {len(d)} waves, flops/point {flops / P:.0f}; s_memtime runs at {mhz:.0f} MHz; first start -> last end {launch_us:.1f} us"
          f" = {flops / launch_us / 1e6 if launch_us == launch_us else float('nan'):.1f} TF")
    print(f"   per wave (mean ticks): K loop {k.mean():8.0f} ({k.mean() / tot.mean():5.1%})  barrier-1 wait {w1.mean():7.0f} ({w1.mean() / tot.mean():5.1%})"
          f"  epilogue {ep.mean():7.0f} ({ep.mean() / tot.mean():5.1%})  barrier-2 wait {w2.mean():7.0f} ({w2.mean() / tot.mean():5.1%})  total {tot.mean():8.0f}")
    print(f"   MFMA ticks needed per wave {mfma_cycles:8.0f}: K-loop efficiency {mfma_cycles / k.mean():5.1%} (2 waves share a SIMD: 50 % = pipe saturated),"
          f" wave-level MFMA share {mfma_cycles / tot.mean():5.1%}")
    for s, (epi, K, N) in enumerate(steps):
        a, b, c, e = per_step[s]
        print(f"      step {s:2d} epi {EPI.get(epi, epi)!s:9s} K {K:3d} N {N:3d}: K loop {a:7.0f}  wait1 {b:6.0f}  epilogue {c:6.0f}  wait2 {e:6.0f}")
    # per-SIMD matrix-pipe occupancy: group waves by (hw id without wave slot) and start time
    hw = d[:, 0]
    key = collections.defaultdict(list)
    for i in range(len(d)):
        key[int(hw[i]) >> 4].append(i)
    print(f"   distinct (se, sh, cu, simd) keys {len(key)} (XCC id is not part of HW_ID: 8 XCDs alias)")
