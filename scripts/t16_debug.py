"""where do the 16-bit-tile kernel and the fp32-tile 16-bit kernel first differ? (forward sweep, every stored layer)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import chain_sweeps as CS
from neuraludf_amd import _lib, mlp
dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 19200
eng = CS.engines(dev)["eng"]
x = (torch.rand(P, 3, generator=torch.Generator().manual_seed(5)) * 2 - 1).to(dev)
mlp.set_precision("mixed16")
lib = _lib.lib()
res = {}
for on in (1, 0, 1):
    lib.nudf_set_chain_t16(on)
    st = eng.forward(x, need_grad_state=True, feat_ld=288)
    torch.cuda.synchronize()
    r = {"udf": st["udf"][:P].clone(), "feat": st["feat"][:P, :256].clone()}
    for l in range(1, 9):
        r[f"X{l}"] = mlp.unpack16(st["X"][l])[:P].clone()
    r["X0"] = st["X"][0][:P].clone()
    res.setdefault(on, []).append(r)
a, b, a2 = res[1][0], res[0][0], res[1][1]
for k in a:
    d = (a[k] - b[k]).abs()
    d2 = (a[k] - a2[k]).abs()
    nz = (d > 0)
    cols = nz.any(dim=0).nonzero().flatten().tolist() if d.dim() == 2 else []
    print(f"{k:5s} t16 vs fp32-tile: max {float(d.max()):.3e}, differing elements {int(nz.sum())} of {d.numel()}, columns {cols[:12]}{'...' if len(cols) > 12 else ''} ({len(cols)}); "
          f"t16 run-to-run max {float(d2.max()):.3e}")
rows = ((a["X5"] - b["X5"]).abs() > 0).any(dim=1).nonzero().flatten()
print("points whose X5 differs:", rows.numel(), rows[:24].tolist())
print("  row % 64:", (rows % 64)[:24].tolist())
print("  tile    :", (rows // 64)[:24].tolist())
xr = x[rows[:8]].cpu()
print("  x:", xr.tolist())
X4 = a["X4"][rows[:8]].cpu()
print("  |X4| min over cols 0..255 per point:", X4[:, :256].abs().min(dim=1)[0].tolist())
print("  X4 PE part (cols 217..255) of first point:", X4[0, 217:256].tolist())
urows = ((a["udf"] - b["udf"]).abs() > 0).nonzero().flatten()
print("points whose udf differs:", urows.numel(), "subset of X5 rows:", bool(set(urows.tolist()) <= set(rows.tolist())))
