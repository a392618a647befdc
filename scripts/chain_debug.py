import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from common import build_modules, perturb_
from neuraludf_amd import mlp
from neuraludf_amd.models import fields
dev = torch.device("cuda:0")
mods = perturb_(build_modules(fields, seed=0))
udf = mods["udf"].to(dev)
eng = udf.engine()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator().manual_seed(0)
x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
mlp.USE_CHAIN = False
a = eng.forward(x, True, 288)
for tile in (32, 64):
    mlp.USE_CHAIN = True
    mlp.CHAIN_TILE = tile
    b = eng.forward(x, True, 288)
    torch.cuda.synchronize()
    for l in range(0, 9):
        d = (a["X"][l] - b["X"][l]).abs()
        bad = d > 1e-4 * a["X"][l].abs().max()
        rows = bad.any(1).nonzero().flatten().tolist()
        cols = bad.any(0).nonzero().flatten().tolist()
        print("tile", tile, "X", l, "max", float(d.max()), "bad rows", rows[:12], len(rows), "bad cols", cols[:12], len(cols))
    print("udf", float((a["udf"] - b["udf"]).abs().max()), "feat", float((a["feat"] - b["feat"]).abs().max()))
