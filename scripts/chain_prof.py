"""profile target: the fused forward chain alone (udf only / with state / gradient) at P points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from common import build_modules, perturb_
from neuraludf_amd import mlp
from neuraludf_amd.models import fields
dev = torch.device("cuda:0")
mods = perturb_(build_modules(fields, seed=0))
eng = mods["udf"].to(dev).engine()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mode = sys.argv[2] if len(sys.argv) > 2 else "udf"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
x = (torch.rand(P, 3) * 2 - 1).to(dev)
st = eng.forward(x, True, 288)
for _ in range(n):
    if mode == "udf":
        eng.forward(x, False, udf_only=True)
    elif mode == "state":
        eng.forward(x, True, 288)
    elif mode == "grad":
        eng.gradient(x, st)
torch.cuda.synchronize()
