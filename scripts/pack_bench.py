"""where does nudf_weightnorm_pack_multi spend its time?  The UDF network's pack launch (9 layers, the bf16x3 fragment kinds of a
train step) with the fragment copies, W / W^T and everything else switched off in turn."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import torch
from common import build_modules, perturb_
from neuraludf_amd import mlp
from neuraludf_amd.models import fields
dev = torch.device("cuda:0")
mods = perturb_(build_modules(fields, seed=0))
udf = mods["udf"].to(dev)
eng = udf.engine()
kinds = eng._frag_kinds()
print("kinds per layer:", kinds[0], kinds[4], kinds[-1])

def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def run(ks):
    def f():
        eng.mark_stale()
        mlp.pack_group(eng.layers, ks)
    return t(f)
print("all fragment kinds      %.1f us" % run(kinds))
print("no fragments            %.1f us" % run([()] * len(kinds)))
for i in range(4):
    print("only kind %d of each layer %.1f us" % (i, run([tuple(k[i:i + 1]) for k in kinds])))
# without W / W^T: drop the buffers' pointers by monkeypatching _ensure_buffers results is intrusive; time an empty-ish launch instead
one = [eng.layers[0]]
print("one layer, no fragments %.1f us" % t(lambda: (one[0].__setattr__("_ver", None), mlp.pack_group(one, [()]))))
