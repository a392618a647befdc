"""per-wave timeline of the fused chain kernel (s_memtime stamps): shows how the workgroups sharing a CU interleave."""
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
x = (torch.rand(P, 3) * 2 - 1).to(dev)
eng.forward(x, False, udf_only=True)
nb = (P + 63) // 64
dbg = torch.zeros(nb * 4, 32, dtype=torch.int64, device=dev)
mlp.CHAIN_DEBUG = dbg
eng.forward(x, False, udf_only=True)
torch.cuda.synchronize()
mlp.CHAIN_DEBUG = None
d = dbg.cpu().numpy()
hw = d[:, 0]
wave_id = hw & 0xf; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
t0 = d[:, 1].min()
import collections
# group waves by (se, sh, cu, simd) -- XCC id is not in HW_ID, so several XCDs alias; use start-time clustering too
key = collections.defaultdict(list)
for i in range(d.shape[0]):
    key[(int(se[i]), int(sh[i]), int(cu[i]), int(simd[i]))].append(i)
print("waves", d.shape[0], "distinct (se,sh,cu,simd)", len(key))
print("wave_id histogram", collections.Counter(wave_id.tolist()))
k0 = sorted(key)[0]
for k in [k0]:
    print("SIMD", k)
    for i in sorted(key[k], key=lambda i: d[i, 1])[:8]:
        st = d[i, 1:20]
        mma = [int(st[1 + 2 * s] - (st[2 * s] if s == 0 else st[2 * s])) for s in range(9)]
        epi = [int(st[2 + 2 * s] - st[1 + 2 * s]) for s in range(9)]
        print("blk", i // 4, "w", i % 4, "slot", int(wave_id[i]), "start", int(st[0] - t0), "total", int(st[18] - st[0]))
        print("    mma+wait", mma)
        print("    epilogue", epi)
