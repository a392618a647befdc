"""why does a chain launch miss its descriptor memo?  logs, per call site, the first differing signature element between a
launch and every memoized launch of the same (site, P, record length)"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
import bench
from neuraludf_amd import mlp, synth
from neuraludf_amd.train import Trainer
dev = torch.device("cuda:0")
rays_per_gpu, rconf, scene_kind = bench.WORKLOADS["dtu_scan24_512x128"]
tr = Trainer(dev, rconf, seed=0, fused_adam=True)
tr.renderer.diagnostics = False
rays = synth.make_rays(synth.make_scene(scene_kind), 0, 256, seed=1234)
batch = {k: v.contiguous().to(dev) for k, v in rays.items()}
orig = mlp.ChainBuilder.launch
log = []
def launch(self):
    sig = self._signature()
    ms = mlp._CHAIN_MEMO.get((self.site, self.P, len(self.rec)))
    if ms and sig is not None and not any(m[0] == sig for m in ms):
        diffs = []
        for m in ms:
            d = [i for i, (a, b) in enumerate(zip(m[0], sig)) if a != b]
            diffs.append((len(d), d[:6], [(m[0][i], sig[i]) for i in d[:3]]))
        log.append((self.site[0], self.P, len(ms), diffs))
    return orig(self)
mlp.ChainBuilder.launch = launch
for it in range(8):
    n0 = len(log); h0 = mlp.chain_memo_hits
    tr.step(batch)
    torch.cuda.synchronize()
    print(f"step {it}: misses-with-memo {len(log) - n0}, hits {mlp.chain_memo_hits - h0}")
for e in log[-14:]:
    print(e)
