"""debug probe: capture + replay the train step for one configuration (run per configuration in its own process)."""
import sys
import faulthandler
faulthandler.enable()
import torch
sys.path.insert(0, ".")
from neuraludf_amd import synth
from neuraludf_amd.train import Trainer, GraphedStep

rays, ns, ni, steps, prior, scene_kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
dev = torch.device("cuda:0")
rconf = dict(n_samples=ns, n_importance=ni, n_outside=0, up_sample_steps=steps, perturb=1.0)
scene = synth.make_scene(scene_kind)
batch = {k: v.to(dev) for k, v in synth.make_rays(scene, 0, rays, seed=5).items()}
if prior:
    t0 = Trainer(dev, rconf, seed=0, fused_adam=True)
    for _ in range(3):
        t0.step(batch)
    torch.cuda.synchronize()
tr = Trainer(dev, rconf, seed=0, fused_adam=True)
tr.renderer.diagnostics = bool(int(sys.argv[7])) if len(sys.argv) > 7 else True
gs = GraphedStep(tr, eager_steps=2)
for i in range(5):
    loss, _ = gs(batch)
torch.cuda.synchronize()
print("OK", sys.argv[1:], float(loss), gs.replays, flush=True)
