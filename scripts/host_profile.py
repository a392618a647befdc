"""host-side cost of one EAGER train step (the data-parallel path runs eager): cProfile over 30 steps at 256 rays, where the
GPU needs 2.3 ms and the host is the bound."""
import os, sys, cProfile, pstats, io, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
import bench
from neuraludf_amd import mlp, synth
from neuraludf_amd.train import Trainer
dev = torch.device("cuda:0")
mlp.set_precision("bf16x3")
rays_per_gpu, rconf, scene_kind = bench.WORKLOADS["dtu_scan24_512x128"]
rays_per_gpu = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tr = Trainer(dev, rconf, seed=0, fused_adam=True)
tr.renderer.diagnostics = False
scene = synth.make_scene(scene_kind)
rays = synth.make_rays(scene, 0, rays_per_gpu, seed=1234)
batch = {k: v.contiguous().to(dev) for k, v in rays.items()}
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
t0 = time.perf_counter()
for _ in range(30):
    tr.step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, with final sync %.3f ms/step" % ((t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(30):
        tr.step(batch)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
    print(s.getvalue()[:8000])
