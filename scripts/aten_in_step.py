"""Which ATen kernels does one train step launch, and from where?  One eager step of the bench workload under
torch.profiler with Python stacks: every CPU-side aten op that launched a GPU kernel, with its input shapes and the
innermost neuraludf_amd frames ("zero at::native kernels in the replayed step")."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
import bench
from neuraludf_amd import mlp, synth
from neuraludf_amd.train import Trainer
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "dtu_scan24_512x128"
mlp.set_precision("bf16x3")
rays_per_gpu, rconf, scene_kind = bench.WORKLOADS[wl]
lconf = bench.BLEND_WORKLOADS.get(wl)
tr = Trainer(dev, rconf, color_loss_conf=lconf, seed=0, data_parallel=False, fused_adam=True)
tr.renderer.diagnostics = False
scene = synth.make_scene(scene_kind)
rays = synth.make_rays(scene, 0, rays_per_gpu, seed=1234, margin=8 if lconf else 0)
batch = {k: v.contiguous().to(dev) for k, v in rays.items()}
kw = {}
if lconf:
    kw["blend"] = {k: v.to(dev) for k, v in synth.make_source_views(scene, 0, 8, hwc=True).items()}
    npx = (2 * rconf["h_patch_size"] + 1) ** 2
    batch["gt_patch_colors"] = torch.rand(batch["rays_o"].shape[0], npx, 3, device=dev)
for _ in range(3):
    tr.step(batch, **kw)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(batch, **kw)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
n = 0
for e in evs:
    kern = [k for k in e.kernels] if hasattr(e, "kernels") else []
    if not kern:
        continue
    # only leaf ops (an aten::zeros holds an aten::zero_ holds an aten::fill_: report the one that owns the kernel directly)
    if any(c.name.startswith("aten::") and getattr(c, "kernels", []) for c in (e.cpu_children or [])):
        continue
    n += 1
    st = [s for s in (e.stack or []) if "neuraludf_amd" in s or "bench.py" in s][:3]
    print("%-22s %-40s %s" % (e.name, str(e.input_shapes)[:40], " <- ".join(s.split("neuraludf_amd/")[-1] for s in st)))
print("aten ops that launched kernels:", n)
