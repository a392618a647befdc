"""Why does bench.py's forward-only figure swing between 2.3 and 5.7 ms per pass from box to box while the train step
stays at 5.8?  Times the forward-only pass (render + loss under no_grad) three ways: wall clock, HIP events on the
stream, and the sum of kernel durations seen by the host loop when every launch is followed by a synchronize
(= pure GPU time)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuraludf_amd import synth
from neuraludf_amd.train import Trainer
dev = torch.device("cuda:0")
rconf = dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0)
tr = Trainer(dev, rconf, seed=0, fused_adam=True)
tr.renderer.diagnostics = False
rays = synth.make_rays(synth.make_scene("dtu"), 0, 512, seed=1234)
batch = {k: v.to(dev) for k, v in rays.items()}
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
if os.environ.get("PROBE_GC_FREEZE") == "1":
    gc.collect(); gc.freeze()
if os.environ.get("PROBE_GC_OFF") == "1":
    gc.disable()
for rep in range(4):
    g0 = [s["collections"] for s in gc.get_stats()]
    m0 = torch.cuda.memory_stats()
    with torch.no_grad():
        for _ in range(3):
            tr.loss(batch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(20):
            tr.loss(batch)
        t_host = time.perf_counter() - t0          # host finished ENQUEUEING
        e1.record(); torch.cuda.synchronize()
        t_wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(batch)
    torch.cuda.synchronize()
    t_step = time.perf_counter() - t0
    g1 = [s["collections"] for s in gc.get_stats()]
    m1 = torch.cuda.memory_stats()
    print(f"   gc collections (gen0,1,2) during rep: {[b - a for a, b in zip(g0, g1)]}; device allocs {m1['num_device_alloc'] - m0['num_device_alloc']} "
          f"frees {m1['num_device_free'] - m0['num_device_free']} retries {m1['num_alloc_retries'] - m0['num_alloc_retries']}")
    print(f"rep {rep}: forward-only wall {t_wall / 20 * 1e3:.2f} ms, host enqueue {t_host / 20 * 1e3:.2f} ms, HIP events "
          f"{e0.elapsed_time(e1) / 20:.2f} ms | train step wall {t_step / 20 * 1e3:.2f} ms", flush=True)
# CPU speed of this box (python-level): a fixed pure-python loop
t0 = time.perf_counter(); s = 0
for i in range(2_000_000):
    s += i & 7
print(f"pure-python 2M-iteration loop: {(time.perf_counter() - t0) * 1e3:.0f} ms; cpu count {os.cpu_count()}")
