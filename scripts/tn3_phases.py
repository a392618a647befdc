"""where does a k-step of the 128 x 128 bf16x3 weight-gradient kernel go?  Needs the stamp build
(scripts/build_variants.sh stamps gemm_tn_f32_mfma.hip -DNUDF_TN3_STAMPS=1; NUDF_LIB=.../libnudf_stamps.so): per-segment
shader-clock ticks of waves 0 / 3 of every workgroup, and the workgroups' start / end on the 100 MHz wall clock -- UDF adjoint
group at M points"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuraludf_amd import _lib, mlp
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
SHAPES = [(256, 40)] + [(256, 256)] * 3 + [(217, 256)] + [(256, 256)] * 3 + [(256, 256), (1, 256)]
torch.manual_seed(0)
jobs = []
for NA, NB in SHAPES:
    lda = max(4, (NA + 3) // 4 * 4)
    jobs.append((torch.randn(M, lda, device=dev), NA, torch.randn(M, NB, device=dev), NB,
                 torch.zeros((NA + 31) // 32 * 32, NB, device=dev), torch.zeros((NA + 31) // 32 * 32, device=dev)))
mlp.set_precision("bf16x3")
for _ in range(3):
    mlp.gemm_tn_grouped(jobs, M)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    mlp.gemm_tn_grouped(jobs, M)
e1.record()
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) * 100:.1f} us per launch (with the reduce)")
dbg = torch.zeros(16 * 1024, dtype=torch.int64, device=dev)
_lib.lib().nudf_set_tn_debug(dbg.data_ptr())
mlp.gemm_tn_grouped(jobs, M)
torch.cuda.synchronize()
_lib.lib().nudf_set_tn_debug(None)
d = dbg.cpu().view(-1, 2, 8)
d = d[d[:, 0, 0] > 0]
t0 = d[:, 0, 0].min()
start = (d[:, 0, 0] - t0).double() / 100.0
end = (d[:, 0, 1] - t0).double() / 100.0
print(f"{len(d)} workgroups; starts {float(start.min()):.1f}..{float(start.max()):.1f} us, ends {float(end.min()):.1f}..{float(end.max()):.1f} us "
      f"(mean {float(end.mean()):.1f}); duration mean {float((end - start).mean()):.1f} min {float((end - start).min()):.1f} max {float((end - start).max()):.1f}")
for w, name in ((0, "wave 0"), (1, "wave 3")):
    nk = (d[:, w, 2] & 0xfffff).double()
    nf = ((d[:, w, 2] >> 20) & 0xfffff).double()
    tot = (d[:, w, 2] >> 40).double()
    seg = [d[:, w, 3 + i].double() for i in range(5)]
    per = [float((s / nf).mean()) for s in seg]
    print(f"{name}: k-steps {float(nk.mean()):.0f} ({float(nf.mean()):.0f} pipelined); ticks per pipelined k-step {sum(per):.0f}: load issue {per[0]:.0f}  "
          f"MFMAs + split {per[1]:.0f}  barrier-1 wait {per[2]:.0f}  LDS stores {per[3]:.0f}  barrier-2 wait {per[4]:.0f}; "
          f"whole loop + prologue {float((tot / nk).mean()):.0f} per k-step   (48 MFMAs = 1536 pipe cycles per wave, two waves share a SIMD)")
ticks = (d[:, 0, 2] >> 40).double()
dur = (end - start)
print(f"shader clock during the kernel: {float((ticks / dur).mean()):.0f} MHz")
