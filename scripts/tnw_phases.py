"""where does a k-step of the wide bf16x3 weight-gradient kernel go? (nudf_set_tn_debug: per-segment shader-clock ticks of
waves 0 / 4 of every workgroup) -- UDF adjoint group at M points"""
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
for flags, tag in ((1024, "wide"), (0, "128x128")):
    _lib.lib().nudf_set_tn_flags(flags)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    mlp.gemm_tn_grouped(jobs, M)
    e0.record()
    for _ in range(10):
        mlp.gemm_tn_grouped(jobs, M)
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) * 100:.1f} us per launch")
_lib.lib().nudf_set_tn_flags(1024)
dbg = torch.zeros(8 * 1024, dtype=torch.int64, device=dev)
_lib.lib().nudf_set_tn_debug(dbg.data_ptr())
mlp.gemm_tn_grouped(jobs, M)
torch.cuda.synchronize()
_lib.lib().nudf_set_tn_debug(None)
d = dbg.cpu().view(-1, 2, 4)
d = d[d[:, 0, 0] > 0]
for w, name in ((0, "wave 0 (matrix first)"), (1, "wave 4 (staging first)")):
    tot = (d[:, w, 0] & ((1 << 40) - 1)).double()
    nk = (d[:, w, 0] >> 40).double()
    mma, store, load, bar = d[:, w, 1].double(), (d[:, w, 2] & 0xffffffff).double(), (d[:, w, 2] >> 32).double(), d[:, w, 3].double()
    print(f"{name}: {len(d)} workgroups, k-steps {nk.mean():.0f}; ticks per k-step: total {float((tot / nk).mean()):.0f}  "
          f"MFMA segment {float((mma / nk).mean()):.0f}  split+LDS stores {float((store / nk).mean()):.0f}  "
          f"load issue {float((load / nk).mean()):.0f}  barrier wait {float((bar / nk).mean()):.0f}   (48 MFMAs = 1536 pipe cycles per wave)")
