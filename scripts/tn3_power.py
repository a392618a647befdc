"""is the bf16x3 weight-gradient kernel clock / power limited?  The same launch (UDF adjoint group, M = 65 536) on operands of
different bit activity: zeros, one constant, small-range uniform, standard normal; kernel time from the workgroups' own wall-clock
stamps (nudf_set_tn_debug) and the shader clock that s_memtime saw.  Needs the stamp build (NUDF_LIB=.../libnudf_stamps.so)."""
import os, sys, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuraludf_amd import _lib, mlp
dev = torch.device("cuda:0")
M = 65536
SHAPES = [(256, 40)] + [(256, 256)] * 3 + [(217, 256)] + [(256, 256)] * 3 + [(256, 256), (1, 256)]
mlp.set_precision("bf16x3")
torch.manual_seed(0)


def fill(kind, *shape):
    if kind == "zeros":
        return torch.zeros(*shape, device=dev)
    if kind == "ones":
        return torch.ones(*shape, device=dev)
    if kind == "bf16_exact":      # values with an exact bf16 representation: the mid / lo planes are zero
        return torch.randn(*shape, device=dev).to(torch.bfloat16).float()
    if kind == "softplus":        # what the chains store: softplus(100 x) / 100-like activations, small deltas
        return torch.nn.functional.softplus(torch.randn(*shape, device=dev))
    return torch.randn(*shape, device=dev)


samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out)
        except Exception as e:
            samples.append(str(e))
        time.sleep(0.05)


for kind in ("zeros", "ones", "bf16_exact", "softplus", "randn"):
    jobs = []
    for NA, NB in SHAPES:
        lda = max(4, (NA + 3) // 4 * 4)
        jobs.append((fill(kind, M, lda), NA, fill(kind, M, NB), NB,
                     torch.zeros((NA + 31) // 32 * 32, NB, device=dev), torch.zeros((NA + 31) // 32 * 32, device=dev)))
    for _ in range(20):
        mlp.gemm_tn_grouped(jobs, M)
    torch.cuda.synchronize()
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
    ticks = (d[:, 0, 2] >> 40).double()
    nf = ((d[:, 0, 2] >> 20) & 0xfffff).double()
    seg = [float((d[:, 0, 3 + i].double() / nf).mean()) for i in range(5)]
    print(f"{kind:>10}: kernel {float(end.max()):.1f} us (workgroup mean {float((end - start).mean()):.1f}), shader clock {float((ticks / (end - start)).mean()):.0f} MHz, "
          f"ticks per k-step {sum(seg):.0f} = load {seg[0]:.0f} + mfma/split {seg[1]:.0f} + bar {seg[2]:.0f} + stores {seg[3]:.0f} + bar {seg[4]:.0f}")
    del jobs
print(subprocess.run(["rocm-smi", "--showmaxpower", "--showpower", "--showclocks"], capture_output=True, text=True).stdout[-1500:])
# power while the randn launch loops for ~1.5 s
jobs = []
for NA, NB in SHAPES:
    lda = max(4, (NA + 3) // 4 * 4)
    jobs.append((fill("randn", M, lda), NA, fill("randn", M, NB), NB,
                 torch.zeros((NA + 31) // 32 * 32, NB, device=dev), torch.zeros((NA + 31) // 32 * 32, device=dev)))
th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50):
        mlp.gemm_tn_grouped(jobs, M)
    torch.cuda.synchronize()
stop = True
th.join()
print("samples under load:", len(samples))
for s in samples[-3:]:
    print(s.strip()[-600:])
