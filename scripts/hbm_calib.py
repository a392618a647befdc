"""achievable HBM bandwidth on this GPU with plain torch kernels (context for the composite kernel's roofline)."""
import torch
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (404, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.rand(n, device=dev); b = torch.empty_like(a)
    tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum()); tf = t(lambda: b.fill_(1.0))
    print(f"{mb} MB: copy {2 * n * 4 / tc / 1e12:.2f} TB/s (r+w)  read(sum) {n * 4 / tr / 1e12:.2f} TB/s  fill {n * 4 / tf / 1e12:.2f} TB/s")
