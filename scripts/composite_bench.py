"""time nudf_composite_fwd / _bwd alone on resident inputs (HBM roofline of the fused sample+composite kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuraludf_amd.models.udf_renderer_blending import _CompositeFn

dev = torch.device("cuda:0")
for (n, s) in [(512, 128), (8192, 256), (32768, 256)]:
    g = torch.Generator().manual_seed(0)
    z = torch.sort(torch.rand(n, s, generator=g) * 2 + 1.5, -1)[0].to(dev)
    ro = torch.randn(n, 3, generator=g).to(dev)
    rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    udf = (torch.rand(n, s, generator=g) * 0.3).to(dev).requires_grad_(True)
    grad = torch.randn(n, s, 3, generator=g).to(dev).requires_grad_(True)
    col = torch.rand(n, s, 3, generator=g).to(dev).requires_grad_(True)
    cb = torch.rand(n, s, 3, generator=g).to(dev).requires_grad_(True)
    scal = torch.tensor([64.0, 128.0, 20.0], device=dev)
    sd = torch.tensor([2.0 / 64], device=dev)
    c = dict(s_nominal=s, cos_anneal=1.0, flip_saturation=1.0, use_norm_grad=False, sparse_scale=25000.0, diagnostics=False)
    outs = _CompositeFn.apply(c, ro, rd, z, sd, None, udf, grad, col, cb, None, None, None, scal)
    loss = outs[0].sum() + outs[1].sum() + 0.1 * outs[7].sum()
    def fwd():
        with torch.no_grad():
            _CompositeFn.apply(c, ro, rd, z, sd, None, udf, grad, col, cb, None, None, None, scal)
    def t(fn, reps=30):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    tf = t(fwd)
    tb = t(lambda: torch.autograd.grad(loss, [udf, grad, col, cb], retain_graph=True))
    bf = 48.0 * n * s + 68.0 * n
    bb = 84.0 * n * s + 68.0 * n
    print(f"{n}x{s}: fwd(+alloc) {tf:.1f} us {bf / tf / 1e3:.0f} GB/s ({bf / tf / 8e6 * 100:.1f}% of 8 TB/s) | bwd(+alloc) {tb:.1f} us {bb / tb / 1e3:.0f} GB/s ({bb / tb / 8e6 * 100:.1f}%)")
