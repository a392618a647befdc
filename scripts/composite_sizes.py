"""kernel-pair (composite + partial sums) durations per size, forward and backward, through the C ABI only, for both
lane layouts of the FULL case (strided = sample i in lane i % 64, the default; blocked = lane owns S/64 consecutive samples,
16-byte vector accesses).  A working set beyond the 256 MB Infinity Cache is what the "HBM" figure needs: 32768 x 256 = 405 / 707 MB,
65536 x 256 = 810 / 1414 MB (forward / backward algorithmic bytes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuraludf_amd._lib import Composite, CompositeGrad, call, lib, ptr
from neuraludf_amd.models.udf_renderer_blending import _fill_composite

dev = torch.device("cuda:0")
for (n, s) in [(512, 128), (8192, 256), (32768, 256), (65536, 256), (65536, 128), (16384, 512), (32768, 146)]:
    g = torch.Generator().manual_seed(0)
    z = torch.sort(torch.rand(n, s, generator=g) * 2 + 1.5, -1)[0].to(dev)
    ro = torch.randn(n, 3, generator=g).to(dev)
    rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    udf = (torch.rand(n, s, generator=g) * 0.3).to(dev)
    grad, col, cb = (torch.randn(n, s, 3, generator=g).to(dev) for _ in range(3))
    scal = torch.tensor([64.0, 128.0, 20.0], device=dev)
    sd = torch.tensor([2.0 / 64], device=dev)
    c = dict(s_nominal=s, cos_anneal=1.0, flip_saturation=1.0, use_norm_grad=False, sparse_scale=25000.0, diagnostics=False)
    a = Composite()
    a.rays_o, a.rays_d, a.z, a.udf, a.grad, a.color, a.color_base = map(ptr, (ro, rd, z, udf, grad, col, cb))
    a.scal, a.sample_dist = ptr(scal), ptr(sd)
    _fill_composite(a, c, n, s, 0)
    bufs = [torch.empty(n, s, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev),
            torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev), torch.empty(5, device=dev)]
    (a.weights, a.out_color, a.out_color_base, a.out_depth, a.out_normals, a.out_wsum, a.out_wsum_all, a.sums) = [ptr(b) for b in bufs]
    ws = torch.empty(5 * ((n + 3) // 4), device=dev); a.ws = ptr(ws)
    gq = CompositeGrad()
    ups = [torch.randn(n, 3, device=dev), torch.randn(n, 3, device=dev), torch.randn(n, s, device=dev), torch.randn(5, device=dev)]
    gq.d_color, gq.d_color_base, gq.d_weights, gq.d_sums = map(ptr, ups)
    outs = [torch.empty(n, s, device=dev), torch.empty(n, s, 3, device=dev), torch.empty(n, s, 3, device=dev),
            torch.empty(n, s, 3, device=dev), torch.empty(3, device=dev)]
    gq.o_d_udf, gq.o_d_grad, gq.o_d_color, gq.o_d_color_base, gq.o_d_scal = map(ptr, outs)
    ws2 = torch.empty(3 * ((n + 3) // 4), device=dev); gq.ws = ptr(ws2)

    def t(fn, reps=40):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    bf, bb = 48.0 * n * s + 68.0 * n, 84.0 * n * s + 68.0 * n
    for layout in ("blocked", "strided"):
        lib().nudf_set_composite_blocked(1 if layout == "blocked" else 0)
        tf = t(lambda: call("nudf_composite_fwd", a))
        tb = t(lambda: call("nudf_composite_bwd", a, gq))
        print(f"{n}x{s} {layout:8s}: fwd {tf:7.1f} us {bf / tf / 1e6:.2f} TB/s ({bf / tf / 8e6 * 100:.1f}% of 8 TB/s, {bf / 1e6:.0f} MB) | "
              f"bwd {tb:7.1f} us {bb / tb / 1e6:.2f} TB/s ({bb / tb / 8e6 * 100:.1f}%, {bb / 1e6:.0f} MB)", flush=True)
    lib().nudf_set_composite_blocked(0)
    del z, udf, grad, col, cb, bufs, ups, outs
    torch.cuda.empty_cache()
