"""GPU check + timing of the fused UDF chains against the per-layer GEMM path (same weights, same points)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from common import build_modules, perturb_
from neuraludf_amd import mlp
from neuraludf_amd.models import fields

dev = torch.device("cuda:0")
mods = perturb_(build_modules(fields, seed=0))
udf = mods["udf"].to(dev)
eng = udf.engine()


def run(P, chain, seed=0):
    mlp.USE_CHAIN = chain
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
    d_udf = torch.randn(P, generator=g).to(dev)
    d_feat = torch.randn(P, 288, generator=g).to(dev)
    d_g = torch.randn(P, 3, generator=g).to(dev)
    st = eng.forward(x, need_grad_state=True, feat_ld=288)
    gr, DA = eng.gradient(x, st)
    grads = eng.backward(x, st, DA, d_udf, d_feat, 288, d_g)
    uo = eng.forward(x, need_grad_state=False, udf_only=True)["udf"]
    return dict(udf=st["udf"], sign=st["sign"], feat=st["feat"][:, :256], g=gr, uo=uo, X4=st["X"][4][:P], X8=st["X"][8][:P],
                DA0=DA[0][:P], DA3=DA[3][:P, :217], DA7=DA[7][:P], **{f"p{i}": t for i, t in enumerate(grads)})


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-20))


for P in (1000, 64 * 300 + 7):
    a = run(P, False)
    b = run(P, True)
    worst = 0
    for k in a:
        r = rel(b[k], a[k])
        worst = max(worst, r)
        if r > 1e-4:
            print("MISMATCH", P, k, r)
    print("P", P, "worst rel", worst)

# timing
for P in (8192, 32768, 65536):
    for chain in (False, True):
        mlp.USE_CHAIN = chain
        x = (torch.rand(P, 3) * 2 - 1).to(dev)
        d_udf = torch.randn(P).to(dev); d_feat = torch.randn(P, 288).to(dev); d_g = torch.randn(P, 3).to(dev)
        def t(fn, n=10):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        st = eng.forward(x, True, 288)
        gr, DA = eng.gradient(x, st)
        r = dict(
            fwd_udf_only=t(lambda: eng.forward(x, False, udf_only=True)),
            fwd_state=t(lambda: eng.forward(x, True, 288)),
            grad=t(lambda: eng.gradient(x, st)),
            bwd=t(lambda: eng.backward(x, st, DA, d_udf, d_feat, 288, d_g)),
        )
        fl = dict(fwd_udf_only=2 * P * (39 * 256 + 256 * 256 * 2 + 256 * 217 + 256 * 256 * 4 + 256), fwd_state=1049088.0 * P,
                  grad=918016.0 * P, bwd=(1049088.0 + 918016.0 + 2 * 1049088.0) * P)
        print(P, "chain" if chain else "layers", {k: f"{v:.0f}us {fl[k] / v / 1e6:.1f}TF" for k, v in r.items()})
