"""debug: intermediate stored buffers of the colour net under the two chain kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from common import build_modules, perturb_
from neuraludf_amd import mlp
from neuraludf_amd.models import fields
dev = torch.device("cuda:0")
mods = perturb_(build_modules(fields, seed=0))
udf = mods["udf"].to(dev); col = mods["color"].to(dev)
eng = udf.engine(); ceng = col.engine()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = 64
g = torch.Generator().manual_seed(0)
x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
rays_d = torch.nn.functional.normalize(torch.randn(P // S, 3, generator=g), dim=-1).to(dev)
d_cb = torch.randn(P, 3, generator=g).to(dev); d_cc = torch.randn(P, 3, generator=g).to(dev)
res = {}
for tile in (64, 128, 64):
    mlp.CHAIN_TILE = tile
    st = eng.forward(x, need_grad_state=True, feat_ld=ceng.cin_ld)
    cb, cc, logits, cst = ceng.forward(st["feat"], rays_d, S, P)
    d_lg = torch.randn(P, logits.shape[1], generator=torch.Generator().manual_seed(5)).to(dev)
    # replicate ColorEngine._backward_chain but keep the intermediates
    import neuraludf_amd.mlp as M
    keep = {}
    orig = M.gemm_tn_grouped
    def spy(jobs, Mrows):
        for i, j in enumerate(jobs):
            keep[f"A{i}"] = j[0][:P].clone(); keep[f"B{i}"] = j[2][:P].clone()
        return orig(jobs, Mrows)
    M.gemm_tn_grouped = spy
    grads, dCIN = ceng.backward(cst, cb, cc, d_cb, d_cc, d_lg)
    M.gemm_tn_grouped = orig
    keep.update(cb=cb.clone(), cc=cc.clone(), dCIN=dCIN[:P, :256].clone())
    for i, t in enumerate(grads): keep[f"g{i}"] = t.clone()
    for i, t in enumerate(cst["HB"]): keep[f"HB{i}"] = t[:P].clone()
    for i, t in enumerate(cst["HV"]): keep[f"HV{i}"] = t[:P].clone()
    if tile in res:
        tag = "64 vs 64(rerun)"; a = res[64]
    elif tile == 128:
        tag = "64 vs 128"; a = res[64]
    else:
        res[tile] = keep; continue
    res.setdefault(tile, keep)
    for k in a:
        d = (a[k] - keep[k]).abs()
        nbad = int((d > 1e-5 * a[k].abs().max()).sum())
        if nbad:
            idx = (d > 1e-5 * a[k].abs().max()).nonzero()[:4].tolist()
            print(tag, k, tuple(a[k].shape), "max diff", float(d.max()), "ref max", float(a[k].abs().max()), "nbad", nbad, "first", idx)
    print(tag, "done", flush=True)
