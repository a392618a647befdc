"""GPU diagnostic: localise gradient discrepancies stage by stage against the fp64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import build_modules, perturb_, state_dicts, oracle_nets
from oracle import udf_oracle as O
from neuraludf_amd import synth
from neuraludf_amd.models import fields
from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending, _CompositeFn

dev = torch.device("cuda:0")
mods = perturb_(build_modules(fields, seed=0)); sds = state_dicts(mods)
for m in mods.values(): m.to(dev)
n = 64
r = synth.make_rays(synth.make_scene("tiny"), 0, n, seed=31)
cfg = O.RenderCfg(n_samples=64, n_importance=50, n_outside=0, up_sample_steps=5)
on32 = oracle_nets(sds)
with torch.no_grad():
    ref32 = O.render(on32, cfg, r["rays_o"], r["rays_d"], r["near"], r["far"], cos_anneal_ratio=0.8, flip_saturation=0.9)
z = ref32["z_vals"]; sd = ref32["_sample_dist"]
S = z.shape[1]

def relerr(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))

# ---------------- fp64 oracle with hooks on the intermediates ----------------
torch.set_default_dtype(torch.float64)
on = oracle_nets(sds, requires_grad=True, dtype=torch.float64)
ro, rd, zz = r["rays_o"].double(), r["rays_d"].double(), z.double()
inter = {}
import oracle.udf_oracle as OM
_uf, _ug, _cf = OM.udf_forward, OM.udf_gradient, OM.color_forward
def uf(sd_, x, cfg_=OM.UDFCfg()):
    y = _uf(sd_, x, cfg_); y.retain_grad(); inter["y"] = y; return y
def ug(sd_, x, cfg_=OM.UDFCfg(), create_graph=True):
    g = _ug(sd_, x, cfg_, create_graph); g.retain_grad(); inter["g"] = g; return g
def cf(sd_, p, nn, d, f, cfg_=OM.ColorCfg()):
    cb, c, lg = _cf(sd_, p, nn, d, f, cfg_); cb.retain_grad(); c.retain_grad(); inter["cb"] = cb; inter["c"] = c; return cb, c, lg
OM.udf_forward, OM.udf_gradient, OM.color_forward = uf, ug, cf
o64 = O.render_core(on, cfg, ro, rd, zz, sd, 0.8, None, None, None, 0.9)
def loss_of(o, rgb):
    return ((o["color"] - rgb).abs().mean() + 0.5 * (o["color_base"] - rgb).abs().mean()
            + 0.1 * o["gradient_error"] + 0.01 * o["gradient_error_near_surface"] + 0.001 * o["sparse_error"])
loss_of(o64, r["true_rgb"].double()).backward()
OM.udf_forward, OM.udf_gradient, OM.color_forward = _uf, _ug, _cf
torch.set_default_dtype(torch.float32)
d_y, d_g, d_cb, d_c = inter["y"].grad, inter["g"].grad, inter["cb"].grad, inter["c"].grad
print("oracle64 upstream norms: d_udf %.3e d_feat %.3e d_g %.3e d_cb %.3e d_c %.3e" % (
    d_y[:, 0].abs().max(), d_y[:, 1:].abs().max(), d_g.abs().max(), d_cb.abs().max(), d_c.abs().max()))

# ---------------- stage A: composite backward alone (oracle intermediates as leaves) ----------------
D = lambda t: t.to(dev)
leaves = [inter["y"][:, 0].detach().float().reshape(n, S), inter["g"].detach().float().reshape(n, S, 3),
          inter["c"].detach().float().reshape(n, S, 3), inter["cb"].detach().float().reshape(n, S, 3)]
dl = [D(t).clone().requires_grad_(True) for t in leaves]
scal = torch.cat([O.inv_s_of(on32.var).reshape(1), O.beta_of(on32.beta).reshape(1), O.gamma_of(on32.beta).reshape(1)]).to(dev).requires_grad_(True)
c = dict(s_nominal=S, cos_anneal=0.8, flip_saturation=0.9, use_norm_grad=False, sparse_scale=25000.0, diagnostics=False)
outs = _CompositeFn.apply(c, D(r["rays_o"]), D(r["rays_d"]), D(z), torch.tensor([sd], device=dev), None, dl[0], dl[1], dl[2], dl[3], None, None, None, scal)
color, cbase, weights, depth, normals, wsum, wall, sums = outs[:8]
o = dict(color=color, color_base=cbase, gradient_error=sums[0] / (sums[1] + 1e-5), gradient_error_near_surface=sums[2] / (sums[3] + 1e-5), sparse_error=sums[4] / n)
loss_of(o, D(r["true_rgb"])).backward()
print("A composite: fwd color %.2e | d_udf %.3e d_g %.3e d_c %.3e d_cb %.3e" % (
    relerr(color, o64["color"]), relerr(dl[0].grad.reshape(-1), d_y[:, 0]), relerr(dl[1].grad.reshape(-1, 3), d_g),
    relerr(dl[2].grad.reshape(-1, 3), d_c), relerr(dl[3].grad.reshape(-1, 3), d_cb)))
du = (dl[0].grad.reshape(-1).double().cpu() - d_y[:, 0]); i = int(du.abs().argmax())
print("   worst d_udf idx", i, "sample", i % S, "ours", float(dl[0].grad.reshape(-1)[i]), "ref", float(d_y[i, 0]), "udf", float(leaves[0].reshape(-1)[i]))

# ---------------- stage B: UDF engine backward given the oracle's upstream gradients ----------------
net = mods["udf"]; net.zero_grad()
pts = D(o64["_pts"].float())
udf, feat, grad = net.evaluate(pts, want_grad=True)
print("B udf fwd: udf %.2e feat %.2e grad %.2e" % (relerr(udf, inter["y"][:, 0]), relerr(feat[:, :256], inter["y"][:, 1:]), relerr(grad, inter["g"])))
((udf * D(d_y[:, 0].float())).sum() + (feat[:, :256] * D(d_y[:, 1:].float())).sum() + (grad * D(d_g.float())).sum()).backward()
worst = sorted(((relerr(p.grad, on.udf[nme].grad), nme) for nme, p in net.named_parameters()), reverse=True)[:6]
print("B udf param grads given oracle upstream (rel to max|ref|):", worst)
# split: only d_g path / only d_y path
for tag, a, b in [("only d_g", 0.0, 1.0), ("only d_y", 1.0, 0.0)]:
    net.zero_grad()
    for t in on.udf.values(): t.grad = None
    torch.set_default_dtype(torch.float64)
    yy = O.udf_forward(on.udf, o64["_pts"].detach()); gg = O.udf_gradient(on.udf, o64["_pts"].detach(), create_graph=True)
    (a * (yy * d_y).sum() + b * (gg * d_g).sum()).backward()
    torch.set_default_dtype(torch.float32)
    udf, feat, grad = net.evaluate(pts, want_grad=True)
    (a * ((udf * D(d_y[:, 0].float())).sum() + (feat[:, :256] * D(d_y[:, 1:].float())).sum()) + b * (grad * D(d_g.float())).sum()).backward()
    worst = sorted(((relerr(p.grad, on.udf[nme].grad), nme) for nme, p in net.named_parameters() if on.udf[nme].grad is not None), reverse=True)[:4]
    print("  ", tag, worst)
