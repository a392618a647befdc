"""which parameter element carries the largest HIP-vs-oracle difference after 5 Adam steps (tests/test_gpu_train_parity.py),
and what its gradient looked like on the oracle side step by step"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import udf_oracle as O
from neuraludf_amd import synth, mlp
from neuraludf_amd.train import Trainer

n_outside = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
rconf = dict(n_samples=32, n_importance=0, n_outside=n_outside, up_sample_steps=1, perturb=0.0)
tr = Trainer(dev, rconf, seed=0, fused_adam=True)
tr.renderer.diagnostics = False
rays = synth.make_rays(synth.make_scene("tiny"), 0, 64, seed=21)
batch = {k: v.to(dev) for k, v in rays.items()}
sds = {k: {n: t.detach().cpu().clone() for n, t in m.state_dict().items()} for k, m in tr.modules().items()}
nets = O.Nets(**{k: {n: t.clone().requires_grad_(True) for n, t in sds[k].items()} for k in ("udf", "color", "var", "beta", "nerf")})
nets.beta["gamma"].requires_grad_(False); nets.beta["zeta"].requires_grad_(False)
geo = list(nets.udf.values()); other = list(nets.var.values()) + list(nets.color.values()) + [nets.beta["beta"]]; nerf = list(nets.nerf.values())
opt = torch.optim.Adam([{"params": geo, "lr": 1e-4}, {"params": other}, {"params": nerf}], lr=5e-4)
cfg = O.RenderCfg(n_samples=32, n_importance=0, n_outside=n_outside, up_sample_steps=1)
hist, ghist = [], []
for it in range(5):
    out = O.render(nets, cfg, rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], cos_anneal_ratio=1.0, flip_saturation=1.0)
    cl = O.color_loss(0.01, 1.0, 0.0, 0.0, 3, out["color_base"], out["color"], rays["true_rgb"], None, None, None, None, None)
    loss = cl["loss"] + 0.1 * out["gradient_error"]
    opt.zero_grad(); loss.backward()
    ghist.append({(k, n): (t.grad.clone() if t.grad is not None else None) for k in ("udf", "color", "var", "beta", "nerf") for n, t in getattr(nets, k).items()})
    opt.step()
    tr.optimizer.zero_grad(set_to_none=True)
    l, _ = tr.loss(batch, cos_anneal_ratio=1.0, flip_saturation=1.0, perturb_overwrite=0)
    l.backward()
    hist.append({(k, n): p.grad.detach().cpu().clone() for k, m in tr.modules().items() for n, p in m.named_parameters() if p.grad is not None})
    tr.optimizer.step()
rows = []
for k, m in tr.modules().items():
    for n, p in m.state_dict().items():
        ref = getattr(nets, k)[n].detach()
        d = (p.detach().cpu() - ref).abs()
        rows.append((float(d.max()), k, n, int(d.argmax())))
rows.sort(reverse=True)
print("precision", mlp.PRECISION, "fwd split", mlp.FWD_F16X2)
for w, k, n, idx in rows[:6]:
    print(f"{k}.{n}[{idx}]: |dw| {w:.3e}")
    for it in range(5):
        go = ghist[it][(k, n)]
        gh = hist[it].get((k, n))
        a = float(go.reshape(-1)[idx]) if go is not None else None
        b = float(gh.reshape(-1)[idx]) if gh is not None else None
        print(f"     step {it}: oracle grad {a:.4e}  hip grad {b:.4e}   tensor max|g| {float(go.abs().max()):.3e}")
# aggregate view: the movement of every network over the 5 steps, HIP against oracle
for k, m in tr.modules().items():
    num = den = 0.0
    mx = 0.0
    for n, p in m.state_dict().items():
        ref = getattr(nets, k)[n].detach().double()
        w0 = sds[k][n].double()
        mv_h, mv_r = p.detach().cpu().double() - w0, ref - w0
        num += float((mv_h - mv_r).pow(2).sum()); den += float(mv_r.pow(2).sum())
        mx = max(mx, float((mv_h - mv_r).abs().max()))
    if den > 0:
        print(f"movement {k}: ||dmove||_2 / ||move_ref||_2 = {(num / den) ** 0.5:.3e}   max |dw| {mx:.3e}   ||move_ref||_2 {den ** 0.5:.3e}")
