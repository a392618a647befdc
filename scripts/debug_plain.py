import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import udf_oracle as O
from neuraludf_amd.models import fields as nf
dev = torch.device("cuda:0")
for sq in (False, True):
    for seed in (5, 6):
        case = dict(mode="no_view_dir", d_in=9, multires_view=0, squeeze_out=sq, blending_cand_views=0)
        torch.manual_seed(seed)
        net = nf.RenderingNetwork(d_feature=256, d_out=3, d_hidden=96, n_layers=3, weight_norm=True, **case)
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
        g = torch.Generator().manual_seed(2)
        P = 333
        pts, nrm, dirs = (torch.randn(P, 3, generator=g) for _ in range(3))
        feat = torch.randn(P, 256, generator=g)
        fr = feat.clone().requires_grad_(True)
        color, extra = O.rendering_forward(sd, pts, nrm, dirs, fr, mode="no_view_dir", multires_view=0, squeeze_out=sq)
        w1 = torch.randn(P, 3, generator=g)
        (color * w1).sum().backward()
        net.to(dev)
        fd = feat.to(dev).requires_grad_(True)
        c2 = net(pts.to(dev), nrm.to(dev), dirs.to(dev), fd)
        (c2 * w1.to(dev)).sum().backward()
        print("squeeze", sq, "seed", seed, "color diff", float((c2.cpu() - color).abs().max()), "max", float(color.abs().max()))
        print("  dfeat diff", float((fd.grad.cpu() - fr.grad).abs().max()), "max", float(fr.grad.abs().max()))
        for n, p in net.named_parameters():
            d = (p.grad.cpu() - sd[n].grad).abs()
            print("  ", n, "diff", float(d.max()), "max", float(sd[n].grad.abs().max()), "n_bad", int((d > 1e-4 * sd[n].grad.abs().max()).sum()))
