"""Validity check of the f16x2-backward timing probe: train losses with the probe switch against the default (debug aid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from test_gpu_graph import _loop

kw = dict(end_iter=30000, learning_rate=1e-3, learning_rate_geo=2e-4, learning_rate_alpha=0.05, warm_up_end=5.0,
          anneal_end=50.0, fix_geo_end=0, color_base_weight=0.5, color_weight=1.0)
rconf = dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0)
res = {}
for mode in ("base", "probe"):
    if mode == "probe":
        os.environ["NUDF_PROBE_BWD_F16X2"] = "1"
    else:
        os.environ.pop("NUDF_PROBE_BWD_F16X2", None)
    tr, losses, _ = _loop(False, kw, 100, 160, rconf=rconf, batch_size=512)
    ps = torch.cat([p.detach().reshape(-1) for p in list(tr.udf.parameters()) + list(tr.color.parameters())])
    res[mode] = ([float(l) for l in losses], ps.clone())
    print(mode, "finite params", bool(torch.isfinite(ps).all()), "losses", ["%.6f" % float(l) for l in losses[::6]])
a, b = res["base"], res["probe"]
print("max |loss diff|", max(abs(x - y) for x, y in zip(a[0], b[0])), "param diff inf", float((a[1] - b[1]).abs().max()),
      "param movement scale", float(a[1].abs().max()))
