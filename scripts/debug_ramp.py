"""Print the losses of tests/test_gpu_graph.py's colour-weight-ramp loop with and without the ramp (debug aid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neuraludf_amd import mlp
from test_gpu_graph import _loop

kw = dict(end_iter=30000, learning_rate=1e-3, learning_rate_geo=2e-4, learning_rate_alpha=0.05, warm_up_end=5.0,
          anneal_end=50.0, fix_geo_end=0, color_base_weight=0.5, color_weight=1.0)
for tn in (False, True):
    mlp.TN_F16X2 = tn
    a, la, wa = _loop(False, kw, 9994, 10006)
    _, l0, _ = _loop(False, dict(kw, color_base_weight=0.0), 9994, 10006)
    print("TN_F16X2", tn)
    for i, (x, y) in enumerate(zip(la, l0)):
        print("  it %d  ramp %.9f  frozen %.9f  diff %.3e  w %s" % (9994 + i, float(x), float(y), float(x) - float(y), wa[i][:2]))
