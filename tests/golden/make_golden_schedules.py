"""Golden values for the iteration schedules, computed by the REFERENCE's own Runner methods (build container only):

    python tests/golden/make_golden_schedules.py  ->  tests/golden/ref_schedules.json

exp_runner_blending.py cannot be imported here (cv2, trimesh, pyhocon, tensorboard, h5py are absent), so the six
schedule methods are lifted out of its source with ``ast`` and executed unmodified on a stand-in ``self``."""
import ast
import json
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
METHODS = ["update_learning_rate", "update_learning_rate_geo", "get_cos_anneal_ratio",
           "regularization_weights_schedule", "get_flip_saturation", "adjust_color_loss_weights"]
CONFS = {
    "dtu": dict(end_iter=300000, learning_rate=5e-4, learning_rate_geo=1e-4, learning_rate_alpha=0.05,
                warm_up_end=5000.0, anneal_end=25000.0, fix_geo_end=500, same_lr=False, igr_ns_weight=0.0,
                sparse_weight=0.0, color_base_weight=0.01, color_weight=1.0, color_pixel_weight=0.0,
                color_patch_weight=0.0, is_finetune=False),
    "dtu_ft": dict(end_iter=50000, learning_rate=5e-4, learning_rate_geo=1e-4, learning_rate_alpha=0.05,
                   warm_up_end=2500.0, anneal_end=0.0, fix_geo_end=500.0, same_lr=True, igr_ns_weight=0.01,
                   sparse_weight=0.01, color_base_weight=0.01, color_weight=1.0, color_pixel_weight=0.1,
                   color_patch_weight=0.1, is_finetune=True),
    "garment": dict(end_iter=300000, learning_rate=5e-4, learning_rate_geo=1e-4, learning_rate_alpha=0.05,
                    warm_up_end=5000.0, anneal_end=25000.0, fix_geo_end=500, same_lr=False, igr_ns_weight=0.01,
                    sparse_weight=0.005, color_base_weight=1.0, color_weight=1.0, color_pixel_weight=0.5,
                    color_patch_weight=0.25, is_finetune=False),
}
STEPS = [0, 1, 499, 500, 2499, 2500, 4999, 5000, 9999, 10000, 15000, 19999, 20000, 25000, 59999, 60000, 90000,
         149999, 150000, 200000, 299999]


def lift_methods():
    src = open("/root/reference/exp_runner_blending.py").read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Runner")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in METHODS]
    assert len(fns) == len(METHODS)
    ns = {"np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "exp_runner_blending.py", "exec"), ns)
    return {m: ns[m] for m in METHODS}


def main():
    fn = lift_methods()
    out = {"steps": STEPS, "confs": CONFS, "values": {}}
    for name, c in CONFS.items():
        rows = []
        for it in STEPS:
            if it >= c["end_iter"]:
                rows.append(None)
                continue
            me = types.SimpleNamespace(**c)
            me.iter_step = it
            me.optimizer = types.SimpleNamespace(param_groups=[{"lr": -1.0}, {"lr": -1.0}, {"lr": -1.0}])
            me.color_loss_func = types.SimpleNamespace(set_color_weights=lambda *a: None)
            if c["same_lr"]:                                    # the loop's branch, exp_runner_blending.py:264-268
                fn["update_learning_rate"](me, start_g_id=0)
            else:
                fn["update_learning_rate"](me, start_g_id=1)
                fn["update_learning_rate_geo"](me)
            ns, sp = fn["regularization_weights_schedule"](me)
            rows.append(dict(lr=[float(g["lr"]) for g in me.optimizer.param_groups],
                             cos_anneal_ratio=float(fn["get_cos_anneal_ratio"](me)),
                             flip_saturation=float(fn["get_flip_saturation"](me)),
                             igr_ns_weight=float(ns), sparse_weight=float(sp),
                             color_weights=[float(x) for x in fn["adjust_color_loss_weights"](me)]))
        out["values"][name] = rows
    with open(os.path.join(HERE, "ref_schedules.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote ref_schedules.json")


if __name__ == "__main__":
    main()
