"""Drive the UNCHANGED reference runner (exp_runner_blending.py: Runner.__init__ :31-165, train_udf :253-447,
save_checkpoint :486-498) through the drop-in on the GPU:

    python scripts/run_reference_runner.py [--ref DIR] [--conf udf_dtu_blending.conf] [--iters 60] [--batch 512]
                                           [--out gpurun_out/runner]

What runs from where
  * reference, untouched:  exp_runner_blending.py (Runner, the whole training loop, schedules, checkpointing),
    dataset/dataset.py (Dataset: camera files, image files, per-iteration ray batches), the shipped conf file;
  * drop-in (neuraludf_amd.dropin): its `models.*` / `loss.*` import lines resolve to this repo's classes, i.e. every
    network, the renderer and the colour loss are the HIP path; the optimizer is the reference's torch.optim.Adam;
  * stand-ins for third-party packages that are absent from this image (scripts/refshim: cv2 -> Pillow + the
    OpenCV-free camera decomposition, pyhocon -> neuraludf_amd.conf, tensorboard -> an in-memory scalar log).

The reference tree is looked for at --ref, /root/reference, then oracle/_ref/reference_tree (a copy that
oracle/make_ref_tree.py makes in the build container so that it travels to the GPU box; never committed).
The IDR-format case directory (cameras.npz, image/*.png, mask/*.png) is synthetic: views of a TEACHER network
rendered with the drop-in, so the run has something to learn.  Schedule constants of the shipped conf that are
counted in iterations are rescaled to the short run; nothing else of the conf changes.

Checks (exit code 1 if any fails): the loss decreases, a ckpt_*.pth is written, and -- when the reference's own model
classes are importable -- its state dicts load into the reference's UDFNetwork / ResidualRenderingNetwork / NeRF.
Prints one JSON line and keeps the log under --out."""
import argparse
import json
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find_reference(arg):
    for d in (arg, "/root/reference", os.path.join(ROOT, "oracle", "_ref", "reference_tree")):
        if d and os.path.isfile(os.path.join(d, "exp_runner_blending.py")):
            return os.path.abspath(d)
    return None


def install_shims(scalars):
    """third-party modules the reference imports: the real one if present, else scripts/refshim / a placeholder."""
    import importlib
    shim_dir = os.path.join(ROOT, "scripts", "refshim")
    used = []
    for name in ("cv2", "pyhocon", "icecream", "termcolor", "trimesh", "h5py"):
        try:
            importlib.import_module(name)
        except ImportError:
            if shim_dir not in sys.path:
                sys.path.append(shim_dir)
            importlib.import_module(name)
            used.append(name)
    for name in ("mcubes", "skimage", "skimage.measure", "custom_mc", "custom_mc._marching_cubes_lewiner"):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
            used.append(name)
    if not hasattr(sys.modules["custom_mc._marching_cubes_lewiner"], "udf_mc_lewiner"):
        sys.modules["custom_mc._marching_cubes_lewiner"].udf_mc_lewiner = None      # mesh extraction is not reached
    try:
        from torch.utils.tensorboard import SummaryWriter  # noqa: F401
    except Exception:
        class SummaryWriter:                     # in-memory scalar log with the two methods the runner calls
            def __init__(self, log_dir=None, **kw):
                self.log_dir = log_dir

            def add_scalar(self, tag, value, step=None):
                scalars.setdefault(tag, []).append((int(step), float(value)))

            def add_image(self, *a, **k):
                pass

            def close(self):
                pass
        for name in ("tensorboard", "torch.utils.tensorboard"):
            sys.modules[name] = types.ModuleType(name)
        sys.modules["torch.utils.tensorboard"].SummaryWriter = SummaryWriter
        import torch.utils
        torch.utils.tensorboard = sys.modules["torch.utils.tensorboard"]
        used.append("torch.utils.tensorboard")
    return used


def write_case(case_dir, dev, n_views, rconf, camera_files=("cameras.npz",)):
    """IDR layout (dataset/dataset.py:59-71): cameras.npz with world_mat_i / scale_mat_i, image/%03d.png, mask/%03d.png.
    The images are renderings of a teacher network (drop-in path), stored in cv2's BGR byte order."""
    import numpy as np
    import torch
    from PIL import Image
    from neuraludf_amd import synth
    from neuraludf_amd.dataset import RayBatchSource
    from neuraludf_amd.train import Trainer
    scene = synth.make_scene("tiny")
    os.makedirs(os.path.join(case_dir, "image"), exist_ok=True)
    os.makedirs(os.path.join(case_dir, "mask"), exist_ok=True)
    dummy = torch.zeros(n_views, scene.H, scene.W, 3)
    src = RayBatchSource(dummy, torch.ones_like(dummy), scene.intrinsics[:n_views], scene.c2w[:n_views], device=dev)
    teacher = Trainer(dev, rconf, seed=1)
    with torch.no_grad():
        for p in teacher.color.parameters():
            p.mul_(1.5)
    cams = {}
    for i in range(n_views):
        with torch.no_grad():
            img = teacher.render_image(src, i, resolution_level=1)["color"].clamp(0, 1)        # [H,W,3] in dataset order
        a = (img.cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
        Image.fromarray(np.ascontiguousarray(a[:, :, ::-1])).save(os.path.join(case_dir, "image", "%03d.png" % i))
        Image.fromarray(np.full((scene.H, scene.W, 3), 255, np.uint8)).save(os.path.join(case_dir, "mask", "%03d.png" % i))
        K = scene.intrinsics[i].double().numpy()
        w2c = np.linalg.inv(scene.c2w[i].double().numpy())
        cams["world_mat_%d" % i] = (K @ w2c).astype(np.float64)
        cams["scale_mat_%d" % i] = np.eye(4)
    for name in set(camera_files):           # dataset.render_cameras_name / object_cameras_name of the conf
        np.savez(os.path.join(case_dir, name), **cams)
    del teacher
    return scene


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default=None)
    ap.add_argument("--conf", default="udf_dtu_blending.conf")
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--finetune", action="store_true", help="--is_finetune of the reference CLI: colour-loss weights not ramped, i.e. the pixel / patch blending terms of the *_ft confs are on from iteration 0")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "runner"))
    a = ap.parse_args()
    ref = find_reference(a.ref)
    if ref is None:
        print(json.dumps({"status": "skipped", "why": "reference tree not found"}))
        return 0
    import torch
    assert torch.cuda.is_available(), "needs the GPU"
    dev = torch.device("cuda:0")
    os.makedirs(a.out, exist_ok=True)
    work = tempfile.mkdtemp(prefix="nudf_runner_")
    scalars = {}
    shims = install_shims(scalars)

    # ---- the case directory and the conf -----------------------------------------------------------------
    conf_text = open(os.path.join(ref, "confs", a.conf)).read()
    import neuraludf_amd.conf as nconf
    shipped = nconf.parse_string(conf_text.replace("CASE_NAME", "synth"))
    rconf = dict(shipped["model.udf_renderer"])
    rconf.pop("sdf2alpha_type", None)
    case_dir = os.path.join(work, "data", "synth")
    write_case(case_dir, dev, a.views, {k: rconf[k] for k in ("n_samples", "n_importance", "n_outside", "up_sample_steps", "perturb")},
               camera_files=(shipped.get_string("dataset.render_cameras_name"), shipped.get_string("dataset.object_cameras_name")))
    import re
    n = a.iters

    def setkey(text, key, value):
        new, k = re.subn(r"(?m)^(\s*%s\s*=\s*).*$" % re.escape(key), lambda m: m.group(1) + str(value), text, count=1)
        assert k == 1, key
        return new
    for key, value in (("data_dir", os.path.join(work, "data", "CASE_NAME") + "/"), ("base_exp_dir", os.path.join(work, "exp", "CASE_NAME") + "/"),
                       ("end_iter", n), ("batch_size", a.batch), ("warm_up_end", max(2, n // 12)), ("anneal_end", max(4, n // 3)),
                       ("save_freq", n), ("val_freq", 10 ** 9), ("val_mesh_freq", 10 ** 9), ("report_freq", max(1, n // 3))):
        conf_text = setkey(conf_text, key, value)
    conf_text = re.sub(r"recording\s*=\s*\[[^\]]*\]", "recording = [\n    ./\n  ]", conf_text, count=1)
    # the geometry learning rate is zero until train.fix_geo_end (runner default 500 iterations): rescaled like the others
    conf_text = conf_text.replace("train {", "train {\n  fix_geo_end = %d" % max(1, n // 12), 1)
    conf_path = os.path.join(work, "run.conf")
    open(conf_path, "w").write(conf_text)

    # ---- the reference's own entry sequence (exp_runner_blending.py:868-901), mode 'train' ----------------------
    import neuraludf_amd.dropin as dropin
    dropin.install()
    sys.path.insert(0, ref)
    os.chdir(work)
    torch.set_default_tensor_type("torch.cuda.FloatTensor")                  # :872
    torch.manual_seed(0)
    import exp_runner_blending as R
    assert os.path.abspath(R.__file__).startswith(ref), R.__file__
    assert R.UDFRendererBlending.__module__ == "neuraludf_amd.models.udf_renderer_blending"
    assert R.Dataset.__module__ == "dataset.dataset" and os.path.abspath(sys.modules["dataset.dataset"].__file__).startswith(ref)
    args = argparse.Namespace(conf=conf_path, mode="train", model_type="", threshold=0.005, is_continue=False, is_finetune=bool(a.finetune),
                              reg_weights_schedule=False, vis_ray=False, gpu=0, resolution=128, case="synth", learning_rate=0,
                              learning_rate_geo=0, sparse_weight=0)
    R.args = args            # train_udf reads the module-level `args` (:437) only on the mesh-validation branch
    t0 = time.time()
    runner = R.Runner(args.conf, args.mode, args.case, args.model_type, args.is_continue, args)
    t_init = time.time() - t0
    t0 = time.time()
    runner.train()
    torch.cuda.synchronize()
    t_train = time.time() - t0
    torch.set_default_tensor_type("torch.FloatTensor")

    # ---- checks ----------------------------------------------------------------------------------------------
    loss = [v for _, v in scalars.get("Loss/loss", [])]
    psnr = [v for _, v in scalars.get("Sta/psnr", [])]
    k = max(3, n // 10)
    ok = {}
    ok["iterations_ran"] = runner.iter_step == n and len(loss) in (0, n)
    if loss:
        first, last = sum(loss[:k]) / k, sum(loss[-k:]) / k
        ok["loss_decreases"] = last < 0.9 * first and all(x == x for x in loss)
    else:
        first = last = None
    ck_dir = os.path.join(runner.base_exp_dir, "checkpoints")
    cks = sorted(f for f in os.listdir(ck_dir) if f.endswith(".pth")) if os.path.isdir(ck_dir) else []
    ok["checkpoint_written"] = len(cks) > 0
    loaded = None
    if cks:
        # weights_only=False: the runner stores numpy floats (its learning rates, update_learning_rate :167-177) in the optimizer state
        ck = torch.load(os.path.join(ck_dir, cks[-1]), map_location="cpu", weights_only=False)
        ok["checkpoint_keys"] = set(["nerf", "udf_network_fine", "variance_network_fine", "color_network_fine", "optimizer", "iter_step"]) <= set(ck)
        if os.path.isdir(os.path.join(ref, "models")):
            # the REFERENCE's own classes (private import, not the drop-in) load the checkpoint the runner wrote
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            try:
                import refload
                refload.REF_ROOT = ref
                rf, _, _ = refload.load_reference()
                conf = runner.conf
                import contextlib
                import io
                import inspect
                dropped = {}

                def build(cls, section):
                    # the garment confs pass `udf_shift` / `predict_grad`, which the reference's own UDFNetwork does not
                    # accept (models/fields.py:116-128): keys outside the constructor's signature are left out and reported
                    kw = dict(conf[section])
                    ok_keys = set(inspect.signature(cls.__init__).parameters)
                    extra = sorted(k for k in kw if k not in ok_keys)
                    if extra:
                        dropped[section] = extra
                    return cls(**{k: v for k, v in kw.items() if k in ok_keys})
                with contextlib.redirect_stdout(io.StringIO()):
                    m = {"udf_network_fine": build(rf.UDFNetwork, "model.udf_network"),
                         "color_network_fine": build(rf.ResidualRenderingNetwork, "model.rendering_network"),
                         "nerf": build(rf.NeRF, "model.nerf"),
                         "variance_network_fine": build(rf.SingleVarianceNetwork, "model.variance_network")}
                for name, mod in m.items():
                    mod.load_state_dict(ck[name])
                loaded = sorted(m) + ([{"conf_keys_the_reference_classes_reject": dropped}] if dropped else [])
                ok["reference_classes_load_checkpoint"] = True
            except Exception as e:           # pragma: no cover
                ok["reference_classes_load_checkpoint"] = False
                loaded = repr(e)
    res = {"status": "ok" if all(ok.values()) else "failed", "checks": ok, "reference": ref, "conf": a.conf, "iterations": n,
           "batch_size": a.batch, "shims": shims, "loss_first": first, "loss_last": last,
           "psnr_first": (sum(psnr[:k]) / k) if psnr else None, "psnr_last": (sum(psnr[-k:]) / k) if psnr else None,
           "checkpoint": cks[-1] if cks else None, "reference_modules_loaded": loaded, "init_s": round(t_init, 2),
           "train_s": round(t_train, 2), "ms_per_iteration": round(1e3 * t_train / max(n, 1), 2),
           "renderer": dict(rconf), "n_params": sum(p.numel() for g in runner.optimizer.param_groups for p in g["params"])}
    res["is_finetune"] = bool(a.finetune)
    with open(os.path.join(a.out, "runner_%s.json" % os.path.splitext(a.conf)[0]), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))
    return 0 if res["status"] == "ok" else 1


if __name__ == "__main__":
    sys.exit(main())
