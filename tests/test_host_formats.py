"""CPU: the host-side data formats and schedules either side of the hot path -- iteration schedules against the
reference Runner's own methods (golden), conf reader, IDR camera decomposition, checkpoint layout."""
import json
import os
import types

import numpy as np
import pytest
import torch

from neuraludf_amd import checkpoint as ck
from neuraludf_amd import conf as nconf
from neuraludf_amd import schedules as sch
from neuraludf_amd.dataset import cameras as cam

HERE = os.path.dirname(__file__)


def test_schedules_match_reference_runner():
    g = json.load(open(os.path.join(HERE, "golden", "ref_schedules.json")))
    for name, c in g["confs"].items():
        # the golden rows hold the reference's regularization_weights_schedule(), i.e. the runner WITH its
        # --reg_weights_schedule flag; without the flag (the default, and what the *_ft launch scripts use) the loop
        # takes the conf's constants (exp_runner_blending.py:361-365)
        s = sch.Schedules(reg_weights_schedule=True, **c)
        s_off = sch.Schedules(**c)
        assert s_off.reg_weights_schedule is False
        for it in g["steps"]:
            a = s_off.at(it)
            assert a["igr_ns_weight"] == c["igr_ns_weight"] and a["sparse_weight"] == c["sparse_weight"], (name, it)
        for it, row in zip(g["steps"], g["values"][name]):
            if row is None:
                continue
            opt = types.SimpleNamespace(param_groups=[{"lr": -1.0}, {"lr": -1.0}, {"lr": -1.0}])
            s.apply_learning_rates(opt, it)
            assert [g_["lr"] for g_ in opt.param_groups] == pytest.approx(row["lr"], rel=1e-14, abs=0), (name, it)
            a = s.at(it)
            for k in ("cos_anneal_ratio", "flip_saturation", "igr_ns_weight", "sparse_weight"):
                assert a[k] == pytest.approx(row[k], rel=1e-14, abs=0), (name, it, k)
            got = [a["color_base_weight"], a["color_weight"], a["color_pixel_weight"], a["color_patch_weight"]]
            assert got == pytest.approx(row["color_weights"], rel=1e-14, abs=0), (name, it)


CONF_TEXT = """
general {
  base_exp_dir = ./exp/udf/dtu/CASE_NAME/   # trailing comment
  recording = [
    ./,
    ./models,
  ]
}
train { learning_rate = 5e-4, end_iter = 300000
  use_white_bkgd = False
  warm_up_end = 5000 }
color_loss { pixel_loss_type = l1
  h_patch_size = 3 }
model {
  nerf { D = 8, skips = [4], use_viewdirs = True }
  udf_network { bias = 0.5
    udf_type = abs  # square or abs
  }
  "quoted key" : "a string, with = punctuation"
}
model.udf_renderer.n_samples = 64
"""


def test_conf_reader():
    c = nconf.parse_string(CONF_TEXT.replace("CASE_NAME", "scan24"))
    assert c["general.base_exp_dir"] == "./exp/udf/dtu/scan24/"
    assert c["general"]["recording"] == ["./", "./models"]
    assert c.get_float("train.learning_rate") == 5e-4 and isinstance(c["train.end_iter"], int)
    assert c.get_bool("train.use_white_bkgd") is False
    assert c.get_float("train.warm_up_end", default=0.0) == 5000.0 and c.get_float("train.anneal_end", default=0.0) == 0.0
    assert c["model.nerf"] == {"D": 8, "skips": [4], "use_viewdirs": True}
    assert dict(**c["model.udf_network"]) == {"bias": 0.5, "udf_type": "abs"}
    assert c["model"]["quoted key"] == "a string, with = punctuation"
    assert c["model.udf_renderer.n_samples"] == 64
    with pytest.raises(KeyError):
        c.get_int("train.batch_size")
    c["train"]["learning_rate"] = 1e-3          # the runner's command-line overrides, exp_runner_blending.py:48-53
    c["dataset.data_dir"] = "/x"
    assert c.get_float("train.learning_rate") == 1e-3 and c["dataset"]["data_dir"] == "/x"
    with pytest.raises(ValueError):
        nconf.parse_string("a { b = 1")
    with pytest.raises(ValueError):
        nconf.parse_string("a = ${b}")
    # the constructors take the sections as **kwargs
    from neuraludf_amd.schedules import Schedules
    full = nconf.parse_string("train { end_iter = 10, learning_rate = 1, learning_rate_geo = 2, learning_rate_alpha = 0.5 }\n"
                              "color_loss { color_weight = 1.0 }")
    s = Schedules.from_conf(full)
    assert s.fix_geo_end == 500 and s.color_weight == 1.0 and s.same_lr is False


def _rot(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return q * np.sign(np.linalg.det(q))


def test_camera_decomposition_round_trip():
    rng = np.random.default_rng(0)
    for trial in range(20):
        K = np.array([[2892.33 + trial, 0.3 * trial, 823.2], [0, 2883.18, 619.07 - trial], [0, 0, 1.0]])
        R = _rot(rng)
        C = rng.normal(size=3) * 3
        P = K @ np.concatenate([R, (-R @ C)[:, None]], 1) * rng.uniform(0.1, 10) * (1 if trial % 2 else -1) ** 0
        K4, pose = cam.load_K_Rt_from_P(None, P.astype(np.float32))
        np.testing.assert_allclose(K4[:3, :3], K, rtol=2e-4, atol=2e-2)
        np.testing.assert_allclose(pose[:3, :3], R.T, atol=2e-5)
        np.testing.assert_allclose(pose[:3, 3], C, atol=2e-4)
        assert K4[3].tolist() == [0, 0, 0, 1] and pose.dtype == np.float32
        Kd, Rd, ch = cam.decompose_projection_matrix(P)
        assert abs(np.linalg.det(Rd) - 1) < 1e-9 and Kd[0, 0] > 0 and Kd[1, 1] > 0 and abs(Kd[1, 0]) + abs(Kd[2, 0]) + abs(Kd[2, 1]) < 1e-9
        np.testing.assert_allclose(P @ ch, 0, atol=1e-6 * np.abs(P).max())


def test_idr_camera_file(tmp_path):
    rng = np.random.default_rng(1)
    d = {}
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    centres = []
    for i in range(3):
        R, C = _rot(rng), rng.normal(size=3) * 100 + 300
        centres.append(C)
        W = np.eye(4); W[:3, :4] = K @ np.concatenate([R, (-R @ C)[:, None]], 1)
        S = np.eye(4); S[:3, :3] *= 150.0; S[:3, 3] = [300, 310, 290]
        d[f"world_mat_{i}"], d[f"scale_mat_{i}"] = W, S
    np.savez(tmp_path / "cameras.npz", **d)
    intr, poses, scale_mats = cam.load_idr_cameras(np.load(tmp_path / "cameras.npz"), 3, downsample_factor=0.5)
    assert intr.shape == (3, 4, 4) and poses.shape == (3, 4, 4) and intr.dtype == np.float32
    np.testing.assert_allclose(intr[0, :3, :3], np.diag([0.5, 0.5, 1]) @ K, rtol=1e-3, atol=0.05)
    for i in range(3):        # camera centre in the normalised frame = S^-1 C
        np.testing.assert_allclose(poses[i, :3, 3], (centres[i] - [300, 310, 290]) / 150.0, atol=1e-3)
    lo, hi = cam.object_bbox(scale_mats[0], scale_mats[0])
    np.testing.assert_allclose(lo, [-1.01] * 3, atol=1e-6); np.testing.assert_allclose(hi, [1.01] * 3, atol=1e-6)


def test_checkpoint_interchange_with_reference_modules(tmp_path):
    """a checkpoint written from the REFERENCE modules + torch Adam loads into the drop-in modules + FusedAdam
    (and back) with identical tensors; file name / key layout as exp_runner_blending.py:486-498."""
    from refload import have_reference, load_reference
    if not have_reference():
        pytest.skip("reference tree not present")
    from common import build_modules
    from neuraludf_amd.models import fields as nf
    from neuraludf_amd.optim import FusedAdam
    rf, _, _ = load_reference()
    ref = build_modules(rf, seed=0)
    order = ["udf", "var", "color", "beta", "nerf"]

    def groups(m):
        return [{"params": list(m["udf"].parameters()), "lr": 1e-4},
                {"params": list(m["var"].parameters()) + list(m["color"].parameters()) + list(m["beta"].parameters())},
                {"params": list(m["nerf"].parameters())}]
    opt = torch.optim.Adam(groups(ref), lr=5e-4)
    g = torch.Generator().manual_seed(0)
    for k in order:
        for p in ref[k].parameters():
            if p.requires_grad:
                p.grad = torch.randn(p.shape, generator=g) * 1e-2
    opt.step()
    path = ck.save_checkpoint(str(tmp_path), 1234, ref["nerf"], ref["udf"], ref["var"], ref["color"], ref["beta"], opt)
    assert os.path.basename(path) == "ckpt_001234.pth" and ck.latest_checkpoint(str(tmp_path)) == "ckpt_001234.pth"
    raw = torch.load(path)
    assert set(raw) == set(ck.NETWORK_KEYS) | {"optimizer", "iter_step"}
    mine = build_modules(nf, seed=1)
    opt2 = FusedAdam(groups(mine), lr=5e-4)
    it = ck.load_checkpoint(path, mine["nerf"], mine["udf"], mine["var"], mine["color"], mine["beta"], opt2)
    assert it == 1234
    for k in order:
        a, b = ref[k].state_dict(), mine[k].state_dict()
        assert list(a) == list(b)
        for n in a:
            assert torch.equal(a[n], b[n]), (k, n)
    sa, sb = opt.state_dict(), opt2.state_dict()
    assert [g_["params"] for g_ in sa["param_groups"]] == [g_["params"] for g_ in sb["param_groups"]]
    assert [g_["lr"] for g_ in sa["param_groups"]] == [g_["lr"] for g_ in sb["param_groups"]]
    for i in sa["state"]:
        for n in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(sa["state"][i][n], sb["state"][i][n])
        assert float(sa["state"][i]["step"]) == float(sb["state"][i]["step"])
    # and back: written from the drop-in side, read by the reference side
    path2 = ck.save_checkpoint(str(tmp_path), 2000, mine["nerf"], mine["udf"], mine["var"], mine["color"], mine["beta"], opt2)
    ref2 = build_modules(rf, seed=2)
    opt3 = torch.optim.Adam(groups(ref2), lr=5e-4)
    assert ck.load_checkpoint(path2, ref2["nerf"], ref2["udf"], ref2["var"], ref2["color"], ref2["beta"], opt3,
                              is_finetune=True) == 0
    assert torch.equal(ref2["udf"].state_dict()["lin3.weight_v"], ref["udf"].state_dict()["lin3.weight_v"])
    assert ck.latest_checkpoint(str(tmp_path)) == "ckpt_002000.pth"


def test_dropin_aliases_resolve_the_runner_imports():
    """the import lines of exp_runner_blending.py:15-21 resolve to the drop-in classes after dropin.install()
    (in a subprocess: the aliases must not leak into this test session)."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import neuraludf_amd.dropin as d; d.install()\n"
        "from models.fields import ResidualRenderingNetwork, SDFNetwork, UDFNetwork, BetaNetwork\n"
        "from models.fields import SingleVarianceNetwork, NeRF, RenderingNetwork\n"
        "from models.udf_renderer_blending import UDFRendererBlending, extract_fields, extract_gradient_fields\n"
        "from loss.loss import ColorLoss\n"
        "import models.embedder, models.patch_projector, loss.patch_metric\n"
        "assert UDFNetwork.__module__ == 'neuraludf_amd.models.fields', UDFNetwork.__module__\n"
        "assert UDFRendererBlending.__module__ == 'neuraludf_amd.models.udf_renderer_blending'\n"
        "assert ColorLoss.__module__ == 'neuraludf_amd.loss.loss'\n"
        "d.uninstall(); assert 'models.fields' not in sys.modules\n"
        "print('ok')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_reference_runner_imports_through_the_dropin():
    """the UNCHANGED reference runner (exp_runner_blending.py) imports with its `models.*` / `loss.*` lines bound to
    the drop-in classes; its third-party imports that are absent from this image are stubbed.  Build container only."""
    import subprocess
    import sys
    from refload import have_reference
    if not have_reference():
        pytest.skip("reference tree not present")
    root = os.path.dirname(HERE)
    code = f"""
import sys, types
sys.path.insert(0, {root!r}); sys.path.insert(0, '/root/reference')
for name in ["cv2", "trimesh", "pyhocon", "icecream", "termcolor", "h5py", "mcubes", "skimage", "skimage.measure",
             "tensorboard", "torch.utils.tensorboard", "custom_mc", "custom_mc._marching_cubes_lewiner"]:
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["icecream"].ic = lambda *a, **k: None
sys.modules["termcolor"].colored = lambda s, *a, **k: s
sys.modules["pyhocon"].ConfigFactory = object; sys.modules["pyhocon"].HOCONConverter = object
sys.modules["torch.utils.tensorboard"].SummaryWriter = object
sys.modules["custom_mc._marching_cubes_lewiner"].udf_mc_lewiner = None
import neuraludf_amd.dropin as d
d.install()
import exp_runner_blending as r
assert r.__file__.startswith('/root/reference/')
for name in ("ResidualRenderingNetwork", "SDFNetwork", "UDFNetwork", "BetaNetwork", "SingleVarianceNetwork", "NeRF"):
    assert getattr(r, name).__module__ == "neuraludf_amd.models.fields", name
assert r.UDFRendererBlending.__module__ == "neuraludf_amd.models.udf_renderer_blending"
assert r.extract_fields.__module__ == "neuraludf_amd.models.udf_renderer_blending"
assert r.ColorLoss.__module__ == "neuraludf_amd.loss.loss"
assert r.Dataset.__module__ == "dataset.dataset"          # the rest of the reference is untouched
print("ok")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_shipped_confs_construct_the_dropin_objects():
    """every conf file the reference ships parses, and its sections construct the drop-in networks, renderer, loss and
    schedules exactly the way exp_runner_blending.py:64-146 consumes them (CPU construction only; build container)."""
    import contextlib
    import glob
    import io
    from refload import have_reference
    if not have_reference():
        pytest.skip("reference tree not present")
    from neuraludf_amd.loss.loss import ColorLoss
    from neuraludf_amd.models import fields
    from neuraludf_amd.models.udf_renderer_blending import UDFRendererBlending
    files = sorted(glob.glob("/root/reference/confs/*.conf"))
    assert len(files) == 4
    for f in files:
        c = nconf.parse_file(f, case="scan24")
        assert "CASE_NAME" not in c["dataset.data_dir"] and c["general.model_type"] == "udf"
        with contextlib.redirect_stdout(io.StringIO()):
            nerf = fields.NeRF(**c["model.nerf"])
            udf = fields.UDFNetwork(**c["model.udf_network"])
            var = fields.SingleVarianceNetwork(**c["model.variance_network"])
            col = fields.ResidualRenderingNetwork(**c["model.rendering_network"])
            beta = fields.BetaNetwork(**c["model.beta_network"])
            rend = UDFRendererBlending(nerf, udf, var, col, beta, **c["model.udf_renderer"])
            loss = ColorLoss(**c["color_loss"])
        s = sch.Schedules.from_conf(c, is_finetune=f.endswith("_ft.conf"))
        assert rend.n_samples == 64 and rend.n_importance in (50, 80) and loss.h_patch_size in (3, 5)
        assert s.end_iter == c.get_int("train.end_iter") and s.learning_rate == c.get_float("train.learning_rate")
        n_par = sum(p.numel() for m in (nerf, udf, var, col, beta) for p in m.parameters())
        assert n_par == 1291484, n_par          # 1 291 484 floats
        a = s.at(0)
        assert 0.0 <= a["cos_anneal_ratio"] <= 1.0


def test_checkpoint_loader_refuses_arbitrary_pickles(tmp_path):
    """load_checkpoint unpickles tensors, containers and the numpy scalars the reference runner leaves in its optimizer
    state -- nothing else, unless the caller opts in (weights_only=False executes code on load)."""
    import numpy as np
    from neuraludf_amd import checkpoint as C
    ok = tmp_path / "ok.pth"
    torch.save({"w": torch.ones(3), "optimizer": {"param_groups": [{"lr": np.float64(5e-4) * 0.5}], "state": {}},
                "iter_step": 7}, ok)
    d = C._load_ckpt(str(ok), None, False)
    assert d["iter_step"] == 7 and float(d["optimizer"]["param_groups"][0]["lr"]) == 2.5e-4

    class Evil:
        def __reduce__(self):
            return (os.path.join, ("executed", "on", "load"))
    bad = tmp_path / "bad.pth"
    torch.save({"w": torch.ones(3), "x": Evil()}, bad)
    with pytest.raises(RuntimeError, match="allow_pickle"):
        C._load_ckpt(str(bad), None, False)
    assert C._load_ckpt(str(bad), None, True)["x"] == os.path.join("executed", "on", "load")
    # a missing or truncated file is reported as what it is, also with the opt-in set
    with pytest.raises(FileNotFoundError):
        C._load_ckpt(str(tmp_path / "missing.pth"), None, True)
    cut = tmp_path / "cut.pth"
    cut.write_bytes(ok.read_bytes()[:100])
    with pytest.raises(Exception) as ei:
        C._load_ckpt(str(cut), None, False)
    assert "allow_pickle" not in str(ei.value)
