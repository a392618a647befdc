"""The UNCHANGED reference runner (exp_runner_blending.py: Runner.__init__, train_udf :253-447, save_checkpoint) driven
through the drop-in on the GPU by scripts/run_reference_runner.py: the reference's own training loop, schedules,
Dataset and torch.optim.Adam, with every network / the renderer / the colour loss on the HIP path.  Needs the reference
tree (/root/reference, or the copy oracle/make_ref_tree.py leaves under oracle/_ref/ for GPU boxes): skipped without."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_ref():
    return any(os.path.isfile(os.path.join(d, "exp_runner_blending.py"))
               for d in ("/root/reference", os.path.join(ROOT, "oracle", "_ref", "reference_tree")))


@pytest.mark.parametrize("conf,iters,batch,ft", [("udf_dtu_blending.conf", 60, 512, False),          # classical + NeRF background
                                                 ("udf_garment_blending.conf", 36, 256, False),     # mix schedule, no background
                                                 ("udf_dtu_blending_ft.conf", 36, 256, True)])      # pixel + patch blending on
def test_reference_runner_trains_through_the_dropin(conf, iters, batch, ft, tmp_path):
    if not _have_ref():
        pytest.skip("reference tree not present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_reference_runner.py"), "--conf", conf, "--iters",
                        str(iters), "--batch", str(batch), "--out", str(tmp_path)] + (["--finetune"] if ft else []),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["status"] == "ok", res
    assert res["checks"]["loss_decreases"] and res["checks"]["checkpoint_written"], res
    assert res.get("checks", {}).get("reference_classes_load_checkpoint", True), res
    print(json.dumps({k: res[k] for k in ("conf", "iterations", "loss_first", "loss_last", "psnr_first", "psnr_last", "checkpoint",
                                         "ms_per_iteration", "shims", "reference_modules_loaded")}))
