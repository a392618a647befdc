"""Build-container helper: copy the parts of the reference that are its CALLER side (runner, dataset loader, conf files)
plus its model / loss classes (only for the checkpoint-interchange check) into oracle/_ref/reference_tree/, so that
scripts/run_reference_runner.py and tests/test_gpu_runner.py can run the unchanged reference runner on a GPU box,
where /root/reference does not exist.  oracle/_ref/ is git-ignored: nothing of this is ever committed.

    python oracle/make_ref_tree.py            # no-op when /root/reference is absent
"""
import os
import shutil
import sys

SRC = "/root/reference"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference_tree")
ITEMS = ["exp_runner_blending.py", "extract_mesh.py", "dataset", "confs", "models", "loss"]


def main():
    if not os.path.isfile(os.path.join(SRC, "exp_runner_blending.py")):
        print("reference tree not present: nothing to do")
        return 0
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    os.makedirs(DST)
    for it in ITEMS:
        s, d = os.path.join(SRC, it), os.path.join(DST, it)
        if os.path.isdir(s):
            shutil.copytree(s, d, ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))
        else:
            shutil.copy2(s, d)
    print("copied", ITEMS, "->", DST)
    return 0


if __name__ == "__main__":
    sys.exit(main())
