"""Import the read-only reference (/root/reference) with stubs for its unused third-party
imports.  Only available in the build container; GPU boxes do not have the reference, so
everything that uses this module is skipped there (tests) or run ahead of time (goldens)."""
import os
import sys
import types
import warnings

REF_ROOT = "/root/reference"
# git-ignored copy of the reference's runner / models / loss made by oracle/make_ref_tree.py in the build container; it
# travels to the GPU box with the gpurun snapshot (where /root/reference does not exist)
REF_TREE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "reference_tree")


def have_reference(root=None) -> bool:
    return os.path.isdir(os.path.join(root or REF_ROOT, "models"))


def load_reference(root=None):
    """Returns (fields_module, renderer_module, loss_module) of the reference under `root` (default /root/reference)."""
    REF_ROOT = root or globals()["REF_ROOT"]
    if not have_reference(REF_ROOT):
        raise RuntimeError("reference tree not present")
    warnings.filterwarnings("ignore")
    for name in ["mcubes", "icecream", "skimage", "skimage.measure", "termcolor"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["icecream"].ic = lambda *a, **k: None
    sys.modules["termcolor"].colored = lambda s, *a, **k: s
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    # the reference packages are called `models` and `loss`; import them under private
    # names so they cannot shadow / be shadowed by this repo's drop-in packages
    import importlib.util

    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "models" or k.startswith("models.")
             or k == "loss" or k.startswith("loss.")}
    sys.path.insert(0, REF_ROOT)
    try:
        import models.fields as rf            # noqa
        import models.udf_renderer_blending as rr   # noqa
        import models.patch_projector as rp   # noqa
        import loss.loss as rl                # noqa
        mods = (rf, rr, rl)
        ref_mods = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")
                    or k == "loss" or k.startswith("loss.")}
    finally:
        sys.path.remove(REF_ROOT)
        for k in list(sys.modules):
            if k == "models" or k.startswith("models.") or k == "loss" or k.startswith("loss."):
                del sys.modules[k]
        sys.modules.update(saved)
    for k, v in ref_mods.items():
        sys.modules["_nudf_ref_" + k] = v
    return mods
