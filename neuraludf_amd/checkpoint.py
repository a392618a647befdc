"""Checkpoint files in the reference's layout (exp_runner_blending.py:467-498): one ``torch.save`` dict with the five
network ``state_dict``s, the optimizer state and ``iter_step``, named ``ckpt_{iter:06d}.pth``.  The drop-in modules keep
the reference's parameter names, shapes and order, so files written by either side load in the other."""
from __future__ import annotations

import os

import torch

NETWORK_KEYS = ("nerf", "udf_network_fine", "variance_network_fine", "color_network_fine", "beta_network")


def checkpoint_name(iter_step: int) -> str:
    return "ckpt_{:0>6d}.pth".format(iter_step)


def save_checkpoint(base_exp_dir, iter_step, nerf, udf_network, variance_network, color_network, beta_network,
                    optimizer):
    ckpt = {"nerf": nerf.state_dict(), "udf_network_fine": udf_network.state_dict(),
            "variance_network_fine": variance_network.state_dict(), "color_network_fine": color_network.state_dict(),
            "beta_network": beta_network.state_dict(), "optimizer": optimizer.state_dict(), "iter_step": iter_step}
    d = os.path.join(base_exp_dir, "checkpoints")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, checkpoint_name(iter_step))
    torch.save(ckpt, path)
    return path


def latest_checkpoint(base_exp_dir, end_iter=None):
    """exp_runner_blending.py:148-158: the last ``*.pth`` name in sorted order (the reference's ``<= end_iter`` filter
    is commented out there; pass ``end_iter`` to apply it)."""
    d = os.path.join(base_exp_dir, "checkpoints")
    if not os.path.isdir(d):
        return None
    def keep(n):
        if not n.endswith("pth"):                 # the reference's own test: model_name[-3:] == 'pth' -- any name
            return False
        if end_iter is None:
            return True
        digits = n[5:-4]
        return n.startswith("ckpt_") and digits.isdigit() and int(digits) <= end_iter
    names = sorted(n for n in os.listdir(d) if keep(n))
    return names[-1] if names else None


def _load_ckpt(path, map_location, allow_pickle):
    """torch.load restricted to tensors / containers plus the numpy scalar types the reference runner leaves in the
    optimizer state (its learning rates are numpy floats); arbitrary pickles (code execution on load) only on explicit
    opt-in: `allow_pickle=True` or NUDF_CKPT_ALLOW_PICKLE=1."""
    import numpy as np
    safe = [np.dtype, np.float64, np.float32, np.int64, np.int32, np.bool_, np.ndarray]
    core = getattr(np, "_core", None) or getattr(np, "core")         # numpy >= 2 renamed numpy.core
    for name in ("scalar", "_reconstruct"):
        if hasattr(core.multiarray, name):
            safe.append(getattr(core.multiarray, name))
    try:
        from numpy import dtypes as _dt
        safe += [getattr(_dt, n) for n in ("Float64DType", "Float32DType", "Int64DType", "Int32DType", "BoolDType")
                 if hasattr(_dt, n)]
    except ImportError:
        pass
    import pickle
    if not os.path.exists(path):                    # a missing file is a missing file, not an unsafe pickle
        raise FileNotFoundError(path)
    if hasattr(torch.serialization, "safe_globals"):
        ctx = torch.serialization.safe_globals(safe)
    else:                                           # older torch: process-wide allow-list instead of a scoped one
        import contextlib
        torch.serialization.add_safe_globals(safe)
        ctx = contextlib.nullcontext()
    try:
        with ctx:
            return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:     # the weights-only unpickler met something beyond tensors + numpy scalars;
        # OSError, EOFError, zip / storage errors of a corrupt file propagate unchanged
        if allow_pickle or os.environ.get("NUDF_CKPT_ALLOW_PICKLE", "0") == "1":
            return torch.load(path, map_location=map_location, weights_only=False)
        raise RuntimeError("checkpoint %s holds objects beyond tensors and numpy scalars (%s); pass allow_pickle=True (or "
                           "NUDF_CKPT_ALLOW_PICKLE=1) only for files you trust" % (path, e)) from e


def load_checkpoint(path, nerf, udf_network, variance_network, color_network, beta_network, optimizer=None,
                    map_location=None, is_finetune=False, allow_pickle=False):
    """-> iter_step (0 when fine-tuning, :483-484).  Loading bumps the parameters' version counters (load_state_dict
    copies in place), which is what invalidates the packed-weight caches of the MLP engines."""
    ckpt = _load_ckpt(path, map_location, allow_pickle)
    nerf.load_state_dict(ckpt["nerf"])
    udf_network.load_state_dict(ckpt["udf_network_fine"])
    variance_network.load_state_dict(ckpt["variance_network_fine"])
    color_network.load_state_dict(ckpt["color_network_fine"])
    beta_network.load_state_dict(ckpt["beta_network"])
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    return 0 if is_finetune else ckpt["iter_step"]
