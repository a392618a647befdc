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


def load_checkpoint(path, nerf, udf_network, variance_network, color_network, beta_network, optimizer=None,
                    map_location=None, is_finetune=False):
    """-> iter_step (0 when fine-tuning, :483-484).  Loading bumps the parameters' version counters (load_state_dict
    copies in place), which is what invalidates the packed-weight caches of the MLP engines."""
    # weights_only=False: the reference runner stores numpy floats (its learning rates) in the optimizer state
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    nerf.load_state_dict(ckpt["nerf"])
    udf_network.load_state_dict(ckpt["udf_network_fine"])
    variance_network.load_state_dict(ckpt["variance_network_fine"])
    color_network.load_state_dict(ckpt["color_network_fine"])
    beta_network.load_state_dict(ckpt["beta_network"])
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    return 0 if is_finetune else ckpt["iter_step"]
