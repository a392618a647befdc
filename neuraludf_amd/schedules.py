"""Iteration schedules of the reference runner, as pure functions of ``iter_step`` (host scalars; nothing here touches
the GPU).  Each function cites the ``exp_runner_blending.py`` method it mirrors; a ``Schedules`` object bundles the
conf-derived constants and applies the learning rates to an optimizer the way the training loop does (:264-268)."""
from __future__ import annotations

import math
from dataclasses import dataclass


def learning_factor(iter_step, warm_up_end, end_iter, alpha):
    """update_learning_rate, exp_runner_blending.py:167-176: linear warm-up, then cosine decay to ``alpha``."""
    if iter_step < warm_up_end:
        return iter_step / warm_up_end
    progress = (iter_step - warm_up_end) / (end_iter - warm_up_end)
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha


def learning_factor_geo(iter_step, fix_geo_end, warm_up_end, end_iter, alpha):
    """update_learning_rate_geo, :178-191: geometry frozen first, slower warm-up, flat until half time, cosine."""
    if iter_step < fix_geo_end:
        return 0.0
    if iter_step < warm_up_end * 2:
        return iter_step / (warm_up_end * 2)
    if iter_step < end_iter * 0.5:
        return 1.0
    progress = (iter_step - end_iter * 0.5) / (end_iter - end_iter * 0.5)
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha


def cos_anneal_ratio(iter_step, anneal_end):
    """get_cos_anneal_ratio, :193-197."""
    return 1.0 if anneal_end == 0.0 else min(1.0, iter_step / anneal_end)


def regularization_weights(iter_step, end_iter, igr_ns_weight, sparse_weight):
    """regularization_weights_schedule, :199-211 -> (igr_ns_weight, sparse_weight) at this iteration."""
    end1, end2 = end_iter // 5, end_iter // 2
    ns = igr_ns_weight * min(max((iter_step - end1) / end1, 0.0), 1.0) if iter_step >= end1 else 0.0
    return ns, (sparse_weight if iter_step >= end2 else 0.0)


def flip_saturation(iter_step, end_iter, is_finetune=False, flip_saturation_max=0.9):
    """get_flip_saturation, :216-228."""
    if is_finetune:
        return 1.0
    if iter_step < 10000:
        return 0.0
    return flip_saturation_max if iter_step < end_iter * 0.5 else 1.0


def color_loss_weights(iter_step, color_base_weight, color_weight, color_pixel_weight, color_patch_weight,
                       is_finetune=False):
    """adjust_color_loss_weights, :230-251 -> (base, colour, pixel, patch) weights."""
    if is_finetune:
        factor = 1.0
    elif iter_step < 10000:
        factor = 0
    elif iter_step < 20000:
        factor = min(max((iter_step - 10000) / 10000, 0), 1)
    else:
        factor = 1.0
    base = color_base_weight * factor if color_base_weight < color_weight else color_base_weight
    return base, color_weight, color_pixel_weight * factor, color_patch_weight * factor


@dataclass
class Schedules:
    """constants read at exp_runner_blending.py:64-97 (same names, same defaults)."""
    end_iter: int
    learning_rate: float
    learning_rate_geo: float
    learning_rate_alpha: float
    warm_up_end: float = 0.0
    anneal_end: float = 0.0
    fix_geo_end: float = 500
    same_lr: bool = False
    igr_ns_weight: float = 0.0
    sparse_weight: float = 0.0
    color_base_weight: float = 0.0
    color_weight: float = 0.0
    color_pixel_weight: float = 0.0
    color_patch_weight: float = 0.0
    is_finetune: bool = False
    # the runner's --reg_weights_schedule flag (default False, exp_runner_blending.py:885, 361-365): without it the loop
    # uses the CONSTANT igr_ns_weight / sparse_weight of the conf; only with it regularization_weights_schedule applies
    reg_weights_schedule: bool = False

    @classmethod
    def from_conf(cls, conf, is_finetune=False, reg_weights_schedule=False):
        t, c = conf["train"], conf["color_loss"]
        return cls(end_iter=t.get_int("end_iter"), learning_rate=t.get_float("learning_rate"),
                   learning_rate_geo=t.get_float("learning_rate_geo"),
                   learning_rate_alpha=t.get_float("learning_rate_alpha"),
                   warm_up_end=t.get_float("warm_up_end", default=0.0), anneal_end=t.get_float("anneal_end", default=0.0),
                   fix_geo_end=t.get_float("fix_geo_end", default=500), same_lr=t.get_bool("same_lr", default=False),
                   igr_ns_weight=t.get_float("igr_ns_weight", default=0.0),
                   sparse_weight=t.get_float("sparse_weight", default=0.0),
                   color_base_weight=c.get_float("color_base_weight", 0.0), color_weight=c.get_float("color_weight", 0.0),
                   color_pixel_weight=c.get_float("color_pixel_weight", 0.0),
                   color_patch_weight=c.get_float("color_patch_weight", 0.0), is_finetune=is_finetune,
                   reg_weights_schedule=reg_weights_schedule)

    def apply_learning_rates(self, optimizer, iter_step):
        """the first lines of every training iteration (:264-268): group 0 = geometry, groups 1.. = the rest."""
        f = learning_factor(iter_step, self.warm_up_end, self.end_iter, self.learning_rate_alpha)
        start = 0 if self.same_lr else 1
        for g in optimizer.param_groups[start:]:
            g["lr"] = self.learning_rate * f
        if not self.same_lr:
            fg = learning_factor_geo(iter_step, self.fix_geo_end, self.warm_up_end, self.end_iter,
                                     self.learning_rate_alpha)
            for g in optimizer.param_groups[:1]:
                g["lr"] = self.learning_rate_geo * fg

    def at(self, iter_step):
        """everything the loop needs at this iteration, as a dict of python floats."""
        if self.reg_weights_schedule:
            ns, sp = regularization_weights(iter_step, self.end_iter, self.igr_ns_weight, self.sparse_weight)
        else:                                      # exp_runner_blending.py:361-363
            ns, sp = self.igr_ns_weight, self.sparse_weight
        b, c, px, pt = color_loss_weights(iter_step, self.color_base_weight, self.color_weight,
                                          self.color_pixel_weight, self.color_patch_weight, self.is_finetune)
        return dict(cos_anneal_ratio=cos_anneal_ratio(iter_step, self.anneal_end),
                    flip_saturation=flip_saturation(iter_step, self.end_iter, self.is_finetune),
                    igr_ns_weight=ns, sparse_weight=sp, color_base_weight=b, color_weight=c, color_pixel_weight=px,
                    color_patch_weight=pt)
