"""Fused Adam for the hot path's five networks: one `nudf_adam_step` launch per <= 64 tensors instead
of the ~12 foreach kernels (x 3 parameter groups) of torch.optim.Adam.

Same constructor surface as `torch.optim.Adam(param_groups, lr=...)` as the reference runner uses it
(exp_runner_blending.py:136-139: three groups, per-group `lr`; :167-191 rewrites `g['lr']` every
iteration) and the same per-parameter state (`step`, `exp_avg`, `exp_avg_sq`), so `state_dict()` /
`load_state_dict()` interchange with torch.optim.Adam checkpoints (:484-498).  Not supported (never
used by the reference): amsgrad, weight_decay, maximize."""
from __future__ import annotations

import math

import torch

from ._lib import ADAM_MAX_GROUPS, ADAM_MAX_TENSORS, Adam, call, lib, NudfError


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam: weight_decay / amsgrad are not used by the reference runner")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        super().__init__(params, defaults)
        if len(self.param_groups) > ADAM_MAX_GROUPS:
            raise NudfError("FusedAdam supports at most %d parameter groups" % ADAM_MAX_GROUPS)
        self._chunk = None
        # graph capture (train.GraphedStep): device address of a float buffer that holds {neg_step_size, bc2_sqrt} per
        # tensor of the launch order -- the two numbers of a step that depend on the iteration (learning-rate schedule,
        # bias corrections).  None: they travel by value (eager steps).
        self.dyn_base = None
        self._plan = None         # the last step's filled launch descriptors (reused while every address stays put)
        self.plan_hits = 0        # steps that reused them (tests)
        self._order = []          # parameters of the last step() in launch order
        self.state_epoch = 0      # bumped whenever the state tensors are replaced (load_state_dict): a captured step
                                  # holds raw pointers to exp_avg / exp_avg_sq (train.GraphedStep drops its captures)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.state_epoch += 1

    def dyn_values(self):
        """{neg_step_size, bc2_sqrt} of the NEXT step for the parameters of the last step(), in launch order (the layout of
        the device buffer behind `dyn_base`), from the groups' current learning rates and the parameters' step counts."""
        vals = []
        for p, gi in self._order:
            group = self.param_groups[gi]
            step = float(self.state[p]["step"]) + 1.0
            b1, b2 = group["betas"]
            vals.append(-group["lr"] / (1.0 - b1 ** step))
            vals.append(math.sqrt(1.0 - b2 ** step))
        return vals

    def advance(self):
        """host bookkeeping of one replayed step (the kernels ran from a captured graph): step counts and version
        counters of the parameters that took part, exactly what step() does after its launch."""
        for p, _ in self._order:
            self.state[p]["step"] += 1
            torch.autograd.graph.increment_version(p)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._chunk is None:
            self._chunk = int(lib().nudf_adam_chunk())
        chunk = self._chunk
        # ---- repeat step: the filled launch descriptors of the previous step are reused when the same parameters (same
        # addresses, same state tensors) have gradients again -- the steady state of a training loop; only the gradient address and
        # the two per-iteration numbers of each tensor are rewritten (eager host cost)
        plan = self._plan
        if plan is not None and plan["epoch"] == self.state_epoch and plan["dyn"] == self.dyn_base:
            ok = plan["n_with_grad"] == sum(1 for group in self.param_groups for q in group["params"] if q.grad is not None)
            if ok:
                memo = {}
                state = self.state
                for (pp, gi, t), pa, st_step in zip(plan["entries"], plan["addr"], plan["steps"]):
                    g = pp.grad
                    st = state[pp]
                    # the descriptor holds raw addresses of the parameter AND of both moment tensors: a state tensor replaced
                    # behind the optimizer's back (opt.state[p]["exp_avg"] = ..., module.to()) must not be written through
                    if (g is None or pp.data_ptr() != pa or not g.is_contiguous() or st["step"] is not st_step
                            or st["exp_avg"].data_ptr() != t.m or st["exp_avg_sq"].data_ptr() != t.v
                            or g.dtype != torch.float32 or not g.is_cuda):
                        ok = False
                        break
                    k = (gi, float(st_step))
                    v = memo.get(k)
                    if v is None:
                        group = self.param_groups[gi]
                        step = k[1] + 1.0
                        b1, b2 = group["betas"]
                        v = memo[k] = (-group["lr"] / (1.0 - b1 ** step), math.sqrt(1.0 - b2 ** step))
                    t.g = g.data_ptr()
                    t.neg_step_size, t.bc2_sqrt = v
            if ok:
                self.plan_hits += 1
                for a in plan["launches"]:
                    self._fill_groups(a)
                    call("nudf_adam_step", a)
                self._order = plan["order"]
                if torch.cuda.is_current_stream_capturing():
                    return loss
                torch._foreach_add_(plan["steps"], 1.0)
                torch.autograd.graph.increment_version(plan["params"])
                return loss
        a = Adam()
        launches, entries = [], []
        keep = []           # keep contiguous grad copies alive until the launch is enqueued
        nt = 0
        blocks = 0

        order = []
        launched = 0            # tensors of earlier launches of this step (offset into the dynamic-scalar buffer)

        def flush():
            nonlocal a, nt, blocks, launched
            if nt:
                a.n_tensors = nt
                a.block_start[nt] = blocks
                if self.dyn_base is not None:
                    a.dyn = self.dyn_base + 8 * launched
                call("nudf_adam_step", a)
                launches.append(a)
            launched += nt
            a = Adam()
            self._fill_groups(a)
            nt = 0
            blocks = 0

        self._fill_groups(a)
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise NudfError("FusedAdam needs device parameters (no CPU fallback)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g = p.grad
                if g.dtype != torch.float32 or not g.is_cuda or p.dtype != torch.float32:
                    raise NudfError("FusedAdam needs fp32 device parameters and gradients")
                if not g.is_contiguous():
                    g = g.contiguous()
                    keep.append(g)
                if not p.is_contiguous():
                    raise NudfError("FusedAdam needs contiguous parameters")
                if nt == ADAM_MAX_TENSORS:
                    flush()
                t = a.t[nt]
                t.p, t.g, t.m, t.v = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                t.n, t.group = p.numel(), gi
                step = float(st["step"]) + 1.0
                b1, b2 = group["betas"]
                t.neg_step_size = -group["lr"] / (1.0 - b1 ** step)
                t.bc2_sqrt = math.sqrt(1.0 - b2 ** step)
                a.block_start[nt] = blocks
                blocks += (p.numel() + chunk - 1) // chunk
                nt += 1
                order.append((p, gi))
                entries.append((p, gi, t))
        flush()
        self._order = order
        self._plan = None if keep else {
            "epoch": self.state_epoch, "dyn": self.dyn_base, "entries": entries, "launches": launches, "order": order,
            "addr": [q.data_ptr() for q, _, _ in entries], "params": [q for q, _ in order],
            "steps": [self.state[q]["step"] for q, _ in order],
            "n_with_grad": len(entries)}
        if torch.cuda.is_current_stream_capturing():
            return loss         # nothing ran: train.GraphedStep replays the graph and calls advance() per step
        for p, _ in order:
            self.state[p]["step"] += 1
            # the kernel wrote through a raw pointer: tell autograd (and every cache keyed on the version
            # counter, e.g. the packed weights of mlp.PackedLinear) that the tensor changed
            torch.autograd.graph.increment_version(p)
        return loss

    def _fill_groups(self, a):
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            g = a.group[gi]
            g.one_minus_beta1 = 1.0 - b1
            g.beta2 = b2
            g.one_minus_beta2 = 1.0 - b2
            g.eps = group["eps"]
