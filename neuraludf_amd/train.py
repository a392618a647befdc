"""One training step of the hot path, assembled exactly like the reference runner does it
(exp_runner_blending.py:125-139 networks + Adam groups, :296-375 loss assembly), for bench.py,
`__graft_entry__.smoke()` and the tests.  The runner itself is out of scope (it is the caller that
drops in unchanged); this is the minimum of it needed to time / check a full step."""
from __future__ import annotations

import contextlib
import io

import os

import torch

from . import dist as nudf_dist
from ._lib import LW, LW_COUNT
from .loss.loss import ColorLoss
from .models import fields
from .models.udf_renderer_blending import UDFRendererBlending

def other_scalars(var, beta):
    return list(var.parameters()) + list(beta.parameters())


# shipped DTU conf (confs/udf_dtu_blending.conf:56-118)
DTU_MODEL_CONF = dict(
    nerf=dict(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
              use_viewdirs=True),
    udf_network=dict(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                     geometric_init=True, weight_norm=True, udf_type="abs"),
    variance_network=dict(init_val=0.3),
    rendering_network=dict(d_feature=256, mode="no_normal", d_in=6, d_out=3, d_hidden=128, n_layers=4,
                           weight_norm=True, multires_view=4, squeeze_out=True, blending_cand_views=10),
    beta_network=dict(init_var_beta=0.5, init_var_gamma=0.3, init_var_zeta=0.3, beta_min=0.00005,
                      requires_grad_beta=True, requires_grad_gamma=False, requires_grad_zeta=False),
)


class StepScalars:
    """The per-iteration numbers a graph-captured step reads from device memory -- {cos_anneal_ratio, flip_saturation}
    for the composite kernels (NudfComposite.sched), the loss / regulariser weights (NUDF_LW_*: every loss kernel's
    `w_dev` and the 0-d views the generic loss expressions multiply by) and {neg_step_size, bc2_sqrt} per tensor for the
    fused Adam (NudfAdam.dyn) -- in ONE device buffer, refreshed by ONE asynchronous H2D copy per step from a ring of
    pinned host slots (a slot is rewritten only after the copy that read it has completed, so the host may run several
    steps ahead of the GPU).  `n_adam` = optimizer tensors of the step the buffer is made for."""
    SCHED, LOSSW, ADAM = 0, 8, 8 + LW_COUNT          # float offsets of the three regions

    def __init__(self, device, n_adam=96, slots=8):
        n = self.ADAM + 2 * n_adam
        self.n_adam = n_adam
        self.dev = torch.zeros(n, device=device)
        self.host = [torch.zeros(n).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.k = 0

    def upload(self, sched, lossw, adam):
        i = self.k % len(self.host)
        self.k += 1
        if self.events[i] is not None:
            self.events[i].synchronize()
        h = self.host[i].numpy()
        h[self.SCHED:self.SCHED + len(sched)] = sched
        h[self.LOSSW:self.LOSSW + len(lossw)] = lossw
        if len(adam) > 2 * self.n_adam:
            raise ValueError("too many optimizer tensors for the step-scalar buffer")
        h[self.ADAM:self.ADAM + len(adam)] = adam
        self.dev.copy_(self.host[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[i] = ev


SINGLE_THREAD_BACKWARD = os.environ.get("NUDF_HOST_FAST", "1") != "0"


class GraphedStep:
    """`Trainer.step` captured once in a HIP graph and replayed: ~65-75 kernel launches (~50 us of Python + ctypes each,
    3.5 ms per step) become one graph launch, so the host stays far ahead of the GPU and a host hiccup on one rank no
    longer stalls a collective.  Every libnudf entry point takes the stream and never synchronises, torch's fills / cats /
    random draws are graph-safe; what changes from iteration to iteration travels through device memory (StepScalars):
    the learning rates and Adam's bias corrections, cos_anneal_ratio and flip_saturation, and every loss / regulariser
    weight (the colour weights ramp over iterations 10 000-20 000 and the regulariser weights under
    --reg_weights_schedule, exp_runner_blending.py:199-211, 230-251: a weight baked into the capture would be replayed
    stale, a weight in the key would re-capture on every iteration of a ramp).  What is left of the capture key is what
    changes the SEQUENCE of launches: tensor shapes, which loss terms are switched on at all (weight > 0), loss types,
    which parameters train.  A new key runs `eager_steps` ordinary steps first (allocator and caches settle) and is
    captured on the next call.  Replays are bit-identical to eager steps (tests/test_gpu_graph.py).  Single process only
    by default: a data-parallel trainer stays eager unless `capture_collectives` (its two all-reduces capture on this
    image, scripts/rccl_capture_probe.py, but have not been replayed on two GPUs yet).

    The returned (loss, out) of a replayed call are the capture's STATIC tensors: the next replay overwrites them --
    clone what is kept across iterations (logging lists).  A captured step holds raw pointers to the parameters, their
    gradients and Adam's moments: `optimizer.load_state_dict` (FusedAdam bumps `state_epoch`) and `invalidate()` drop
    the captures; parameters reloaded in place (`module.load_state_dict`) keep their storage and need nothing."""

    def __init__(self, trainer, eager_steps=2, max_graphs=4, capture_collectives=False):
        from .optim import FusedAdam
        self.tr = trainer
        self.eager_steps = eager_steps
        self.max_graphs = max_graphs
        self.graphs = {}
        self.enabled = isinstance(trainer.optimizer, FusedAdam) and (capture_collectives or not trainer.data_parallel)
        self.scalars = None
        self.replays = 0
        self.captures = 0
        self._last = None
        self._epoch = getattr(trainer.optimizer, "state_epoch", 0)

    def invalidate(self):
        """drop every captured step (their pointers into optimizer state / parameters / gradients may be stale); the next
        calls run eagerly and re-capture."""
        self.graphs.clear()
        self._last = None

    def _key(self, batch, blend, has_anneal, perturb_overwrite):
        tr = self.tr
        sig = lambda d: tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in d.items() if torch.is_tensor(v)))
        live = tuple(p.requires_grad for g in tr.param_groups for p in g)
        # weights: only whether a term is on at all (that decides which kernels run); their VALUES travel through memory
        cw = tr.color_loss
        terms = (cw.color_pixel_weight > 0, cw.color_patch_weight > 0, tr.lc["color_pixel_weight"] > 0,
                 tr.lc["color_patch_weight"] > 0, tr.tc["mask_weight"] > 0)
        static = tuple(sorted((k, v) for k, v in tr.lc.items() if not k.endswith("_weight")))
        return (sig(batch), sig(blend) if blend is not None else None, bool(has_anneal), perturb_overwrite, terms, static,
                live)

    def __call__(self, batch, cos_anneal_ratio=1.0, flip_saturation=1.0, blend=None, perturb_overwrite=-1):
        tr = self.tr
        kw = dict(cos_anneal_ratio=cos_anneal_ratio, flip_saturation=flip_saturation, blend=blend,
                  perturb_overwrite=perturb_overwrite)
        if not self.enabled:
            return tr.step(batch, **kw)
        if getattr(tr.optimizer, "state_epoch", 0) != self._epoch:       # optimizer state reloaded: moments moved
            self._epoch = tr.optimizer.state_epoch
            self.invalidate()
        if (blend is not None and "ref_cam" not in blend and tr.lc["color_pixel_weight"] > 0
                and tr.lc["color_patch_weight"] > 0):
            # the camera constants of the patch warp hold four small matrix inverses (a solver library: not capturable):
            # computed here, per step, outside the graph; the captured step reads them as inputs
            from .models import blend as _blend
            blend = dict(blend)
            blend["ref_cam"], blend["src_cam"] = _blend.patch_cameras(blend["intrinsics"][0], blend["intrinsics"],
                                                                      blend["query_c2w"], torch.inverse(blend["w2cs"]))
            kw["blend"] = blend
        key = self._key(batch, blend, cos_anneal_ratio is not None, perturb_overwrite)
        ent = self.graphs.get(key)
        if ent is None:
            if len(self.graphs) >= self.max_graphs:          # schedules moved on: drop the oldest capture
                self.graphs.pop(next(iter(self.graphs)))
            ent = self.graphs[key] = dict(calls=0, graph=None)
        ent["calls"] += 1
        self._last = ent
        if ent["graph"] is None:
            if ent["calls"] <= self.eager_steps or ent.get("failed"):
                return tr.step(batch, **kw)
            prev_stream = torch.cuda.current_stream()
            try:
                self._capture(ent, batch, blend, cos_anneal_ratio is not None, flip_saturation, perturb_overwrite)
            except Exception as e:      # an op of this configuration cannot be captured: this key stays eager
                import warnings
                ent["failed"], ent["graph"] = True, None
                torch.cuda.set_stream(prev_stream)      # torch.cuda.graph.__exit__ skips this when capture_end raises
                torch.cuda.synchronize()
                warnings.warn("GraphedStep: capture failed (%s); this configuration keeps running eagerly" % (e,))
                return tr.step(batch, **kw)
        # inputs: copied into the captured step's static tensors -- unless the caller already writes them there
        # (`static_inputs`: a resident batch, or a batch generator given these tensors as its outputs): every copy is a
        # ~5 us launch on the critical path
        for k, v in ent["batch"].items():
            src = batch[k]
            if torch.is_tensor(v) and src is not v and src.data_ptr() != v.data_ptr():
                v.copy_(src)
        if blend is not None:
            for k, v in ent["blend"].items():
                src = blend[k]
                if torch.is_tensor(v) and src is not v and src.data_ptr() != v.data_ptr():
                    v.copy_(src)
        tr.optimizer._order = ent["order"]       # the tensors of THIS capture's Adam launch (an eager step of another
                                                 # configuration in between may have left another list)
        ent["scalars"].upload([0.0 if cos_anneal_ratio is None else float(cos_anneal_ratio), float(flip_saturation)],
                              tr.loss_weights().values(), tr.optimizer.dyn_values())
        ent["graph"].replay()
        tr.optimizer.advance()
        self.replays += 1
        return ent["loss"], ent["out"]

    def static_inputs(self):
        """(batch, blend) input tensors of the step captured / replayed by the last call (None before a capture): fill
        them in place -- or pass them back as the batch -- and the replay needs no input copies."""
        ent = self._last
        if ent is None or ent.get("graph") is None:
            return None
        return ent["batch"], ent["blend"]

    def _capture(self, ent, batch, blend, has_anneal, flip_saturation, perturb_overwrite):
        tr = self.tr
        dev = batch["rays_o"].device
        n_adam = sum(len(g) for g in tr.param_groups)
        if self.scalars is None or self.scalars.n_adam < n_adam:
            # sized for every tensor the optimizer owns (an over-long launch order can then not overflow it); captures made
            # with an older buffer keep a reference to it (ent["scalars"])
            self.scalars = StepScalars(dev, n_adam=max(96, n_adam))
        ent["scalars"] = self.scalars
        ent["batch"] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        ent["blend"] = ({k: (v.clone() if torch.is_tensor(v) else v) for k, v in blend.items()}
                        if blend is not None else None)
        for m in tr.modules().values():                      # the weight-pack launches must be inside the capture
            e = getattr(m, "_engine", None)
            if e is not None and hasattr(e, "mark_stale"):
                e.mark_stale()
        sc = self.scalars
        tr.renderer.sched_scalars = sc.dev[sc.SCHED:sc.SCHED + 2]
        tr.optimizer.dyn_base = sc.dev.data_ptr() + 4 * sc.ADAM
        tr.loss_weights().bind(sc.dev[sc.LOSSW:sc.LOSSW + LW_COUNT])
        # constants the step caches on first use are created HERE, outside the capture (a pageable host-to-device copy is not
        # capturable): normally the eager steps before the capture made them already; a (W, H) or patch size first seen by the
        # captured step itself would otherwise abort the capture
        if blend is not None:
            tr.color_loss.prefill_constants(dev)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        # Python's cyclic collector stays OFF while the stream is capturing: a collection pass in there may free objects of
        # EARLIER work (pinned host buffers, events, other graphs) whose destructors call into the runtime, which a
        # global-mode capture forbids -- the process aborts inside a destructor.  (torch.cuda.graph collects once on entry.)
        import gc
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(g):
                loss, out = tr.step(ent["batch"], cos_anneal_ratio=(1.0 if has_anneal else None),
                                    flip_saturation=flip_saturation, blend=ent["blend"],
                                    perturb_overwrite=perturb_overwrite)
        finally:
            if gc_was_on:
                gc.enable()
            tr.renderer.sched_scalars = None
            tr.optimizer.dyn_base = None
            tr.loss_weights().unbind()
        ent["graph"], ent["loss"], ent["out"], ent["order"] = g, loss, out, list(tr.optimizer._order)
        self.captures += 1


class Trainer:
    def __init__(self, device, renderer_conf, color_loss_conf=None, train_conf=None, seed=0, data_parallel=False,
                 fields_mod=fields, renderer_cls=UDFRendererBlending, loss_cls=ColorLoss, fused_adam=False,
                 model_conf=None):
        """`model_conf`: the `model { ... }` section of a conf (nerf, udf_network, variance_network, rendering_network,
        beta_network); default = the shipped DTU conf."""
        torch.manual_seed(seed)
        c = dict(DTU_MODEL_CONF)
        if model_conf is not None:
            c.update({k: dict(model_conf[k]) for k in DTU_MODEL_CONF if k in model_conf})
        with contextlib.redirect_stdout(io.StringIO()):
            self.nerf = fields_mod.NeRF(**c["nerf"]).to(device)
            self.udf = fields_mod.UDFNetwork(**c["udf_network"]).to(device)
            self.var = fields_mod.SingleVarianceNetwork(**c["variance_network"]).to(device)
            self.color = fields_mod.ResidualRenderingNetwork(**c["rendering_network"]).to(device)
            self.beta = fields_mod.BetaNetwork(**c["beta_network"]).to(device)
        tc = dict(learning_rate=5e-4, learning_rate_geo=1e-4, igr_weight=0.1, igr_ns_weight=0.0, mask_weight=0.0,
                  sparse_weight=0.0)
        tc.update(train_conf or {})
        self.tc = tc
        geo = list(self.udf.parameters())
        other = list(self.var.parameters()) + list(self.color.parameters()) + list(self.beta.parameters())
        nerf = list(self.nerf.parameters())
        self.param_groups = [geo, other, nerf]
        groups = [{'params': geo, 'lr': tc["learning_rate_geo"]}, {'params': other}, {'params': nerf}]
        if fused_adam:
            from .optim import FusedAdam
            self.optimizer = FusedAdam(groups, lr=tc["learning_rate"])
        else:
            self.optimizer = torch.optim.Adam(groups, lr=tc["learning_rate"])
        self.renderer = renderer_cls(self.nerf, self.udf, self.var, self.color, self.beta, **renderer_conf)
        lc = dict(color_base_weight=0.01, color_weight=1.0, color_pixel_weight=0.0, color_patch_weight=0.0,
                  pixel_loss_type="l1", patch_loss_type="ssim", h_patch_size=3)
        lc.update(color_loss_conf or {})
        self.lc = lc
        with contextlib.redirect_stdout(io.StringIO()):
            self.color_loss = loss_cls(**lc)
        self.data_parallel = data_parallel
        self.status_every = 100        # iterations between looks at the non-finite status word (the runner's report_freq)
        self._beta_flag = True
        # single process: the step's loss assembly is fused (see `loss`), so the renderer hands over the composite kernel's
        # sums instead of the three error terms -- requested per call inside `loss` only: a direct `renderer.render(...)`
        # (validation, debugging, the reference runner) keeps the reference's result dict
        self.fuse_loss = not data_parallel
        self._one = {}
        self.defer_sums_reduce = True      # (False: the composite launch is followed by its own reduction launch -- A/B, tests)
        import os
        # (False / NUDF_FUSE_BLEND_LOSS=0: the blending step's loss through the generic ColorLoss expressions -- A/B, tests)
        self.fuse_blend_loss = os.environ.get("NUDF_FUSE_BLEND_LOSS", "1") != "0"
        if data_parallel:
            self.renderer.data_parallel = True
            self.renderer.defer_loss_sums = True
            self.color_loss.set_data_parallel(True)
            # bucket layout in order of readiness in the backward; the NeRF (unused when n_outside = 0) last
            # a network with an MLP engine is laid out in the ENGINE's parameter order ([v, g, bias] per layer -- the order
            # its unpack kernel writes), not in nn.Module registration order ([bias, g, v] under weight_norm)
            def seg(m):
                e = m.engine() if hasattr(m, "engine") else None
                ps = list(e.params()) if e is not None else list(m.parameters())
                assert {id(p) for p in ps} == {id(p) for p in m.parameters()}
                return ps, e
            self.bucket = nudf_dist.GradBucket([(other_scalars(self.var, self.beta), None), seg(self.color), seg(self.udf),
                                                seg(self.nerf)], device=device)

    def modules(self):
        return dict(nerf=self.nerf, udf=self.udf, var=self.var, color=self.color, beta=self.beta)

    def enable_graph(self, eager_steps=2):
        """replay `step` from a captured HIP graph in `iteration` (see GraphedStep); -> the GraphedStep."""
        self.graphed = GraphedStep(self, eager_steps=eager_steps)
        return self.graphed

    def loss_weights(self):
        """the step's LossWeights (owned by the ColorLoss, so a runner that calls `color_loss.set_color_weights` reaches
        them too) with the regulariser / mask weights of `self.tc` written in."""
        lw = self.color_loss.weights
        tc = self.tc
        lw.set(igr=tc["igr_weight"], igr_ns=tc["igr_ns_weight"], sparse=tc["sparse_weight"], mask=tc["mask_weight"])
        return lw

    def loss(self, batch, cos_anneal_ratio=1.0, flip_saturation=1.0, blend=None, perturb_overwrite=-1):
        """-> (loss, render_out).  Every weight enters as a read of the device weight vector (LossWeights): by `w_dev` in
        the fused kernels, as 0-d tensor views in the torch expressions."""
        tc, lc = self.tc, self.lc
        w = self.loss_weights().device(batch["rays_o"].device)
        kw = {}
        if blend is not None and lc["color_pixel_weight"] > 0:
            kw = dict(color_maps=blend["color_maps"], w2cs=blend["w2cs"], intrinsics=blend["intrinsics"],
                      query_c2w=blend["query_c2w"],
                      rays_uv=batch["rays_uv"].clone() if lc["color_patch_weight"] > 0 else None)
            if "ref_cam" in blend:        # patch-camera constants computed by the caller (GraphedStep: outside the capture)
                kw["patch_cams"] = (blend["ref_cam"], blend["src_cam"])
        defer, defer_r = self.renderer.defer_loss_sums, self.renderer.defer_sums_reduce
        if self.fuse_loss:
            self.renderer.defer_loss_sums = True
            self.renderer.defer_sums_reduce = self.defer_sums_reduce   # the fused loss launch finishes the composite's sums
        try:
            out = self.renderer.render(batch["rays_o"], batch["rays_d"], batch["near"], batch["far"],
                                       flip_saturation=flip_saturation, cos_anneal_ratio=cos_anneal_ratio,
                                       perturb_overwrite=perturb_overwrite, **kw)
        finally:
            self.renderer.defer_loss_sums, self.renderer.defer_sums_reduce = defer, defer_r
        weight_sum = out["weight_sum"]
        pixel_mask = batch["mask"] if tc["mask_weight"] > 0 else None
        if (self.fuse_blend_loss and "_loss_sums" in out and not self.data_parallel and tc["mask_weight"] <= 0
                and out["patch_mask"] is not None
                and self.color_loss.blend_fusable(out["color_base"], out["color"], batch["true_rgb"], out["color_pixel"],
                                                  pixel_mask, out["patch_colors"], batch.get("gt_patch_colors"))):
            # the blending step (pixel + patch terms): the patch-mask algebra, ColorLoss, the trimmed patch loss, the three
            # regularisers and the total in three launches around one sort (loss._BlendStepLossFn)
            sums = out.pop("_loss_sums")
            r = self.color_loss.blend_step_loss(out["color_base"], out["color"], batch["true_rgb"], out["color_pixel"],
                                                out["patch_colors"], batch["gt_patch_colors"], out["patch_mask"], weight_sum,
                                                sums, float(weight_sum.shape[0]), w)
            out["gradient_error"], out["gradient_error_near_surface"], out["sparse_error"] = r[6], r[7], r[8]
            return r[0], out
        patch_mask = None
        if out["patch_mask"] is not None:
            patch_mask = (out["patch_mask"].float()[:, None] * (weight_sum > 0.5).float()) > 0.
        cargs = (out["color_base"], out["color"], batch["true_rgb"], out["color_pixel"], pixel_mask, out["patch_colors"],
                 batch.get("gt_patch_colors"), patch_mask)
        bce_sum = None
        if tc["mask_weight"] > 0:
            bce_sum = torch.nn.functional.binary_cross_entropy(weight_sum.clip(1e-3, 1.0 - 1e-3), batch["mask"],
                                                               reduction="sum")
        if "_loss_sums" in out and not self.data_parallel:
            # single-process step: the renderer handed over the composite kernel's five sums; when only the two L1 colour
            # terms are active (every shipped conf outside the *_ft blending ones) the whole loss assembly -- ColorLoss,
            # the three regularisers, the weighted total -- is ONE launch each way (loss._StepLossFn)
            sums = out.pop("_loss_sums")
            if bce_sum is None and self.color_loss.fusable(out["color_base"], out["color"], batch["true_rgb"],
                                                           out["color_pixel"], out["patch_colors"]):
                from .loss.loss import _StepLossFn
                loss, _, _, _, ge, gens, se = _StepLossFn.apply(
                    out["color_base"], out["color"], batch["true_rgb"], pixel_mask, sums, float(weight_sum.shape[0]), w)
                out["gradient_error"], out["gradient_error_near_surface"], out["sparse_error"] = ge, gens, se
                return loss, out
            ge, gens, se = self.renderer.errors_from_sums(sums, weight_sum.shape[0])
            out["gradient_error"], out["gradient_error_near_surface"], out["sparse_error"] = ge, gens, se
            cl = self.color_loss(*cargs)
            mask_loss = bce_sum / float(weight_sum.numel()) if bce_sum is not None else None
        elif "_loss_sums" in out:
            # ray-sharded step: ONE all-reduce of every batch-global partial sum -- the renderer's five, the fused colour
            # loss's three and the mask term's two (dist.py (1)); everything after it is the same arithmetic on every rank
            parts = [out.pop("_loss_sums")]
            fused = self.color_loss.fusable(out["color_base"], out["color"], batch["true_rgb"], out["color_pixel"],
                                            out["patch_colors"])
            if fused:
                parts.append(self.color_loss.local_sums(out["color_base"], out["color"], batch["true_rgb"], pixel_mask))
            if bce_sum is not None:
                parts.append(torch.stack([bce_sum, torch.full_like(bce_sum, float(weight_sum.numel()))]))
            packed = nudf_dist.all_reduce_sum(torch.cat([p.reshape(-1) for p in parts]))
            ge, gens, se = self.renderer.errors_from_sums(packed[:5], weight_sum.shape[0])
            out["gradient_error"], out["gradient_error_near_surface"], out["sparse_error"] = ge, gens, se
            k = 5
            if fused:
                cl = self.color_loss.from_global_sums(out["color_base"], out["color"], batch["true_rgb"], pixel_mask,
                                                      packed[k:k + 3])
                k += 3
            else:
                cl = self.color_loss(*cargs)            # pixel / patch terms: their own (counted) collectives
            mask_loss = packed[k] / packed[k + 1] if bce_sum is not None else None
        else:
            cl = self.color_loss(*cargs)
            mask_loss = bce_sum / float(weight_sum.numel()) if bce_sum is not None else None
        loss = cl["loss"] + out["gradient_error_near_surface"] * w[LW["igr_ns"]] \
            + out["sparse_error"] * w[LW["sparse"]] + out["gradient_error"] * w[LW["igr"]]
        if mask_loss is not None:
            loss = loss + mask_loss * w[LW["mask"]]
        return loss, out

    def step(self, batch, **kw):
        loss, out = self.loss(batch, **kw)
        self.optimizer.zero_grad(set_to_none=True)
        # the seed of the backward pass: a resident 1.0 instead of the ones_like(loss) fill autograd would launch
        one = self._one.get(loss.device)
        if one is None or one.dtype != loss.dtype:
            one = self._one[loss.device] = torch.ones((), device=loss.device, dtype=loss.dtype)
        # the backward functions of this step are ~35 kernel launches of Python each: on autograd's device thread they cost the
        # hand-over and the interpreter lock they fight the main thread for (measured, scripts/host_profile.py: 2.7 ms of
        # enqueueing per step with the engine's worker thread, 2.1 ms on the calling thread); one device, one stream --
        # nothing runs concurrently either way
        with torch.autograd.set_multithreading_enabled(not SINGLE_THREAD_BACKWARD):
            loss.backward(gradient=one)
        if self.data_parallel:
            self.bucket.all_reduce()
        self.optimizer.step()
        # detached: the autograd graph of the step is done with, and an `out` that kept it alive would also keep the
        # parameters' AccumulateGrad nodes (and the stream they were created on) alive into the next iteration -- which
        # breaks a later graph capture on another stream (GraphedStep)
        return loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}

    def iteration(self, source, iter_step, schedules, image_perm=None, batch_size=512, num_src=8):
        """one pass of the reference training loop body (exp_runner_blending.py:262-375) with every per-iteration
        input produced on the GPU: schedules -> ray / patch batch (one launch) -> render -> loss -> backward -> Adam.
        ``source`` is a ``dataset.RayBatchSource``; -> (loss, render_out, sample)."""
        schedules.apply_learning_rates(self.optimizer, iter_step)
        a = schedules.at(iter_step)
        self.color_loss.set_color_weights(a["color_base_weight"], a["color_weight"], a["color_pixel_weight"],
                                          a["color_patch_weight"])
        self.lc.update(color_pixel_weight=a["color_pixel_weight"], color_patch_weight=a["color_patch_weight"])
        self.tc.update(igr_ns_weight=a["igr_ns_weight"], sparse_weight=a["sparse_weight"])
        n = source.n_images
        img_idx = int(image_perm[iter_step % len(image_perm)]) if image_perm is not None else iter_step % n
        s = source.gen_random_rays_patches_at(img_idx, batch_size, crop_patch=a["color_patch_weight"] > 0.0,
                                              h_patch_size=self.color_loss.h_patch_size, with_near_far=True)
        data = s["rays"]
        batch = dict(rays_o=data[:, :3], rays_d=data[:, 3:6], true_rgb=data[:, 6:9], mask=(data[:, 9:10] > 0.5).float(),
                     near=s["near"], far=s["far"], rays_uv=s["rays_ndc_uv"], gt_patch_colors=s["rays_patch_color"])
        blend = None
        if a["color_pixel_weight"] > 0.0 or a["color_patch_weight"] > 0.0:
            ref_c2w, src_c2ws, src_intr, src_images, _ = source.get_ref_src_info(img_idx, num_src)
            blend = dict(color_maps=src_images, w2cs=source.src_w2cs(img_idx, num_src), intrinsics=src_intr, query_c2w=ref_c2w)
        stepper = self.graphed if getattr(self, "graphed", None) is not None else self.step
        loss, out = stepper(batch, cos_anneal_ratio=a["cos_anneal_ratio"], flip_saturation=a["flip_saturation"],
                            blend=blend)
        self._trainability_toggles(out, iter_step)
        # the reference stops on the host the moment a NaN appears (udf_renderer_blending.py:97, 265, 543, 860); here the
        # kernels leave a bit in a device word and the loop looks at it every `status_every` iterations (one 4-byte read,
        # the only sync of the loop) -- a run that went non-finite stops within that many iterations instead of silently
        # writing black images
        if self.status_every and iter_step % self.status_every == 0:
            self.check_finite(loss)
        return loss, out, s

    def check_finite(self, loss=None):
        """renderer.check_finite(), agreed on by all ranks of a ray-sharded job: the status word is rank-local (a NaN on one
        rank's rays), and a rank that raised alone would leave the others waiting in the next all-reduce until the RCCL
        timeout.  The words are OR-ed (all-reduce MAX per bit) on the check iterations, then every rank raises together.
        Which bits exist: the composite and up-sampling launches report on every path; the LOSS bit is set by the fused
        single-process loss kernels (nudf_step_loss_fwd, nudf_blend_loss_fwd) only -- the ray-sharded loss is formed from
        all-reduced sums by torch ops and is looked at here instead."""
        if not self.data_parallel:
            return self.renderer.check_finite()
        import torch.distributed as dist
        bits = self.renderer.status(clear=True)
        if loss is not None and not bool(torch.isfinite(loss.detach()).all()):
            from neuraludf_amd import _lib as _l
            bits |= _l.STATUS_NONFINITE_LOSS
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dev = self.renderer._device()
            t = torch.tensor([float((bits >> b) & 1) for b in range(3)], device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bits = sum(int(v) << b for b, v in enumerate(t.tolist()))
        if bits:
            from neuraludf_amd import _lib
            raise FloatingPointError("NeuralUDF renderer (some rank of the ray-sharded job): " + _lib.status_text(bits))

    def _trainability_toggles(self, out, iter_step):
        """exp_runner_blending.py:352-358, evaluated on this iteration's render output like there: once the variance has
        dropped below 2 beta and 0.01, beta becomes trainable (takes effect from the next iteration's graph on, as in the
        reference, where the flag flips after the forward pass); a frozen variance becomes trainable once
        iter_step > 20 000.  The first test reads two device scalars (a host sync the reference pays every iteration):
        it is evaluated every iteration, but only while beta is still frozen -- confs whose beta is trainable from the
        start (all shipped ones: set_beta_trainable would be a no-op) never sync."""
        var_p = getattr(self.var, "variance", None)
        beta_p = getattr(self.beta, "beta", None)
        if (self._beta_flag and beta_p is not None and not beta_p.requires_grad and var_p is not None
                and var_p.requires_grad):
            variance, beta = float(out["variance"].mean()), float(out["beta"].reshape(-1)[0])
            if variance < 2 * beta and variance < 0.01:
                self.beta.set_beta_trainable()
                self._beta_flag = False
        if var_p is not None and var_p.requires_grad is False and iter_step > 20000:
            self.var.set_trainable()

    @torch.no_grad()
    def render_image(self, source, img_idx, resolution_level=4, chunk=65536, cos_anneal_ratio=1.0):
        """validation render of one view (the loop of exp_runner_blending.py:506-560 without the file output): whole-image
        rays from the source, rendered in chunks of `chunk` rays -- large chunks, the renderer's working set at 65 536
        rays x 146 samples is a few GB of the 288 GB -- -> dict(color [H', W', 3], depth [H', W'], normals [H', W', 3])."""
        rays_o, rays_d = source.gen_rays_at(img_idx, resolution_level=resolution_level)
        Hh, Ww, _ = rays_o.shape
        rays_o, rays_d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
        outs = {"color": [], "depth": [], "normals": []}
        for i in range(0, rays_o.shape[0], chunk):
            o, d = rays_o[i:i + chunk], rays_d[i:i + chunk]
            near, far = source.near_far_from_sphere(o, d)
            r = self.renderer.render(o, d, near, far, cos_anneal_ratio=cos_anneal_ratio, perturb_overwrite=0,
                                     flip_saturation=1.0)
            outs["color"].append(r["color"]); outs["depth"].append(r["depth"].reshape(-1)); outs["normals"].append(r["normals"])
        return {"color": torch.cat(outs["color"]).reshape(Hh, Ww, 3), "depth": torch.cat(outs["depth"]).reshape(Hh, Ww),
                "normals": torch.cat(outs["normals"]).reshape(Hh, Ww, 3)}
