"""Drop-in for the reference's loss/loss.py: `ColorLoss(**conf.color_loss)` with the same call surface
and result dict (loss/loss.py:87-133).

The per-ray reductions are tiny ([N,3] tensors); the SSIM patch error runs in the `nudf_ssim_patch`
kernel.  With `data_parallel=True` the L1 numerators / mask counts are all-reduced over the process
group so every rank forms the same global loss (see neuraludf_amd/dist.py)."""
import torch
import torch.nn as nn

from .. import dist as nudf_dist
from .._lib import LW, LW_COUNT
from .patch_metric import PATCH_TYPES, patch_error


_PIN_RING = {"slots": [], "events": [], "k": 0}     # pinned staging slots of every LossWeights.device() (never freed)


class LossWeights:
    """The loss / regulariser weights of a train step (include/nudf.h NUDF_LW_*) as host floats plus a device mirror.

    The runner's schedules move them from iteration to iteration (adjust_color_loss_weights,
    regularization_weights_schedule: exp_runner_blending.py:199-211, 230-251), and a HIP-graph replay repeats the kernel
    ARGUMENTS of its capture -- so every consumer (the fused loss kernels through `w_dev`, the torch expressions of the
    generic path through 0-d views of the same vector) reads them from device memory, eager and replayed steps alike:
    the two stay bit-identical and a moving weight needs no new capture.  Eager steps refresh the mirror themselves when
    a value changed (`device()`); a captured step is bound to a slot of the trainer's per-step scalar upload
    (`bind()`, train.StepScalars) for the duration of the capture."""

    def __init__(self, **kw):
        self.host = [0.0] * LW_COUNT
        self._own = None            # this object's own device vector (eager steps)
        self._own_vals = None       # what it holds
        self._bound = None          # a caller-owned device vector whose contents the caller keeps current (graph capture)
        self.set(**kw)

    def set(self, **kw):
        for k, v in kw.items():
            self.host[LW[k]] = float(v)

    def get(self, name):
        return self.host[LW[name]]

    def values(self):
        """the device layout as python floats; the colour denominator is formed in double like the reference's
        `(color_base_weight + color_weight + color_pixel_weight)` (loss/loss.py:127-128) and rounded once."""
        v = list(self.host)
        v[LW["color_sum"]] = v[LW["color_base"]] + v[LW["color"]] + v[LW["color_pixel"]]
        return v

    def bind(self, dev_vector):
        self._bound = dev_vector

    def unbind(self):
        self._bound = None

    def device(self, device):
        """-> device float vector [LW_COUNT] holding `values()`."""
        if self._bound is not None:
            return self._bound
        vals = self.values()
        if self._own is not None and self._own.device == torch.device(device) and vals == self._own_vals:
            return self._own
        # a NEW small device tensor per change (a colour-weight ramp changes the values every iteration): autograd nodes
        # of a loss that has not run its backward yet may still hold views of the previous vector, so it is never
        # overwritten in place.  Staged through a RING of pinned host slots (as train.StepScalars): the copy does not stall
        # the host-ahead pipeline, and no pinned block is ever freed before the interpreter exits -- the host allocator
        # touches the block's events when a pinned block dies, which is not permitted while a stream is being captured in
        # global mode, and a garbage-collection pass inside GraphedStep._capture that freed the pinned slots of an EARLIER
        # trainer's weights hit exactly that (GraphedStep._capture also keeps the collector off for its duration).
        import torch as _t
        dev = _t.device(device)
        h = _t.tensor(vals, dtype=_t.float32, device="cpu")     # (the reference runner makes CUDA the default tensor type)
        if dev.type != "cuda":
            self._own, self._own_vals = h, vals
            return self._own
        ring = _PIN_RING          # ONE process-wide ring (module level: it outlives every LossWeights object, see above)
        if not ring["slots"]:
            ring["slots"] = [_t.empty(LW_COUNT, dtype=_t.float32, device="cpu").pin_memory() for _ in range(8)]
            ring["events"] = [None] * 8
        i = ring["k"] % len(ring["slots"])
        ring["k"] += 1
        if ring["events"][i] is not None:
            ring["events"][i].synchronize()      # (only ever waits when the host is eight weight changes ahead of the GPU)
        ring["slots"][i][:len(vals)].copy_(h)
        own = _t.empty(len(vals), dtype=_t.float32, device=dev)
        own.copy_(ring["slots"][i][:len(vals)], non_blocking=True)
        ev = _t.cuda.Event()
        ev.record()
        ring["events"][i] = ev
        self._own, self._own_vals = own, vals
        return self._own


class _L1SumFn(torch.autograd.Function):
    """sum |pred - gt| in one launch (and sign(pred - gt) * upstream in one launch back)."""

    @staticmethod
    def forward(ctx, pred, gt):
        from .._lib import call, ptr
        p, g = pred.detach().contiguous(), gt.detach().contiguous()
        out = torch.empty(1, device=p.device)
        call("nudf_l1_sum_fwd", ptr(p), ptr(g), p.numel(), ptr(out))
        ctx.save_for_backward(p, g)
        return out[0]

    @staticmethod
    def backward(ctx, d_out):
        from .._lib import call, ptr
        p, g = ctx.saved_tensors
        d_pred = torch.empty_like(p)
        call("nudf_l1_sum_bwd", ptr(p), ptr(g), p.numel(), ptr(d_out.reshape(1).contiguous()), ptr(d_pred))
        return d_pred, None


class _ColorLossFn(torch.autograd.Function):
    """(color_base, color, gt[, mask]) -> (total, color_base_loss, color_loss): the two L1 terms of ColorLoss and their
    weighting in one launch each way (the generic path costs ~12 + ~15 one-element launches)."""

    @staticmethod
    def forward(ctx, cb, c, gt, mask, w_dev, data_parallel=False):
        from .._lib import call, ptr
        ctx.set_materialize_grads(False)
        w_dev = w_dev.detach()
        cb_, c_, gt_ = cb.detach().contiguous(), c.detach().contiguous(), gt.detach().contiguous()
        m_ = mask.detach().float().contiguous() if mask is not None else None
        out = torch.empty(3, device=cb_.device)
        den = torch.empty(1, device=cb_.device)
        if data_parallel and nudf_dist.exchanging():
            # ray-sharded: local sums -> ONE all-reduce of 3 floats -> the same final arithmetic on every rank; the
            # backward below then differentiates the global loss w.r.t. the local rays (global denominator)
            import torch.distributed as dist
            sums = torch.empty(3, device=cb_.device)
            call("nudf_color_loss_sums", ptr(cb_), ptr(c_), ptr(gt_), cb_.numel(), ptr(m_),
                 m_.numel() if m_ is not None else 0, ptr(sums))
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            call("nudf_color_loss_finish", ptr(sums), 1 if m_ is not None else 0, 0.0, 0.0, 0.0, ptr(w_dev), ptr(out),
                 ptr(den))
        else:
            call("nudf_color_loss_fwd", ptr(cb_), ptr(c_), ptr(gt_), cb_.numel(), ptr(m_),
                 m_.numel() if m_ is not None else 0, 0.0, 0.0, 0.0, ptr(w_dev), ptr(out), ptr(den))
        ctx.save_for_backward(cb_, c_, gt_, den, w_dev)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, d0, d1, d2):
        from .._lib import call, ptr
        cb_, c_, gt_, den, w_dev = ctx.saved_tensors
        if d0 is None and d1 is None and d2 is None:
            return None, None, None, None, None, None
        z = den.new_zeros(())
        d_out = torch.stack([d if d is not None else z for d in (d0, d1, d2)])
        d_cb, d_c = torch.empty_like(cb_), torch.empty_like(c_)
        call("nudf_color_loss_bwd", ptr(cb_), ptr(c_), ptr(gt_), cb_.numel(), ptr(den), 0.0, 0.0, 0.0, ptr(w_dev),
             ptr(d_out), ptr(d_cb), ptr(d_c))
        return d_cb, d_c, None, None, None, None


class _StepLossFn(torch.autograd.Function):
    """(color_base, color, gt[, mask], composite sums[5]) -> (total, colour total, Lb, Lc, gradient_error,
    gradient_error_near_surface, sparse_error): ColorLoss's two L1 terms, the three regularisers and the runner's weighted
    total (exp_runner_blending.py:330-371) in ONE launch each way (nudf_step_loss_fwd / _bwd) instead of ~20 one-element
    launches.  Same values as the unfused chain (same reductions, every product / sum of the total rounded on its own)."""

    @staticmethod
    def forward(ctx, cb, c, gt, mask, sums, n_rays, w_dev):
        from .._lib import call, ptr
        ctx.set_materialize_grads(False)
        w_dev = w_dev.detach()
        cb_, c_, gt_ = cb.detach().contiguous(), c.detach().contiguous(), gt.detach().contiguous()
        m_ = mask.detach().float().contiguous() if mask is not None else None
        # the composite launch may have left the five sums as per-block partials (NudfComposite.defer_sums): this launch
        # finishes the reduction and fills `sums` (udf_renderer_blending.pending_sums)
        pend = getattr(sums, "_nudf_ws", None)
        if pend is not None:
            del sums._nudf_ws
        sums_ = sums.detach()
        assert sums_.is_contiguous()
        out = torch.empty(8, device=cb_.device)
        den = torch.empty(1, device=cb_.device)
        call("nudf_step_loss_fwd", ptr(cb_), ptr(c_), ptr(gt_), cb_.numel(), ptr(m_), m_.numel() if m_ is not None else 0,
             ptr(sums_), float(n_rays), 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, ptr(w_dev), ptr(out), ptr(den),
             ptr(pend[0]) if pend is not None else None, pend[1] if pend is not None else 0)
        ctx.save_for_backward(cb_, c_, gt_, den, sums_, w_dev)
        ctx.n_rays = float(n_rays)
        return tuple(out[i] for i in range(7))

    @staticmethod
    def backward(ctx, *d):
        from .._lib import call, ptr
        if all(x is None for x in d):
            return (None,) * 7
        cb_, c_, gt_, den, sums_, w_dev = ctx.saved_tensors
        d_total = d[0].contiguous() if d[0] is not None else None      # autograd's own tensor: no extra launch
        d_extra = None
        if any(x is not None for x in d[1:]) or d_total is None:      # gradients into the logged terms: rare
            z = den.new_zeros(())
            d_extra = torch.stack([z] + [x if x is not None else z for x in d[1:]] + [z])
            if d_total is None:
                d_total = z
        d_cb, d_c = torch.empty_like(cb_), torch.empty_like(c_)
        d_sums = torch.empty(5, device=cb_.device)
        call("nudf_step_loss_bwd", ptr(cb_), ptr(c_), ptr(gt_), cb_.numel(), ptr(den), ptr(sums_), ctx.n_rays,
             0.0, 0.0, 0.0, 0.0, 0.0, 0.0, ptr(w_dev), ptr(d_total), ptr(d_extra), ptr(d_cb), ptr(d_c), ptr(d_sums))
        return (d_cb, d_c, None, None, d_sums, None, None)


class _BlendStepLossFn(torch.autograd.Function):
    """The blending step's whole loss (BASELINE config 3) around ONE torch.sort: ColorLoss with its pixel and trimmed patch
    terms, the runner's patch-mask algebra, the three regularisers and the weighted total -- nudf_patch_metric +
    nudf_blend_loss_prepare + sort + nudf_blend_loss_fwd (and nudf_blend_loss_bwd + nudf_patch_metric back) instead of ~90
    one-element / per-ray torch launches.  -> (total, colour total, Lb, Lc, Lpix, Lpatch, gradient_error,
    gradient_error_near_surface, sparse_error, patch mask [N] bool).  Same expressions in the same order as the generic path
    (loss/loss.py:105-133, 21-44, 66-84; exp_runner_blending.py:313-315, 330-371); sums of N terms in this kernel's order."""

    @staticmethod
    def forward(ctx, cb, c, pix, gt, patch_colors, gt_patch, pm_raw, wsum, sums, n_rays, w_dev, window, kind, ratio):
        from .._lib import BlendLoss, call, ptr
        ctx.set_materialize_grads(False)
        t = lambda x: x.detach().float().contiguous()
        cb_, c_, pix_, gt_, pc_, gp_ = t(cb), t(c), t(pix), t(gt), t(patch_colors), t(gt_patch)
        pm_, ws_ = t(pm_raw).reshape(-1), t(wsum).reshape(-1)
        w_dev = w_dev.detach()
        dev = cb_.device
        N, npx = pc_.shape[0], pc_.shape[1]
        err = torch.empty(N, device=dev)
        call("nudf_patch_metric", kind, ptr(pc_), ptr(gp_), ptr(window), N, npx, ptr(err), None, None)
        a = BlendLoss()
        m, em = torch.empty(N, device=dev), torch.empty(N, device=dev)
        a.err, a.patch_mask, a.weight_sum, a.m, a.err_masked, a.N = ptr(err), ptr(pm_), ptr(ws_), ptr(m), ptr(em), N
        call("nudf_blend_loss_prepare", a)
        es, order = torch.sort(em, descending=True)          # (the same call as ColorPatchLoss.forward: same permutation)
        pend = getattr(sums, "_nudf_ws", None)
        if pend is not None:
            del sums._nudf_ws
        sums_ = sums.detach()
        out = torch.empty(12, device=dev)
        a.cb, a.c, a.pix, a.gt = ptr(cb_), ptr(c_), ptr(pix_), ptr(gt_)
        a.err_sorted, a.order, a.sums, a.w_dev, a.out = ptr(es), ptr(order), ptr(sums_), ptr(w_dev), ptr(out)
        if pend is not None:
            a.sums_ws, a.sums_nblk = ptr(pend[0]), pend[1]
        a.n_rays, a.trim_ratio = float(n_rays), float(ratio)
        call("nudf_blend_loss_fwd", a)
        ctx.save_for_backward(cb_, c_, pix_, gt_, pc_, gp_, m, order, sums_, w_dev, out, window)
        ctx.meta = (N, npx, kind, float(n_rays), float(ratio))
        mask = m > 0
        ctx.mark_non_differentiable(mask)
        return tuple(out[i] for i in range(9)) + (mask,)

    @staticmethod
    def backward(ctx, *d):
        from .._lib import BlendLoss, call, ptr
        if d[0] is None:
            return (None,) * 14
        if any(x is not None for x in d[1:9]):
            raise NotImplementedError("_BlendStepLossFn: gradients flow through the total only (the other outputs are logged)")
        cb_, c_, pix_, gt_, pc_, gp_, m, order, sums_, w_dev, out, window = ctx.saved_tensors
        N, npx, kind, n_rays, ratio = ctx.meta
        dev = cb_.device
        a = BlendLoss()
        a.cb, a.c, a.pix, a.gt, a.m, a.order, a.sums, a.w_dev, a.out = (ptr(cb_), ptr(c_), ptr(pix_), ptr(gt_), ptr(m), ptr(order),
                                                                           ptr(sums_), ptr(w_dev), ptr(out))
        a.N, a.n_rays, a.trim_ratio = N, n_rays, ratio
        d_total = d[0].contiguous()
        d_cb, d_c, d_pix = torch.empty_like(cb_), torch.empty_like(c_), torch.empty_like(pix_)
        d_err, d_sums = torch.empty(N, device=dev), torch.empty(5, device=dev)
        a.d_total, a.d_cb, a.d_c, a.d_pix, a.d_err, a.d_sums = ptr(d_total), ptr(d_cb), ptr(d_c), ptr(d_pix), ptr(d_err), ptr(d_sums)
        call("nudf_blend_loss_bwd", a)
        d_pc = torch.empty_like(pc_)
        scratch = torch.empty(N, device=dev)
        call("nudf_patch_metric", kind, ptr(pc_), ptr(gp_), ptr(window), N, npx, ptr(scratch), ptr(d_err), ptr(d_pc))
        return (d_cb, d_c, d_pix, None, d_pc, None, None, None, d_sums, None, None, None, None, None)


class _ColorLossFromSumsFn(torch.autograd.Function):
    """second half of the ray-sharded fused ColorLoss: the GLOBAL sums [sum|cb-gt|, sum|c-gt|, D] (already all-reduced,
    packed with the renderer's sums by the caller, dist.py (1)) -> (total, color_base_loss, color_loss); the backward
    differentiates the global loss w.r.t. the LOCAL rays (global denominator), one launch."""

    @staticmethod
    def forward(ctx, cb, c, gt, sums, has_mask, w_dev):
        from .._lib import call, ptr
        ctx.set_materialize_grads(False)
        w_dev = w_dev.detach()
        cb_, c_, gt_ = cb.detach().contiguous(), c.detach().contiguous(), gt.detach().contiguous()
        out = torch.empty(3, device=cb_.device)
        den = torch.empty(1, device=cb_.device)
        call("nudf_color_loss_finish", ptr(sums.detach().contiguous()), 1 if has_mask else 0, 0.0, 0.0, 0.0, ptr(w_dev),
             ptr(out), ptr(den))
        ctx.save_for_backward(cb_, c_, gt_, den, w_dev)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, d0, d1, d2):
        from .._lib import call, ptr
        cb_, c_, gt_, den, w_dev = ctx.saved_tensors
        if d0 is None and d1 is None and d2 is None:
            return (None,) * 6
        z = den.new_zeros(())
        d_out = torch.stack([d if d is not None else z for d in (d0, d1, d2)])
        d_cb, d_c = torch.empty_like(cb_), torch.empty_like(c_)
        call("nudf_color_loss_bwd", ptr(cb_), ptr(c_), ptr(gt_), cb_.numel(), ptr(den), 0.0, 0.0, 0.0, ptr(w_dev),
             ptr(d_out), ptr(d_cb), ptr(d_c))
        return d_cb, d_c, None, None, None, None


def _l1_sum(pred, gt):
    if not (pred.is_cuda and pred.dtype == torch.float32 and gt.shape == pred.shape and gt.dtype == torch.float32):
        from .._lib import NudfError
        raise NudfError("ColorLoss runs on fp32 CUDA tensors (no host path)")
    return _L1SumFn.apply(pred, gt)


class ColorPixelLoss(nn.Module):
    """loss/loss.py:21-44 -- the mask only enters the denominator, never the error."""

    def __init__(self, type='mse'):
        super().__init__()
        self.data_parallel = False

    def forward(self, pred, gt, mask):
        num = _l1_sum(pred, gt)
        if mask is not None:
            den = mask.sum().float()
            if self.data_parallel:
                packed = nudf_dist.all_reduce_sum(torch.stack([num, den]))
                num, den = packed[0], packed[1]
            return num / (den + 1e-4)
        cnt = torch.full((), float(pred.numel()), device=pred.device)      # a fill kernel: safe under graph capture
        if self.data_parallel:
            packed = nudf_dist.all_reduce_sum(torch.stack([num, cnt]))
            num, cnt = packed[0], packed[1]
        return num / cnt


class ColorPatchLoss(nn.Module):
    """loss/loss.py:47-84: per-ray patch error ('ssim' in every shipped conf; 'l1', 'ncc', 'ssd' as in the reference),
    masked, top-30 % errors trimmed."""

    def __init__(self, type='ssim', h_patch_size=3):
        super().__init__()
        if type not in PATCH_TYPES:
            raise ValueError("patch_loss_type %r: one of %s" % (type, sorted(PATCH_TYPES)))
        self.type = type
        self.h_patch_size = h_patch_size
        self.data_parallel = False

    def forward(self, pred, gt, mask, penalize_ratio=0.3):
        error = patch_error(pred, gt, self.h_patch_size, self.type)      # [N]
        m = mask.reshape(-1).bool()
        error = error * m.float()
        if self.data_parallel and nudf_dist.exchanging():
            # order statistics over the whole ray batch: gather errors + masks, trim identically everywhere
            return _global_trimmed_mean(error, m, penalize_ratio)
        err_s, idx = torch.sort(error, descending=True)
        return _trimmed_mean(err_s, m[idx], penalize_ratio)


def _trimmed_mean(err_sorted, mask_sorted, ratio):
    """loss/loss.py:79-84 on the descending-sorted errors: clear the mask of the first int(ratio * mask.sum()) entries, mean
    of the errors still masked.  The reference reads that count on the host (`int(...)`, then boolean indexing): here it
    stays on the device -- floor(ratio * count) in the same float32 arithmetic, a position test and a masked sum -- so the
    step never waits for the GPU and can be captured in a HIP graph.  Same value up to the rounding of the final sum."""
    k = torch.floor(ratio * mask_sorted.sum())                        # python float x int64 tensor -> float32, as there
    pos = torch.arange(err_sorted.shape[0], device=err_sorted.device, dtype=torch.float32)
    keep = (mask_sorted & (pos >= k)).to(err_sorted.dtype)
    return (err_sorted * keep).sum() / keep.sum()


def _global_trimmed_mean(error, m, ratio):
    """order statistics over the WHOLE ray batch: one all-gather of the per-ray (error, mask) pairs, then the same
    trim on every rank; the local rows keep their autograd edge."""
    n, r = error.shape[0], nudf_dist.rank()
    both = nudf_dist.all_gather_rows(torch.stack([error.detach().to(torch.float32), m.to(torch.float32)], dim=1))
    e = both[:, 0].to(error.dtype)
    e = torch.cat([e[:r * n], error, e[(r + 1) * n:]])                    # keep the local autograd edge
    mm = both[:, 1] > 0.5
    es, idx = torch.sort(e, descending=True)
    return _trimmed_mean(es, mm[idx], ratio)


class ColorLoss(nn.Module):
    def __init__(self, color_base_weight, color_weight, color_pixel_weight, color_patch_weight,
                 pixel_loss_type='l1', patch_loss_type='ssim', h_patch_size=3):
        super().__init__()
        # the four weights live in `self.weights` (host floats + a device mirror every consumer reads, see LossWeights);
        # the reference's attribute names stay readable / assignable (properties below)
        self.weights = LossWeights(color_base=color_base_weight, color=color_weight, color_pixel=color_pixel_weight,
                                   color_patch=color_patch_weight)
        self.pixel_func = ColorPixelLoss(pixel_loss_type)
        self.patch_func = ColorPatchLoss(patch_loss_type, h_patch_size)
        self.h_patch_size = h_patch_size

    def set_data_parallel(self, flag=True):
        self.pixel_func.data_parallel = flag
        self.patch_func.data_parallel = flag

    def set_color_weights(self, color_base_weight, color_weight, color_pixel_weight, color_patch_weight):
        self.weights.set(color_base=color_base_weight, color=color_weight, color_pixel=color_pixel_weight,
                         color_patch=color_patch_weight)

    color_base_weight = property(lambda self: self.weights.get("color_base"),
                                 lambda self, v: self.weights.set(color_base=v))
    color_weight = property(lambda self: self.weights.get("color"), lambda self, v: self.weights.set(color=v))
    color_pixel_weight = property(lambda self: self.weights.get("color_pixel"),
                                  lambda self, v: self.weights.set(color_pixel=v))
    color_patch_weight = property(lambda self: self.weights.get("color_patch"),
                                  lambda self, v: self.weights.set(color_patch=v))

    # ---- ray-sharded two-phase form: the caller packs `local_sums` with its other batch-global partial sums into ONE
    # all-reduce (dist.py (1)) and hands the global values to `from_global_sums` -----------------------------------
    def fusable(self, color_base, color, gt_color, color_pixel, patch_colors):
        return (color_base is not None and color is not None and color_pixel is None and patch_colors is None
                and color.is_cuda and color.dtype == torch.float32 and color.shape == gt_color.shape == color_base.shape)

    def blend_fusable(self, color_base, color, gt_color, color_pixel, pixel_mask, patch_colors, gt_patch_colors):
        """the blending step's fused loss (_BlendStepLossFn): all four colour terms present, no pixel mask, fp32 on the GPU,
        single process"""
        ts = (color_base, color, gt_color, color_pixel, patch_colors, gt_patch_colors)
        return (all(t is not None and t.is_cuda and t.dtype == torch.float32 for t in ts) and pixel_mask is None
                and color.shape == gt_color.shape == color_base.shape == color_pixel.shape
                and patch_colors.shape == gt_patch_colors.shape and not self.patch_func.data_parallel)

    def prefill_constants(self, device):
        """the device-resident constants this loss caches on first use (the SSIM Gaussian window), created now: called by
        train.GraphedStep before a capture, inside which the host-to-device copy would be illegal"""
        from .patch_metric import _win_cache, gaussian_window
        key = (self.h_patch_size, str(device))
        if key not in _win_cache:
            _win_cache[key] = gaussian_window(self.h_patch_size).to(device)
        return _win_cache[key]

    def blend_step_loss(self, color_base, color, gt_color, color_pixel, patch_colors, gt_patch_colors, patch_mask_raw,
                        weight_sum, sums, n_rays, w_dev):
        from .patch_metric import PATCH_TYPES, _win_cache
        key = (self.h_patch_size, str(color.device))
        self.prefill_constants(color.device)
        return _BlendStepLossFn.apply(color_base, color, color_pixel, gt_color, patch_colors, gt_patch_colors, patch_mask_raw,
                                      weight_sum, sums, n_rays, w_dev, _win_cache[key], PATCH_TYPES[self.patch_func.type], 0.3)

    def local_sums(self, color_base, color, gt_color, pixel_mask):
        """-> [sum|cb-gt|, sum|c-gt|, mask count (or element count)] of the LOCAL rays, no autograd."""
        from .._lib import call, ptr
        cb_, c_, gt_ = color_base.detach().contiguous(), color.detach().contiguous(), gt_color.detach().contiguous()
        m_ = pixel_mask.detach().float().contiguous() if pixel_mask is not None else None
        sums = torch.empty(3, device=cb_.device)
        call("nudf_color_loss_sums", ptr(cb_), ptr(c_), ptr(gt_), cb_.numel(), ptr(m_), m_.numel() if m_ is not None else 0,
             ptr(sums))
        return sums

    def from_global_sums(self, color_base, color, gt_color, pixel_mask, sums):
        total, lb, lc = _ColorLossFromSumsFn.apply(color_base, color, gt_color, sums, pixel_mask is not None,
                                                   self.weights.device(color.device))
        return {'loss': total, 'color_base_loss': lb, 'color_loss': lc, 'color_pixel_loss': 0.0, 'color_patch_loss': 0.0}

    def forward(self, color_base, color, gt_color, color_pixel, pixel_mask, patch_colors, gt_patch_colors, patch_mask):
        if (color_base is not None and color is not None and color_pixel is None and patch_colors is None
                and color.is_cuda and color.dtype == torch.float32
                and color.shape == gt_color.shape == color_base.shape):
            total, lb, lc = _ColorLossFn.apply(color_base, color, gt_color, pixel_mask,
                                               self.weights.device(color.device), self.pixel_func.data_parallel)
            return {'loss': total, 'color_base_loss': lb, 'color_loss': lc, 'color_pixel_loss': 0.0,
                    'color_patch_loss': 0.0}
        color_base_loss = color_loss = color_pixel_loss = color_patch_loss = 0.0
        if color_base is not None:
            color_base_loss = self.pixel_func(color_base, gt_color, pixel_mask)
        if color is not None:
            color_loss = self.pixel_func(color, gt_color, pixel_mask)
        if color_pixel is not None:
            color_pixel_loss = self.pixel_func(color_pixel, gt_color, patch_mask)
        if patch_colors is not None:
            color_patch_loss = self.patch_func(patch_colors, gt_patch_colors, patch_mask)
        # (loss/loss.py:125-129) with the weights as 0-d views of the device vector: the same expression runs eagerly and
        # inside a captured step, and follows the schedules on replay
        dev = next((t.device for t in (color_base, color, color_pixel, patch_colors) if t is not None), None)
        w = self.weights.device(dev) if dev is not None and dev.type == "cuda" else self.weights.values()
        total_loss = (color_base_loss * w[LW["color_base"]] + color_loss * w[LW["color"]]
                      + color_pixel_loss * w[LW["color_pixel"]]) / w[LW["color_sum"]] \
            + color_patch_loss * w[LW["color_patch"]]
        return {'loss': total_loss, 'color_base_loss': color_base_loss, 'color_loss': color_loss,
                'color_pixel_loss': color_pixel_loss, 'color_patch_loss': color_patch_loss}
