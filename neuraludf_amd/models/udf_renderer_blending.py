"""Drop-in for the reference's `models/udf_renderer_blending.py`: `UDFRendererBlending` with the
same constructor / `render` signature and the same 32-key result dict
(models/udf_renderer_blending.py:107-148, 586-721), driving hand-written HIP kernels:

  coarse samples            nudf_coarse_z / nudf_outside_z          (:605-630)
  importance re-sampling    nudf_upsample + nudf_merge + UDF MLP     (:723-755, 762-866, 66-104, 274-290)
  background                NeRF MLP on the outside samples          (:161-195, see note below)
  render_core               UDF MLP (+ d udf/dx), colour MLP, fused composite kernel (:327-584)

Host code here only allocates buffers, draws the random numbers (so seeding matches the
reference, Appendix B-18 of SURVEY.md) and sequences kernels; there is no PyTorch fallback.

Notes on value-preserving deviations (all verified against the oracle):
  * render_core evaluates the UDF MLP once (the reference runs the identical forward twice, :364/:368);
  * the NeRF is evaluated on the `n_outside` samples only when `color_maps is None` (the
    reference evaluates all S+n_outside points and discards the inside ones, :493-501);
  * NaN guards that block on the host (:97, :265, :543, :860) are a status word in device memory instead: the up-sampling,
    composite and step-loss kernels OR a bit into it when they produce a non-finite sample / weight / loss
    (include/nudf.h: nudf_set_status_flag); `renderer.status()` reads it, `renderer.check_finite()` raises -- when the
    caller chooses to look (Trainer.iteration: every `status_every` iterations), not in the middle of every step;
  * `sparse_random_error` (:681-686, never consumed by the runner) is 0.0 unless
    `renderer.compute_sparse_random = True`.
"""
from __future__ import annotations

import numpy as np
import torch

import os

from .. import _lib
from .._lib import Composite, CompositeGrad, Upsample, call, ptr
from .. import dist as nudf_dist
from .. import mlp

# Arithmetic of the up-sampling kernel (include/nudf.h), all four on by default -- together they reproduce the CPU
# reference's new samples on EVERY ray of BASELINE config 2 when fed the reference's own (z, udf) (0 of 512 rays moved by
# > 1e-4 in each of the four rounds, worst |dz| 7e-7; profiles/r04_upsample_first_diff.txt):
#   512  = NUDF_UP_SERIAL      the three scans accumulated serially in double like torch's CPU cumprod / cumsum;
#   1024 = NUDF_UP_NOCONTRACT  no floating-point contraction (separate torch ops round separately);
#   2048 = NUDF_UP_SLEEF       sigmoid as torch's CPU kernel forms it (Sleef expf u10 restated bit for bit + a true
#                              division): alpha_plus / alpha_minus become bit-identical to the reference's on all rays;
#   4096 = NUDF_UP_EXPCR       the exp call sites (torch.exp = MKL vsExp HA on the CPU) correctly rounded: THE term that
#                              moved rays -- on rays that miss the surface one ulp of exp is the visibility product
#                              (first differing stage on every moved ray: vis_prob, 1-4 ulp, with libm's expf).
# Why any of it matters: such rays' pdf sits on sample_pdf's 1e-5 floor, where 1e-7 of noise is 1 %.  NUDF_UP_FLAGS=0
# restores the wave-parallel fp32 scans and libm transcendentals of rounds 1-2 (what the same ops do on a GPU).
UPSAMPLE_FLAGS = int(os.environ.get("NUDF_UP_FLAGS", str(512 | 1024 | 2048 | 4096)))
# the renderer scalars (inv_s, beta, gamma) formed inside the composite launches instead of by nudf_scalars_fwd / _bwd
# (same expressions; NUDF_FUSE_SCALARS=0 restores the separate launches: A/B and tests)
FUSE_SCALARS = os.environ.get("NUDF_FUSE_SCALARS", "1") == "1"
# no-grad rendering: composite weights first, colour sums inside the colour network's epilogues (render_core); 0 = the
# per-sample colours through memory as in a training step (A/B, tests)
FUSE_COLOUR = os.environ.get("NUDF_FUSE_COLOUR", "1") == "1"

_DIAG = ["alpha", "alpha_plus", "alpha_minus", "vis_prob", "alpha_occ", "raw_occ", "true_cos", "grad_mag", "mid_z",
         "dists", "inside", "flip"]


def _fill_composite(a, c, N, S, n_out):
    a.N, a.S, a.n_out, a.s_nominal = N, S, n_out, c["s_nominal"]
    a.has_anneal = 0 if c["cos_anneal"] is None else 1
    a.cos_anneal = 0.0 if c["cos_anneal"] is None else float(c["cos_anneal"])
    a.flip_saturation = float(c["flip_saturation"])
    a.use_norm_grad = 1 if c["use_norm_grad"] else 0
    a.sparse_scale = float(c["sparse_scale"])
    a.alpha_type = int(c.get("alpha_type", 0))
    a.sched = ptr(c.get("sched"))         # device {cos_anneal_ratio, flip_saturation} of a graph-captured step, or NULL


class _CompositeFn(torch.autograd.Function):
    """fused alpha/transmittance/composite; differentiable w.r.t. udf, grad, colours, background and scalars."""

    @staticmethod
    def forward(ctx, c, rays_o, rays_d, z, sample_dist, background_rgb, udf, grad, color, color_base, bg_z, bg_sigma,
                bg_color, scal, p_var=None, p_beta=None, p_gamma=None):
        """`scal` [3] (clipped inv_s, beta, gamma), or None with the three 1-element parameters `p_var`, `p_beta`,
        `p_gamma` (+ c["beta_hi"]): the kernels then form the scalars themselves and the two extra outputs at the END of
        the result tuple are scal [3] and recip [2] (what nudf_scalars_fwd returned), not differentiable."""
        ctx.set_materialize_grads(False)     # unused outputs (depth, normals, diagnostics ...) arrive as None
        N, S = z.shape
        n_out = 0 if bg_z is None else bg_z.shape[1]
        dev = z.device
        t = lambda x: None if x is None else x.detach().contiguous()
        rays_o, rays_d, z, udf, grad, color, color_base = map(t, (rays_o, rays_d, z, udf, grad, color, color_base))
        bg_z, bg_sigma, bg_color, scal = map(t, (bg_z, bg_sigma, bg_color, scal))
        p_var, p_beta, p_gamma = map(t, (p_var, p_beta, p_gamma))
        a = Composite()
        a.rays_o, a.rays_d, a.z, a.udf, a.grad = ptr(rays_o), ptr(rays_d), ptr(z), ptr(udf), ptr(grad)
        a.color, a.color_base = ptr(color), ptr(color_base)
        a.bg_z, a.bg_sigma, a.bg_color = ptr(bg_z), ptr(bg_sigma), ptr(bg_color)
        a.scal, a.sample_dist, a.background_rgb = ptr(scal), ptr(sample_dist), ptr(t(background_rgb))
        _fill_composite(a, c, N, S, n_out)
        scal_out = recip = None
        if p_var is not None:
            scal_out, recip = torch.empty(3, device=dev), torch.empty(2, device=dev)
            a.p_variance, a.p_beta, a.p_gamma, a.beta_hi = ptr(p_var), ptr(p_beta), ptr(p_gamma), float(c["beta_hi"])
            a.scal_out, a.recip_out = ptr(scal_out), ptr(recip)
        weights = torch.empty(N, S + n_out, device=dev)
        out_color = torch.empty(N, 3, device=dev)
        out_cb = torch.empty(N, 3, device=dev)
        depth = torch.empty(N, 1, device=dev)
        normals = torch.empty(N, 3, device=dev)
        wsum = torch.empty(N, 1, device=dev)
        wsum_all = torch.empty(N, 1, device=dev)
        sums = torch.empty(5, device=dev)                     # assigned by the two-stage reduction (ws given)
        ws = torch.empty(5 * ((N + 3) // 4), device=dev)     # per-block partial sums
        a.ws = ptr(ws)
        a.weights, a.out_color, a.out_color_base = ptr(weights), ptr(out_color), ptr(out_cb)
        a.out_depth, a.out_normals, a.out_wsum, a.out_wsum_all, a.sums = (ptr(depth), ptr(normals), ptr(wsum),
                                                                          ptr(wsum_all), ptr(sums))
        diag = []
        if c["diagnostics"]:
            for k in _DIAG:
                d = torch.empty(N, S, device=dev)
                setattr(a, "o_" + k, ptr(d))
                diag.append(d)
        if c.get("defer_sums"):
            # the consumer's launch finishes the two-stage reduction (loss._StepLossFn, or `finish_sums`), found through a
            # python attribute of the tensor.  A consumer that loses the attribute on the way (detach / clone / .to) must
            # not read uninitialised memory silently: the five sums start as NaN, which the loss kernels' non-finite
            # status bit and any reader then make loud
            a.defer_sums = 1
            sums.fill_(float("nan"))
            sums._nudf_ws = (ws, (N + 3) // 4)
        ST = weights.shape[1]
        mlp.call_timed("composite", "composite_fwd N=%d S=%d" % (N, ST), (48.0 if color is not None else 24.0) * N * ST + 68.0 * N,
                       "nudf_composite_fwd", a, units=N * ST)
        ctx.c = c
        ctx.save_for_backward(rays_o, rays_d, z, sample_dist, t(background_rgb), udf, grad, color, color_base, bg_z,
                              bg_sigma, bg_color, scal, p_var, p_beta, p_gamma)
        extra = () if p_var is None else (scal_out, recip)
        ctx.mark_non_differentiable(*diag, *extra)
        return (out_color, out_cb, weights, depth, normals, wsum, wsum_all, sums) + tuple(diag) + extra

    @staticmethod
    def backward(ctx, d_color, d_cb, d_weights, d_depth, d_normals, d_wsum, d_wsum_all, d_sums, *_):
        (rays_o, rays_d, z, sample_dist, background_rgb, udf, grad, color, color_base, bg_z, bg_sigma, bg_color,
         scal, p_var, p_beta, p_gamma) = ctx.saved_tensors
        c = ctx.c
        N, S = z.shape
        n_out = 0 if bg_z is None else bg_z.shape[1]
        dev = z.device
        a = Composite()
        a.rays_o, a.rays_d, a.z, a.udf, a.grad = ptr(rays_o), ptr(rays_d), ptr(z), ptr(udf), ptr(grad)
        a.color, a.color_base = ptr(color), ptr(color_base)
        a.bg_z, a.bg_sigma, a.bg_color = ptr(bg_z), ptr(bg_sigma), ptr(bg_color)
        a.scal, a.sample_dist, a.background_rgb = ptr(scal), ptr(sample_dist), ptr(background_rgb)
        _fill_composite(a, c, N, S, n_out)
        if p_var is not None:
            a.p_variance, a.p_beta, a.p_gamma, a.beta_hi = ptr(p_var), ptr(p_beta), ptr(p_gamma), float(c["beta_hi"])
        g = CompositeGrad()
        cg = lambda x: None if x is None else x.contiguous()
        keep = [cg(x) for x in (d_color, d_cb, d_weights, d_depth, d_normals, d_wsum, d_wsum_all, d_sums)]
        (g.d_color, g.d_color_base, g.d_weights, g.d_depth, g.d_normals, g.d_wsum, g.d_wsum_all,
         g.d_sums) = [ptr(x) for x in keep]
        o_udf = torch.empty(N, S, device=dev)
        o_grad = torch.empty(N, S, 3, device=dev)
        o_col = torch.empty(N, S, 3, device=dev)
        o_cb = torch.empty(N, S, 3, device=dev)
        o_sig = torch.empty(N, n_out, device=dev) if n_out else None
        o_bgc = torch.empty(N, n_out, 3, device=dev) if n_out else None
        o_scal = o_par = None
        if p_var is not None:
            o_par = torch.empty(3, device=dev)                # d variance, d beta, d gamma: reduction + scalars_bwd, one launch
        else:
            o_scal = torch.empty(3, device=dev)               # assigned (ws given)
        ws = torch.empty(3 * ((N + 3) // 4), device=dev)
        g.ws = ptr(ws)
        g.o_d_udf, g.o_d_grad, g.o_d_color, g.o_d_color_base = ptr(o_udf), ptr(o_grad), ptr(o_col), ptr(o_cb)
        g.o_d_bg_sigma, g.o_d_bg_color, g.o_d_scal, g.o_d_param = ptr(o_sig), ptr(o_bgc), ptr(o_scal), ptr(o_par)
        mlp.call_timed("composite", "composite_bwd N=%d S=%d" % (N, S + n_out), 84.0 * N * (S + n_out) + 68.0 * N,
                       "nudf_composite_bwd", a, g, units=N * (S + n_out))
        d_par = (None, None, None) if o_par is None else (o_par[0:1].reshape(p_var.shape), o_par[1:2].reshape(p_beta.shape),
                                                         o_par[2:3].reshape(p_gamma.shape))
        return (None, None, None, None, None, None, o_udf, o_grad, o_col, o_cb, None, o_sig, o_bgc, o_scal) + d_par


class _ScalarsFn(torch.autograd.Function):
    """(variance, beta, gamma parameters) -> clipped [inv_s, beta, gamma] and [1/inv_s, 1/beta], one launch each way
    (fields.py:654-655, 674-678 + the call-site clips of udf_renderer_blending.py:373-377)."""

    @staticmethod
    def forward(ctx, variance, beta, gamma, beta_hi):
        ctx.set_materialize_grads(False)
        v, b, g = variance.detach().contiguous(), beta.detach().contiguous(), gamma.detach().contiguous()
        scal = torch.empty(3, device=v.device)
        recip = torch.empty(2, device=v.device)
        call("nudf_scalars_fwd", ptr(v), ptr(b), ptr(g), float(beta_hi), ptr(scal), ptr(recip))
        ctx.save_for_backward(v, b, g)
        ctx.beta_hi = float(beta_hi)
        ctx.mark_non_differentiable(recip)
        return scal, recip

    @staticmethod
    def backward(ctx, d_scal, _d_recip):
        if d_scal is None:
            return None, None, None, None
        v, b, g = ctx.saved_tensors
        d = torch.empty(3, device=v.device)
        call("nudf_scalars_bwd", ptr(v), ptr(b), ptr(g), ctx.beta_hi, ptr(d_scal.contiguous()), ptr(d))
        return d[0:1].reshape(v.shape), d[1:2].reshape(b.shape), d[2:3].reshape(g.shape), None


class _ErrorsFn(torch.autograd.Function):
    """[eik_num, eik_den, eikns_num, eikns_den, sparse_sum] -> (gradient_error, gradient_error_near_surface,
    sparse_error) (:531-536, 553), one launch each way.  Three separate 0-d outputs: indexing one [3] tensor would
    cost a zero-fill + scatter per consumer in the backward."""

    @staticmethod
    def forward(ctx, sums, n_rays):
        ctx.set_materialize_grads(False)
        sums = sums.detach().contiguous()
        err = torch.empty(3, device=sums.device)
        call("nudf_sums_errors_fwd", ptr(sums), float(n_rays), ptr(err))
        ctx.save_for_backward(sums)
        ctx.n_rays = float(n_rays)
        return err[0], err[1], err[2]

    @staticmethod
    def backward(ctx, d0, d1, d2):
        if d0 is None and d1 is None and d2 is None:
            return None, None
        (sums,) = ctx.saved_tensors
        z = sums.new_zeros(())
        d_err = torch.stack([d if d is not None else z for d in (d0, d1, d2)])
        d = torch.empty(5, device=sums.device)
        call("nudf_sums_errors_bwd", ptr(sums), ctx.n_rays, ptr(d_err), ptr(d))
        return d, None


GRID_CHUNK_POINTS = 1 << 21      # points per field query: large launches (>= 32768 row tiles) instead of 64^3 blocks


def _grid_query(bound_min, bound_max, resolution, query_func, device, channels):
    """Dense-grid evaluation of a field, GPU-resident.  Same grid as the reference (per-axis
    ``linspace(bound_min, bound_max, resolution)``, x-major ordering) but the points are generated on the device in
    slabs of whole x-planes (up to GRID_CHUNK_POINTS each) and the values land in one device volume that is copied to
    the host once; the reference walks 64^3 blocks with a host meshgrid, an upload and a download per block
    (models/udf_renderer_blending.py:16-49)."""
    dev = torch.device(device)
    R = int(resolution)
    axes = [torch.linspace(float(bound_min[k]), float(bound_max[k]), R, device=dev) for k in range(3)]
    shape = (R, R, R) if channels == 1 else (R, R, R, channels)
    vol = torch.empty(shape, dtype=torch.float32, device=dev)
    planes = max(1, (GRID_CHUNK_POINTS // (1 if channels == 1 else 4)) // (R * R))   # gradient queries keep activations
    yz = torch.stack([axes[1][:, None].expand(R, R), axes[2][None, :].expand(R, R)], -1).reshape(1, R * R, 2)
    for x0 in range(0, R, planes):
        xs = axes[0][x0:x0 + planes]
        pts = torch.cat([xs[:, None, None].expand(len(xs), R * R, 1), yz.expand(len(xs), R * R, 2)], -1).reshape(-1, 3)
        val = query_func(pts).detach()
        vol[x0:x0 + len(xs)] = val.reshape((len(xs),) + shape[1:])
    return vol.cpu().numpy()


def extract_fields(bound_min, bound_max, resolution, query_func, device='cuda'):
    """[R, R, R] float32 numpy volume of ``query_func(pts [P, 3]) -> [P]`` (reference signature, :16)."""
    with torch.no_grad():
        return _grid_query(bound_min, bound_max, resolution, query_func, device, 1)


def extract_gradient_fields(bound_min, bound_max, resolution, query_func, device='cuda'):
    """[R, R, R, 3] volume of a vector field, e.g. ``lambda p: udf_network.gradient(p).squeeze()`` (:33)."""
    return _grid_query(bound_min, bound_max, resolution, query_func, device, 3)


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func, device):
    """(:52-63) field grid (GPU, `extract_fields`) -> marching cubes -> vertices in world units.  The iso-surfacing
    itself is the reference's third-party dependency (PyMCubes) and stays one: imported on use."""
    u = extract_fields(bound_min, bound_max, resolution, query_func, device)
    try:
        import mcubes
    except ImportError as e:          # pragma: no cover - PyMCubes is not in the build image
        raise ImportError("extract_geometry needs PyMCubes (`mcubes`), like the reference; the field grid itself is "
                          "available from extract_fields()") from e
    vertices, triangles = mcubes.marching_cubes(u, threshold)
    b_max = np.asarray([float(v) for v in bound_max], dtype=np.float64)
    b_min = np.asarray([float(v) for v in bound_min], dtype=np.float64)
    vertices = vertices / (resolution - 1.0) * (b_max - b_min)[None, :] + b_min[None, :]
    return vertices, triangles


class UDFRendererBlending:
    def __init__(self, nerf, udf_network, deviation_network, color_network, beta_network, n_samples, n_importance,
                 n_outside, up_sample_steps, perturb, sdf2alpha_type='numerical', upsampling_type='classical',
                 sparse_scale_factor=25000, h_patch_size=3, use_norm_grad_for_cosine=False):
        self.nerf = nerf
        self.udf_network = udf_network
        self.deviation_network = deviation_network
        self.color_network = color_network
        self.beta_network = beta_network
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.perturb = perturb
        self.up_sample_steps = up_sample_steps
        if sdf2alpha_type not in ('numerical', 'theorical'):     # the reference's two branches (:308, :321)
            raise ValueError("sdf2alpha_type %r (the reference knows 'numerical' and 'theorical')" % sdf2alpha_type)
        self.sdf2alpha_type = sdf2alpha_type
        self.upsampling_type = upsampling_type
        self.sparse_scale_factor = sparse_scale_factor
        self.h_patch_size = h_patch_size
        self.use_norm_grad_for_cosine = use_norm_grad_for_cosine
        # knobs that do not exist in the reference (defaults keep its behaviour)
        self.diagnostics = True            # fill the debug keys of the result dict
        self.compute_sparse_random = False
        self.data_parallel = False         # all-reduce the batch-global loss sums over the process group
        self.defer_loss_sums = False       # hand the (local) sums to the caller (key '_loss_sums') instead of the three error
                                           # terms: the ray-sharded step packs them into its one all-reduce, the
                                           # single-process step finishes them inside its fused loss launch (train.Trainer)
        self.defer_sums_reduce = False     # with defer_loss_sums, single process: '_loss_sums' arrives as per-block partials
                                           # (tensor attribute _nudf_ws) that loss._StepLossFn / finish_sums reduce
        from .patch_projector import PatchProjector
        self.patch_projector = PatchProjector(self.h_patch_size)
        self._u_cache = {}
        self._mid = None        # (z_vals, mid points, colour-input buffer) left by _merge_last for render_core

    # ------------------------------------------------------------------------------------
    def _scalars(self, dev):
        """-> (scal [3] = clipped inv_s, beta, gamma ; recip [2] = 1/inv_s, 1/beta)  (:373-377).
        One kernel when the scalar networks are the drop-in ones (1-element `variance`, `beta`, `gamma`
        parameters); any other module goes through its own forward, as the reference does."""
        dn, bn = self.deviation_network, self.beta_network
        if (isinstance(getattr(dn, "variance", None), torch.Tensor) and dn.variance.numel() == 1 and dn.variance.is_cuda
                and isinstance(getattr(bn, "beta", None), torch.Tensor) and hasattr(bn, "gamma")
                and hasattr(bn, "beta_min")):
            return _ScalarsFn.apply(dn.variance, bn.beta, bn.gamma, 1.0 / bn.beta_min)
        inv_s = dn(torch.zeros([1, 3], device=dev))[:, :1].clip(1e-6, 1e6).reshape(1)
        beta = bn.get_beta().clip(1e-6, 1e6).reshape(1)
        gamma = bn.get_gamma().clip(1e-6, 1e6).reshape(1)
        scal = torch.cat([inv_s, beta, gamma])
        return scal, torch.stack([1.0 / inv_s[0], 1.0 / beta[0]]).detach()

    def _scalar_params(self):
        """the three 1-element parameters + 1 / beta_min when the scalar networks are the drop-in ones (the composite
        launch then forms inv_s / beta / gamma itself: no nudf_scalars_fwd / _bwd launches), else None."""
        dn, bn = self.deviation_network, self.beta_network
        # exactly the drop-in classes: the kernel restates THEIR forward / get_beta / get_gamma (exp(10 p), the clips) --
        # a subclass or look-alike with other expressions goes through its own methods (`_scalars`)
        from . import fields as _fields
        if not (FUSE_SCALARS and type(dn) is _fields.SingleVarianceNetwork and type(bn) is _fields.BetaNetwork):
            return None
        ps = (dn.variance, bn.beta, bn.gamma)
        dev = ps[0].device
        if all(isinstance(t, torch.Tensor) and t.numel() == 1 and t.is_cuda and t.device == dev and t.dtype == torch.float32
               and t.is_contiguous() for t in ps):
            return dn.variance, bn.beta, bn.gamma, 1.0 / bn.beta_min
        return None

    @staticmethod
    def finish_sums(sums):
        """second stage of the composite sums when the launch left them as per-block partials (NudfComposite.defer_sums)
        and the consumer is not the fused step loss (which reduces them in its own launch)."""
        pend = getattr(sums, "_nudf_ws", None)
        if pend is not None:
            del sums._nudf_ws
            call("nudf_partial_sums", ptr(pend[0]), pend[1], 5, ptr(sums))
        return sums

    def errors_from_sums(self, sums, n_local):
        """[eik_num, eik_den, eikns_num, eikns_den, sparse_sum] (batch-global when data parallel) -> (gradient_error,
        gradient_error_near_surface, sparse_error); the sparsity mean runs over ALL rays of the batch."""
        n_rays = float(n_local) * (nudf_dist.world_size() if self.data_parallel else 1)
        return _ErrorsFn.apply(self.finish_sums(sums), n_rays)

    def _quantiles(self, k, dev):
        key = (k, str(dev))
        if key not in self._u_cache:
            self._u_cache[key] = torch.linspace(0. + 0.5 / k, 1. - 0.5 / k, steps=k).to(dev).contiguous()
        return self._u_cache[key]

    def _udf_at(self, rays_o, rays_d, z, sample_dist, pts=None):
        N, M = z.shape
        if pts is None:
            pts = torch.empty(N * M, 3, device=z.device)
            call("nudf_ray_points", ptr(rays_o), ptr(rays_d), ptr(z), ptr(sample_dist), N, M, 0, ptr(pts))
        return self.udf_network.udf_only(pts).reshape(N, M)

    def _upsample(self, rays_o, rays_d, z, udf, sample_dist, k, mode, inv_s, beta, gamma, gamma_dev=None, dbg=None,
                  pending=None):
        """`dbg`: optional [N, 7, M] device tensor for the kernel's per-section intermediates (include/nudf.h).
        `pending` = (z_prev, udf_prev, z_add, udf_add): the previous round's merge runs inside this launch (z, udf are
        ignored) and the merged lists are returned as well: -> (z_new, pts_new, z_merged, udf_merged)."""
        dev = rays_o.device
        a = Upsample()
        keep = None
        if pending is not None:
            zp, up, za, ua = keep = [t.contiguous() for t in pending]
            N, M = zp.shape[0], zp.shape[1] + za.shape[1]
            z = torch.empty(N, M, device=dev)
            udf = torch.empty(N, M, device=dev)
            a.prev_z, a.prev_udf, a.add_z, a.add_udf = ptr(zp), ptr(up), ptr(za), ptr(ua)
            a.z_merged, a.udf_merged, a.merge_K = ptr(z), ptr(udf), za.shape[1]
        else:
            N, M = z.shape
        a.rays_o, a.rays_d, a.z, a.udf = ptr(rays_o), ptr(rays_d), ptr(z), ptr(udf)
        a.u, a.sample_dist, a.gamma_dev = ptr(self._quantiles(k, dev)), ptr(sample_dist), ptr(gamma_dev)
        a.N, a.M, a.K, a.mode = N, M, k, mode | (256 if self.sdf2alpha_type == 'theorical' else 0) | UPSAMPLE_FLAGS
        a.inv_s, a.beta, a.gamma = float(inv_s), float(beta), float(gamma)
        z_new = torch.empty(N, k, device=dev)
        pts_new = torch.empty(N * k, 3, device=dev)
        a.z_new, a.pts_new, a.dbg = ptr(z_new), ptr(pts_new), ptr(dbg)
        # algorithmic bytes per ray and round: z, udf in (8 M), quantiles (4 K), the merged lists out (8 (M + K)), ray (24)
        mlp.call_timed("upsample", "upsample N=%d M=%d K=%d" % (N, M, k), float(N) * (8 * M + 4 * k + 8 * (M + k) + 24),
                       "nudf_upsample", a, units=N * M)
        if pending is not None:
            return z_new, pts_new, z, udf
        return z_new, pts_new

    def _merge(self, z, udf, z_new, udf_new):
        N, M = z.shape
        K = z_new.shape[1]
        zo = torch.empty(N, M + K, device=z.device)
        uo = torch.empty(N, M + K, device=z.device) if udf_new is not None else None
        call("nudf_merge", ptr(z), ptr(udf) if uo is not None else None, ptr(z_new), ptr(udf_new), N, M, K, ptr(zo),
             ptr(uo))
        return zo, uo

    def _merge_last(self, rays_o, rays_d, z, z_new, sample_dist, for_render=False):
        """the schedule's last merge (z only) together with what render_core does first with its result: the interval mid
        points, and their copy (+ zero pad) behind the feature columns of the colour network's input rows -- one launch
        instead of merge, ray_points, copy_cols and a fill.  The by-products wait in `self._mid` for render_core."""
        if not for_render:              # a direct call of the sampling schedule: the plain merge, nothing kept
            return self._merge(z, None, z_new, None)[0]
        N, M = z.shape
        K = z_new.shape[1]
        S = M + K
        dev = z.device
        zo = torch.empty(N, S, device=dev)
        pts = torch.empty(N * S, 3, device=dev)
        ceng = self.color_network.engine()
        F, ld = ceng.F, ceng.cin_ld
        feat = None
        if ld >= F + 3:
            feat = torch.empty(mlp.pad_rows(N * S), ld, device=dev)
        call("nudf_merge_points", ptr(z.contiguous()), ptr(z_new), N, M, K, ptr(zo), ptr(rays_o), ptr(rays_d),
             ptr(sample_dist), ptr(pts), (ptr(feat) + 4 * F) if feat is not None else None, ld, ld - F)
        self._mid = (zo, pts, feat)
        return zo

    @torch.no_grad()
    def importance_sample(self, rays_o, rays_d, z_vals, sample_dist, pts0=None, for_render=False):
        """classical schedule (:723-755). `sample_dist` is a 1-element device tensor; `pts0`: the points of `z_vals` if
        the caller has them (nudf_coarse_start).  Each round's merge runs at the head of the next round's launch."""
        N = rays_o.shape[0]
        udf = self._udf_at(rays_o, rays_d, z_vals, sample_dist, pts0)
        steps = self.up_sample_steps
        k = self.n_importance // steps
        pend = None
        for i in range(steps):
            gamma = float(np.clip(20 * 2 ** (steps - i), 20, 320))
            if pend is None:
                z_new, pts_new = self._upsample(rays_o, rays_d, z_vals, udf, sample_dist, k, 0, 64 * 2 ** i,
                                                64 * 2 ** (i + 1), gamma)
            else:
                z_new, pts_new, z_vals, udf = self._upsample(rays_o, rays_d, None, None, sample_dist, k, 0, 64 * 2 ** i,
                                                             64 * 2 ** (i + 1), gamma, pending=pend)
            if i + 1 == steps:
                return self._merge_last(rays_o, rays_d, z_vals, z_new, sample_dist, for_render)
            pend = (z_vals, udf, z_new, self.udf_network.udf_only(pts_new).reshape(N, k))
        return z_vals

    @torch.no_grad()
    def importance_sample_mix(self, rays_o, rays_d, z_vals, sample_dist, pts0=None, for_render=False):
        """mix schedule (:762-832): `steps` not-occlusion-aware rounds + one unbiased round."""
        N = rays_o.shape[0]
        udf = self._udf_at(rays_o, rays_d, z_vals, sample_dist, pts0)
        steps = self.up_sample_steps
        k = self.n_importance // (steps + 1)
        gamma_dev = self.beta_network.get_gamma().clip(1e-6, 1e6).detach().reshape(1).contiguous()
        pend = None
        for i in range(steps):
            if pend is None:
                z_new, pts_new = self._upsample(rays_o, rays_d, z_vals, udf, sample_dist, k, 1, 64 * 2 ** i,
                                                64 * 2 ** (i + 1), 0.0, gamma_dev)
            else:
                z_new, pts_new, z_vals, udf = self._upsample(rays_o, rays_d, None, None, sample_dist, k, 1, 64 * 2 ** i,
                                                             64 * 2 ** (i + 1), 0.0, gamma_dev, pending=pend)
            pend = (z_vals, udf, z_new, self.udf_network.udf_only(pts_new).reshape(N, k))
        i = steps - 1
        if pend is None:
            z_new, _ = self._upsample(rays_o, rays_d, z_vals, udf, sample_dist, k, 0, 64 * 2 ** i, 64 * 2 ** (i + 1),
                                      20 if i < 4 else 10)
        else:
            z_new, _, z_vals, udf = self._upsample(rays_o, rays_d, None, None, sample_dist, k, 0, 64 * 2 ** i,
                                                   64 * 2 ** (i + 1), 20 if i < 4 else 10, pending=pend)
        return self._merge_last(rays_o, rays_d, z_vals, z_new, sample_dist, for_render)

    # ------------------------------------------------------------------------------------
    def render_core_outside(self, rays_o, rays_d, z_out, sample_dist, z_in=None):
        """NeRF++ background (:161-195): -> (sigma [N,n_out], rgb [N,n_out,3], rgb_inside [N,S,3] | None).
        Only the outside samples are evaluated unless `z_in` is given (pixel blending mixes the background
        colour into the inside samples too, :503-506); the outside samples always sort after the inside
        ones (z_out >= far + 1/n_samples), so cat == sort."""
        N, n_out = z_out.shape
        zf = z_out if z_in is None else torch.cat([z_in, z_out], dim=-1).contiguous()
        M = zf.shape[1]
        pts4 = torch.empty(N * M, 4, device=z_out.device)
        call("nudf_ray_points", ptr(rays_o), ptr(rays_d), ptr(zf), ptr(sample_dist), N, M, 2, ptr(pts4))
        sigma, rgb = self.nerf.evaluate(pts4, rays_d, M)
        sigma, rgb = sigma.reshape(N, M), rgb.reshape(N, M, 3)
        if z_in is None:
            return sigma, rgb, None
        S = z_in.shape[1]
        return sigma[:, S:].contiguous(), rgb[:, S:].contiguous(), rgb[:, :S].contiguous()

    def render_core(self, rays_o, rays_d, z_vals, sample_dist, cos_anneal_ratio=None, background_rgb=None,
                    bg_z=None, bg_sigma=None, bg_color=None, flip_saturation=0.0, color_maps=None, w2cs=None,
                    intrinsics=None, query_c2w=None, img_index=None, rays_uv=None, s_nominal=None, bg_color_in=None,
                    patch_cams=None):
        """(:327-584) given sorted samples."""
        N, S = z_vals.shape
        dev = z_vals.device
        P = N * S
        mid, self._mid = getattr(self, "_mid", None), None
        feat_buf = None
        if mid is not None and mid[0] is z_vals:           # nudf_merge_points already produced them (importance sampling)
            pts, feat_buf = mid[1], mid[2]
        else:
            pts = torch.empty(P, 3, device=dev)
            call("nudf_ray_points", ptr(rays_o), ptr(rays_d), ptr(z_vals), ptr(sample_dist), N, S, 1, ptr(pts))
        ceng = self.color_network.engine()
        # (colour-net modes that see the detached unit normal, :371, :425: it rides in the same buffer)
        udf, CIN, grad = self.udf_network.evaluate(pts, want_grad=True, feat_ld=ceng.cin_ld,
                                                   normals_col=(ceng.F + 3) if ceng.nrm else -1, feat_buf=feat_buf)
        # No-grad rendering (validation images, forward-only throughput) without outside samples and blending: the
        # compositing WEIGHTS first (everything of the composite launch but its two colour sums), then the colour network
        # with those weights -- its two sigmoid heads sum weight x colour per 32 points inside their epilogues, so the
        # per-sample colours [P, 3] x 2 are never written or read (:425-429 -> :508-526).  A training
        # step keeps them: its backward reads them (sigmoid', d weights), as the reference's autograd does.
        fuse_colour = (FUSE_COLOUR and not torch.is_grad_enabled() and bg_z is None and color_maps is None and S % 32 == 0
                       and S >= 32 and isinstance(ceng, mlp.ColorEngine) and ceng._chain_ok() and not self.data_parallel)
        cb = col = logits = None
        if not fuse_colour:
            cb, col, logits = self.color_network.evaluate(CIN, rays_d, S)
        spar = self._scalar_params()
        scal = recip = None
        if spar is None:
            scal, recip = self._scalars(dev)
        defer = self.defer_loss_sums and (not self.data_parallel or nudf_dist.exchanging())
        c = dict(s_nominal=(s_nominal if s_nominal is not None else S), cos_anneal=cos_anneal_ratio,
                 beta_hi=(spar[3] if spar is not None else 0.0),
                 # the caller takes the LOCAL sums: single process -> leave the per-block partials to its fused loss launch
                 defer_sums=bool(defer and not self.data_parallel and self.defer_sums_reduce),
                 flip_saturation=flip_saturation, use_norm_grad=self.use_norm_grad_for_cosine,
                 sparse_scale=self.sparse_scale_factor, diagnostics=self.diagnostics,
                 alpha_type=1 if self.sdf2alpha_type == 'theorical' else 0,
                 # device {cos_anneal_ratio, flip_saturation} of a graph-captured step: passed whenever it is set -- with
                 # cos_anneal_ratio = None the kernel still takes flip_saturation from it (has_anneal stays by value)
                 sched=getattr(self, "sched_scalars", None))
        outs = _CompositeFn.apply(c, rays_o, rays_d, z_vals, sample_dist, background_rgb, udf.reshape(N, S),
                                  grad.reshape(N, S, 3), None if fuse_colour else col.reshape(N, S, 3),
                                  None if fuse_colour else cb.reshape(N, S, 3), bg_z, bg_sigma,
                                  bg_color, scal, *(spar[:3] if spar is not None else ()))
        color, color_base, weights, depth, normals, wsum, wsum_all, sums = outs[:8]
        if fuse_colour:
            Pp = mlp.pad_rows(P)
            row_w = weights.reshape(-1)
            if Pp != P:                                       # the chain's last tile reads weights of its pad rows: zeros
                row_w = torch.zeros(Pp, device=dev)
                row_w[:P] = weights.reshape(-1)
            sums_b, sums_c, logits, _ = ceng.forward(CIN.detach(), rays_d, S, P, keep_state=False, row_w=row_w)
            color, color_base = torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev)
            call("nudf_composite_colour_finish", ptr(sums_c), ptr(sums_b), N, S, ptr(background_rgb), ptr(wsum_all), ptr(color),
                 ptr(color_base))
        if spar is not None:
            scal, recip = outs[-2:]
        diag = dict(zip(_DIAG, outs[8:8 + len(_DIAG)])) if self.diagnostics else {}
        local_sums = None
        if defer:
            # ray-sharded step: the caller packs these five LOCAL sums with its other batch-global partial sums into ONE
            # all-reduce and finishes with `errors_from_sums` (train.Trainer.loss; dist.py (1))
            local_sums = sums
            gradient_error = gradient_error_ns = sparse_error = None
        else:
            if self.data_parallel:
                sums = nudf_dist.all_reduce_sum(sums)
            gradient_error, gradient_error_ns, sparse_error = self.errors_from_sums(sums, N)   # (:533, :536, :553)

        color_pixel = patch_colors = patch_mask = None
        if color_maps is not None:
            from . import blend
            if img_index is not None:
                # fields.py:507-508 does torch.index_select(blending_weights [N, S, 10], 1, img_index): it selects along
                # the SAMPLE axis, so the result [N, len(img_index), 10] no longer matches pts_pixel_mask [N, S, V] two
                # lines later (:511) -- the branch cannot run in the reference either; no call site passes img_index
                raise NotImplementedError("img_index: the reference's own branch (fields.py:507-511) is shape-inconsistent "
                                          "and unused; every call site passes None")
            color_pixel, patch_colors, patch_mask = blend.blend_and_composite(
                self.h_patch_size, pts.reshape(N, S, 3), logits.reshape(N, S, -1), weights, grad.reshape(N, S, 3).detach(),
                rays_d, color_maps, w2cs, intrinsics, query_c2w, rays_uv, bg_in=bg_color_in, bg_tail=bg_color,
                patch_cams=patch_cams)

        g3 = grad.reshape(N, S, 3)
        ret = {
            'color_base': color_base, 'color': color, 'color_pixel': color_pixel, 'patch_colors': patch_colors,
            'patch_mask': patch_mask, 'weights': weights, 's_val': recip[0:1].reshape(1, 1),
            'beta': recip[1:2], 'gamma': scal[2:3], 'depth': depth, 'gradient_error': gradient_error,
            'gradient_error_near_surface': gradient_error_ns, 'normals': normals, 'gradients': g3,
            'udf': udf.reshape(N, S), 'sparse_error': sparse_error, 'weight_sum': wsum, 'weight_sum_fg_bg': wsum_all,
        }
        if local_sums is not None:
            ret['_loss_sums'] = local_sums
        if self.diagnostics:
            ret.update({
                'gradients_flip': diag["flip"][:, :, None] * g3, 'inside_sphere': diag["inside"],
                'gradient_mag': diag["grad_mag"], 'true_cos': diag["true_cos"], 'vis_prob': diag["vis_prob"],
                'alpha': diag["alpha"], 'alpha_plus': diag["alpha_plus"], 'alpha_minus': diag["alpha_minus"],
                'mid_z_vals': diag["mid_z"], 'dists': diag["dists"], 'alpha_occ': diag["alpha_occ"],
                'raw_occ': diag["raw_occ"]})
        else:
            for k in ['gradients_flip', 'inside_sphere', 'gradient_mag', 'true_cos', 'vis_prob', 'alpha', 'alpha_plus',
                      'alpha_minus', 'mid_z_vals', 'dists', 'alpha_occ', 'raw_occ']:
                ret[k] = None
        return ret

    # ------------------------------------------------------------------------------------
    # ---- the non-finite status word (replaces the reference's host-side NaN stops, :97-101, 265-269, 543-544, 860-864) ----
    def _device(self):
        return next(self.udf_network.parameters()).device

    def status(self, clear=False):
        """bits seen since the last clear: 1 = non-finite composited ray outputs / renderer scalars, 2 = non-finite new
        samples, 4 = non-finite loss (_lib.STATUS_*).  Reads 4 bytes from the device, i.e. waits for the work enqueued so far."""
        return _lib.read_status(self._device(), clear=clear)

    def clear_status(self):
        _lib.read_status(self._device(), clear=True)

    def check_finite(self):
        """raise FloatingPointError if a kernel has reported a non-finite value since the last clear (the reference drops
        into pdb at that point); clears the word."""
        bits = self.status(clear=True)
        if bits:
            raise FloatingPointError("NeuralUDF renderer: " + _lib.status_text(bits))

    def render(self, rays_o, rays_d, near, far, cos_anneal_ratio=None, perturb_overwrite=-1, background_rgb=None,
               flip_saturation=0, color_maps=None, w2cs=None, intrinsics=None, query_c2w=None, img_index=None,
               rays_uv=None, z_vals_override=None, patch_cams=None):
        dev = rays_o.device
        N = len(rays_o)
        if N == 0:
            return self._empty_result(dev, color_maps is not None, rays_uv is not None)
        _lib.bind_status(dev)          # (one python compare once bound)
        rays_o = rays_o.detach().float().contiguous()
        rays_d = rays_d.detach().float().contiguous()
        if not isinstance(near, torch.Tensor):
            near = torch.tensor([float(near)], device=dev).view(1, 1)
            far = torch.tensor([float(far)], device=dev).view(1, 1)
        near = near.detach().float().contiguous()
        far = far.detach().float().contiguous()
        nf_stride = 0 if near.numel() == 1 else 1

        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        t_rand = None
        lin = None
        if self.n_outside > 0:
            lin = torch.linspace(1e-3, 1.0 - 1.0 / (self.n_outside + 1.0), self.n_outside, device=dev)
        if perturb > 0:
            # drawn on the rays' device: the reference draws on its default device, which the runner makes the GPU
            # (exp_runner_blending.py:872), so the draw order / shapes / generator are the same
            t_rand = torch.rand([N, 1], device=dev)        # (:618); its - 0.5 is applied by the kernel (one launch less)
            if self.n_outside > 0:                                               # (:621-627) stratified jitter
                mids = .5 * (lin[..., 1:] + lin[..., :-1])
                upper = torch.cat([mids, lin[..., -1:]], -1)
                lower = torch.cat([lin[..., :1], mids], -1)
                lin = lower + (upper - lower) * torch.rand(lin.shape, device=dev)
        z_vals = torch.empty(N, self.n_samples, device=dev)
        sample_dist = torch.empty(1, device=dev)
        want_pts0 = z_vals_override is None and self.n_importance > 0 and self.upsampling_type in ('classical', 'mix')
        pts0 = torch.empty(N * self.n_samples, 3, device=dev) if want_pts0 else None
        call("nudf_coarse_start", ptr(near), ptr(far), nf_stride, ptr(t_rand), 1, N, self.n_samples, ptr(z_vals),
             ptr(sample_dist), ptr(rays_o), ptr(rays_d), ptr(pts0))
        z_out = None
        if self.n_outside > 0:
            z_out = torch.empty(N, self.n_outside, device=dev)
            call("nudf_outside_z", ptr(far), nf_stride, ptr(lin.float().contiguous()), N, self.n_outside,
                 self.n_samples, ptr(z_out))
        self._z_vals_outside = z_out      # not in the result dict (the reference's has no such key): read by the jitter parity test

        n_samples = self.n_samples
        if z_vals_override is not None:
            # not part of the reference signature: evaluate GIVEN inside-sphere sample positions [N, S] (sorted) instead
            # of running the hierarchical sampling -- what lets a test hold everything downstream of the (discontinuous)
            # up-sampling against the reference's outputs on the reference's own samples
            z_vals = z_vals_override.detach().float().contiguous()
            n_samples = z_vals.shape[1]
        elif self.n_importance > 0:
            if self.upsampling_type == 'classical':
                z_vals = self.importance_sample(rays_o, rays_d, z_vals, sample_dist, pts0, for_render=True)
            elif self.upsampling_type == 'mix':
                z_vals = self.importance_sample_mix(rays_o, rays_d, z_vals, sample_dist, pts0, for_render=True)
            n_samples = self.n_samples + self.n_importance

        bg_sigma = bg_color = bg_color_in = None
        if self.n_outside > 0:
            bg_sigma, bg_color, bg_color_in = self.render_core_outside(
                rays_o, rays_d, z_out, sample_dist, z_vals if color_maps is not None else None)

        bgrgb = None
        if background_rgb is not None:
            bgrgb = torch.as_tensor(background_rgb, dtype=torch.float32, device=dev).reshape(-1)[:3].contiguous()
        ret = self.render_core(rays_o, rays_d, z_vals, sample_dist, cos_anneal_ratio, bgrgb, z_out, bg_sigma,
                               bg_color, flip_saturation, color_maps, w2cs, intrinsics, query_c2w, img_index, rays_uv,
                               s_nominal=n_samples, bg_color_in=bg_color_in, patch_cams=patch_cams)

        sparse_random_error = 0.0
        if True:
            pts_random = torch.rand([1024, 3], device=dev)                       # keeps the RNG stream aligned (:683)
            if self.compute_sparse_random:
                with torch.no_grad():
                    udf_random = self.udf_network.udf(pts_random * 2 - 1)
                if (udf_random < 0.01).sum() > 10:
                    sparse_random_error = torch.exp(-self.sparse_scale_factor * udf_random[udf_random < 0.01]).mean()
        ret['variance'] = ret.pop('s_val')
        ret['z_vals'] = z_vals
        ret['sparse_random_error'] = sparse_random_error
        return ret

    def _empty_result(self, dev, pixel, patch):
        """the result dict for an empty ray batch (validation chunks can be empty): empty per-ray tensors, zero sums."""
        S = self.n_samples + (self.n_importance if self.n_importance > 0 else 0)
        e = lambda *shape: torch.zeros(*shape, device=dev)
        npx = (2 * self.h_patch_size + 1) ** 2
        ret = {
            'color_base': e(0, 3), 'color': e(0, 3), 'color_pixel': e(0, 3) if pixel else None,
            'patch_colors': e(0, npx, 3) if patch else None, 'patch_mask': e(0) if patch else None,
            'weights': e(0, S + self.n_outside), 'variance': e(1, 1), 'beta': e(1), 'gamma': e(1), 'depth': e(0, 1),
            'gradient_error': e(()), 'gradient_error_near_surface': e(()), 'normals': e(0, 3), 'gradients': e(0, S, 3),
            'udf': e(0, S), 'sparse_error': e(()), 'weight_sum': e(0, 1), 'weight_sum_fg_bg': e(0, 1),
            'z_vals': e(0, S), 'sparse_random_error': 0.0,
        }
        for k in ['gradients_flip', 'inside_sphere', 'gradient_mag', 'true_cos', 'vis_prob', 'alpha', 'alpha_plus',
                  'alpha_minus', 'mid_z_vals', 'dists', 'alpha_occ', 'raw_occ']:
            ret[k] = e(0, S, 3) if k == 'gradients_flip' else e(0, S)
        return ret

    def extract_geometry(self, bound_min, bound_max, resolution, threshold=0.01, device='cpu'):
        """(:757-760) -> (vertices, triangles).  The grid query runs on the network's GPU whatever `device` says (the
        kernels have no CPU path; the reference default 'cpu' only works there for a CPU network)."""
        dev = next(self.udf_network.parameters()).device
        return extract_geometry(bound_min, bound_max, resolution, threshold,
                                lambda pts: self.udf_network.udf(pts)[:, 0], dev)
