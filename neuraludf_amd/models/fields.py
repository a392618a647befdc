"""Drop-in field networks for the reference's `models.fields` (same class names, constructor
signatures, parameter names/shapes/registration order and initialisation, so checkpoints and
Adam state interchange), evaluated by hand-written HIP kernels through libnudf.

  UDFNetwork                 models/fields.py:115-231
  ResidualRenderingNetwork   models/fields.py:400-495   (`RenderingNetwork` is an alias)
  NeRF                       models/fields.py:541-642
  SingleVarianceNetwork      models/fields.py:645-655
  BetaNetwork                models/fields.py:658-700
  color_blend                models/fields.py:498-537   (see models/blend.py kernels)

There is NO PyTorch fallback for the arithmetic: calling these modules without libnudf.so or
on CPU tensors raises `NudfError`.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .embedder import get_embedder
from .. import mlp
from .._lib import NudfError  # noqa: F401


def _wn(lin, enable):
    return nn.utils.weight_norm(lin) if enable else lin


# ------------------------------------------------------------------------------------------
# autograd plumbing: one Function per engine; forward/backward are libnudf kernel chains
# ------------------------------------------------------------------------------------------
class _UDFEvalFn(torch.autograd.Function):
    """(x, *params) -> (udf [P], featbuf [P, ld], grad [P,3]).
    featbuf holds the F appearance features in cols 0..F-1; when `feat_ld` > F it is laid out as the
    colour network's base input ([feat F | pts 3 | zero pad], see mlp.ColorEngine)."""

    @staticmethod
    def forward(ctx, engine, x, want_grad, feat_ld, normals_col, feat_buf, *params):
        ctx.set_materialize_grads(False)     # unused outputs arrive as None, not as zero-filled tensors
        x = x.detach().contiguous()
        need_state = any(ctx.needs_input_grad)      # all False under no_grad (grad mode is off inside forward)
        st = engine.forward(x, need_grad_state=(need_state or want_grad), feat_ld=feat_ld, feat_buf=feat_buf)
        g = DA = None
        if want_grad:
            g, DA = engine.gradient(x, st)
            if normals_col >= 0:
                # colour-net modes that see the normal (fields.py:456-461): gradients / (|gradients| + 1e-5)
                # (udf_renderer_blending.py:371) and its negative go behind the points of the colour net's base input.
                # The reference DETACHES them, so they are constants of this evaluation: written here, no adjoint.
                gn = g / (torch.linalg.norm(g, ord=2, dim=-1, keepdim=True) + 1e-5)
                st["feat"][:, normals_col:normals_col + 3] = gn
                st["feat"][:, normals_col + 3:normals_col + 6] = -gn
        ctx.engine, ctx.x = engine, x
        ctx.st, ctx.DA = (st if need_state else None), (DA if need_state else None)
        # udf_type 'square': d udf / dx = 2 h0 grad h0 depends on h0 a second time (f'' = 2); the backward needs the
        # forward gradient for that term
        ctx.g = g if (need_state and engine.head_type == 1) else None
        return st["udf"], st["feat"], (g if g is not None else x.new_zeros(0))

    @staticmethod
    def backward(ctx, d_udf, d_feat, d_g):
        engine, st = ctx.engine, ctx.st
        if d_udf is None and d_feat is None and (d_g is None or d_g.numel() == 0):
            return (None, None, None, None, None, None) + (None,) * len(engine.params())
        if st is None:
            raise RuntimeError("UDF evaluation was run without gradient state")
        if d_g is not None and d_g.numel() == 0:
            d_g = None
        ldf = 0
        if d_feat is not None:
            d_feat = d_feat.contiguous()
            ldf = d_feat.shape[1]
        if ctx.g is not None and d_g is not None:
            # g = f'(h0) grad_u h0 with f' = 2 h0 (the stored multiplier): h0_bar += f''(h0) (g_bar . grad_u h0)
            # = 2 (g_bar . g) / f'.  The head's adjoint is f' * d_udf / scale, so the term rides in d_udf.
            mult = st["sign"]
            dot = (d_g * ctx.g).sum(-1)
            # dot = mult * (g_bar . grad_u h0) is a product divided by its own factor (well conditioned); dividing twice
            # instead of by mult * mult keeps the intermediate out of the denormals near the surface (h0 -> 0, where
            # samples concentrate: mult^2 underflows at |h0| ~ 1e-19 and the term became inf * 0 = NaN).
            ok = mult.abs() > 1e-30
            safe = torch.where(ok, mult, torch.ones_like(mult))
            extra = torch.where(ok, 2.0 * (dot / safe) / safe, torch.zeros_like(dot)) * float(engine.net.scale)
            d_udf = extra if d_udf is None else d_udf.reshape(-1) + extra
        ctx.g = None
        grads = engine.backward(ctx.x, st, ctx.DA, d_udf.contiguous() if d_udf is not None else None,
                                d_feat, ldf, d_g.contiguous() if d_g is not None else None)
        ctx.st = ctx.DA = None
        return (None, None, None, None, None, None) + tuple(grads)


class UDFNetwork(nn.Module):
    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, scale=1, bias=0.5,
                 geometric_init=True, weight_norm=True, udf_type='abs',
                 udf_shift=None, predict_grad=None):   # the garment confs pass these two; the reference ignores/chokes
        super().__init__()
        if udf_type not in ('abs', 'square', 'sdf'):       # the reference's three udf_out branches (fields.py:184-190)
            raise ValueError("udf_type %r (the reference knows 'abs', 'square' and 'sdf')" % udf_type)
        dims = [d_in] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.embed_fn_fine = None
        self.multires = multires
        self.d_in = d_in
        if multires > 0:
            embed_fn, input_ch = get_embedder(multires, input_dims=d_in)
            self.embed_fn_fine = embed_fn
            dims[0] = input_ch
        self.embed_dim = dims[0]
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        self.geometric_init = geometric_init
        self.udf_type = udf_type
        # identical construction + init order to fields.py:148-178 so torch.manual_seed gives identical weights
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:
                if l == self.num_layers - 2:
                    torch.nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    torch.nn.init.constant_(lin.bias, -bias)
                elif multires > 0 and l == 0:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                    torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif multires > 0 and l in self.skip_in:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            setattr(self, "lin" + str(l), _wn(lin, weight_norm))
        self._engine = None

    # -- HIP evaluation ------------------------------------------------------------------
    def engine(self):
        if self._engine is None:
            self._engine = mlp.UDFEngine(self)
        return self._engine

    def invalidate(self):
        """drop the packed-weight caches after an in-place write through `.data` (which does not bump the
        parameters' version counters, the cache key): see mlp.PackedLinear.invalidate."""
        if self._engine is not None:
            self._engine.invalidate()

    def evaluate(self, x, want_grad=True, feat_ld=0, normals_col=-1, feat_buf=None):
        """fused value + spatial gradient: -> (udf [P], featbuf [P, max(feat_ld, F)], grad [P,3] or empty).
        normals_col >= 0: also write the detached unit normal and its negative at these 6 columns of featbuf.
        feat_buf: a [pad_rows(P), feat_ld] buffer whose columns F.. already hold [x | 0] (nudf_merge_points wrote them):
        used as featbuf, the copy of x and the pad fill are skipped."""
        eng = self.engine()
        return _UDFEvalFn.apply(eng, x, want_grad, feat_ld, normals_col, feat_buf, *eng.params())

    @property
    def n_feature(self):
        return getattr(self, "lin" + str(self.num_layers - 2)).bias.shape[0] - 1

    def udf_only(self, x):
        """no-grad udf [P] (importance sampling, fields.py:731): skips the 256 feature channels."""
        eng = self.engine()
        with torch.no_grad():
            return eng.forward(x.detach().contiguous(), need_grad_state=False, udf_only=True)["udf"]

    def forward(self, inputs):
        udf, feat, _ = self.evaluate(inputs, want_grad=False)
        return torch.cat([udf[:, None], feat[:, :self.n_feature]], dim=-1)

    def udf(self, x):
        udf, _, _ = self.evaluate(x, want_grad=False)
        return udf[:, None]

    def udf_hidden_appearance(self, x):
        return self.forward(x)

    def gradient(self, x):
        _, _, g = self.evaluate(x, want_grad=True)
        return g.unsqueeze(1)


class SDFNetwork(UDFNetwork):
    """`SDFNetwork` of the reference (fields.py:10-112; imported by exp_runner_blending.py:16, never instantiated there):
    the same MLP as UDFNetwork with the identity head -- constructor arguments, parameter names / shapes / registration
    order and the geometric initialisation (incl. `inside_outside`, :50-56) as there, so `torch.manual_seed` gives the same
    weights and `state_dict`s interchange; evaluated by the same HIP chains (`udf_type='sdf'`: value, features, d sdf / d x
    and their double backward), no torch path."""

    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, bias=0.5, scale=1, geometric_init=True,
                 weight_norm=True, inside_outside=False):
        super().__init__(d_in, d_out, d_hidden, n_layers, skip_in=skip_in, multires=multires, bias=bias, scale=scale,
                         geometric_init=geometric_init, weight_norm=weight_norm, udf_type='sdf')
        if geometric_init and inside_outside:
            # cameras inside the scene (:54-56): the last layer's mean and bias change sign.  The draw above consumed the
            # generator exactly like the reference's normal_(mean=+sqrt(pi)/sqrt(d)): a normal_ with the negated mean is the
            # mirrored sample about 0 of the same draw, i.e. -(w - m) - m ... written on the un-normalised direction
            # tensor (weight_v under weight_norm; g = |v| is re-derived), which is what the reference initialises too.
            lin = getattr(self, "lin" + str(self.num_layers - 2))
            m = float(np.sqrt(np.pi) / np.sqrt(lin.weight_v.shape[1] if hasattr(lin, "weight_v") else lin.weight.shape[1]))
            with torch.no_grad():
                w = lin.weight_v if hasattr(lin, "weight_v") else lin.weight
                w.copy_((w - m) - m)                     # N(+m, s) sample -> the N(-m, s) sample of the same draw
                if hasattr(lin, "weight_g"):
                    lin.weight_g.copy_(w.norm(dim=1, keepdim=True))
                lin.bias.fill_(bias)
            self.invalidate()
        self.inside_outside = inside_outside

    def sdf(self, x):
        return self.udf(x)

    def sdf_hidden_appearance(self, x):
        return self.forward(x)


# ------------------------------------------------------------------------------------------
class _ColorFn(torch.autograd.Function):
    """(CIN = [feat F | pts 3 | pad] buffer, rays_d, *params) -> (color_base, color, logits)."""

    @staticmethod
    def forward(ctx, engine, CIN, rays_d, S, *params):
        ctx.set_materialize_grads(False)
        P = CIN.shape[0]
        need = any(ctx.needs_input_grad)
        cb, col, logits, st = engine.forward(CIN.detach(), rays_d.detach().contiguous(), S, P, keep_state=need)
        ctx.engine, ctx.st = engine, st
        ctx.save_for_backward(cb, col)
        if logits is None:
            logits = cb.new_zeros(0)
        return cb, col, logits

    @staticmethod
    def backward(ctx, d_cb, d_col, d_logits):
        cb, col = ctx.saved_tensors
        if d_logits is not None and d_logits.numel() == 0:
            d_logits = None
        grads, dCIN = ctx.engine.backward(ctx.st, cb, col,
                                          d_cb.contiguous() if d_cb is not None else None,
                                          d_col.contiguous() if d_col is not None else None,
                                          d_logits.contiguous() if d_logits is not None else None)
        ctx.st = None
        return (None, dCIN, None, None) + tuple(grads)


class ResidualRenderingNetwork(nn.Module):
    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires_view=0,
                 squeeze_out=True, blending_cand_views=10):
        super().__init__()
        self.mode = mode
        self.squeeze_out = squeeze_out
        self.d_out, self.d_feature, self.d_hidden = d_out, d_feature, d_hidden
        self.multires_view = multires_view
        dims_base = [d_in - 3 + d_feature] + [d_hidden for _ in range(n_layers)] + [d_out]
        dims = [d_hidden + d_out + 3] + [d_hidden for _ in range(n_layers)] + [d_out + blending_cand_views]
        self.embedview_fn = None
        self.view_dim = 3
        if multires_view > 0 and self.mode != 'no_view_dir':
            embedview_fn, input_ch = get_embedder(multires_view)
            self.embedview_fn = embedview_fn
            dims[0] += (input_ch - 3)
            self.view_dim = input_ch
        self.num_layers = len(dims)
        # same creation order as fields.py:429-446: all lin*, then all lin_base*
        for l in range(self.num_layers - 1):
            setattr(self, "lin" + str(l), _wn(nn.Linear(dims[l], dims[l + 1]), weight_norm))
        for l in range(self.num_layers - 1):
            setattr(self, "lin_base" + str(l), _wn(nn.Linear(dims_base[l], dims_base[l + 1]), weight_norm))
        self.if_blending = blending_cand_views > 0
        self._engine = None

    def engine(self):
        if self._engine is None:
            self._engine = mlp.ColorEngine(self)
        return self._engine

    def invalidate(self):
        """drop the packed-weight caches after an in-place write through `.data` (which does not bump the
        parameters' version counters, the cache key): see mlp.PackedLinear.invalidate."""
        if self._engine is not None:
            self._engine.invalidate()

    def evaluate(self, CIN, rays_d, S):
        """CIN: [P, pad(F+3(+6))] = [feat | pts (| n | -n) | 0]: UDFNetwork.evaluate(feat_ld=engine.cin_ld) produces
        [feat | pts | 0], with normals_col = F + 3 also the unit normals of the modes that use them."""
        eng = self.engine()
        return _ColorFn.apply(eng, CIN, rays_d, S, *eng.params())

    def forward(self, points, normals, view_dirs, feature_vectors):
        """reference call surface (fields.py:452-495); view_dirs is per point here."""
        eng = self.engine()
        P = points.shape[0]
        cols = [feature_vectors, points.detach()]
        if eng.nrm:                                     # every mode but 'no_normal' (fields.py:459-461): detached normals
            nd = normals.detach()
            cols += [nd, -nd]
        cols.append(points.new_zeros(P, eng.cin_ld - eng.F - 3 - eng.nrm))
        CIN = torch.cat(cols, dim=1)                    # layout plumbing
        cb, col, logits = _ColorFn.apply(eng, CIN, view_dirs.contiguous(), 1, *eng.params())
        if self.if_blending:
            return cb, col, logits
        return cb, col


class _PlainColorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, X, *params):
        ctx.set_materialize_grads(False)
        need = any(ctx.needs_input_grad)
        color, logits, st = engine.forward(X.detach().float().contiguous(), X.shape[0], keep_state=need)
        ctx.engine, ctx.st = engine, st
        ctx.save_for_backward(color)
        if logits is None:
            logits = color.new_zeros(0)
        return color, logits

    @staticmethod
    def backward(ctx, d_color, d_logits):
        (color,) = ctx.saved_tensors
        if d_logits is not None and d_logits.numel() == 0:
            d_logits = None
        grads, dX = ctx.engine.backward(ctx.st, color, d_color, d_logits)
        ctx.st = None
        return (None, dX) + tuple(grads)


class RenderingNetwork(nn.Module):
    """fields.py:325-397: the plain (NeuS-style) colour MLP.  The runner instantiates ResidualRenderingNetwork; this
    class keeps the reference name, constructor, state_dict and forward() contract for the three modes."""

    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires_view=0,
                 squeeze_out=True, blending_cand_views=0):
        super().__init__()
        self.mode, self.squeeze_out, self.d_out = mode, squeeze_out, d_out
        dims = [d_in + d_feature] + [d_hidden for _ in range(n_layers)] + [d_out + blending_cand_views]
        self.embedview_fn = None
        if multires_view > 0 and self.mode != 'no_view_dir':
            self.embedview_fn, input_ch = get_embedder(multires_view)
            dims[0] += (input_ch - 3)
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            setattr(self, "lin" + str(l), _wn(nn.Linear(dims[l], dims[l + 1]), weight_norm))
        self.relu = nn.ReLU()
        self.if_blending = blending_cand_views > 0
        self._engine = None

    def forward(self, points, normals, view_dirs, feature_vectors):
        if self._engine is None:
            self._engine = mlp.PlainColorEngine(self)
        if self.embedview_fn is not None:
            view_dirs = self.embedview_fn(view_dirs)
        normals = normals.detach() if normals is not None else None
        if self.mode == 'idr':
            x = torch.cat([points, view_dirs, normals, -1 * normals, feature_vectors], dim=-1)
        elif self.mode == 'no_view_dir':
            x = torch.cat([points, normals, -1 * normals, feature_vectors], dim=-1)
        elif self.mode == 'no_normal':
            x = torch.cat([points, view_dirs, feature_vectors], dim=-1)
        else:
            raise ValueError("unknown RenderingNetwork mode %r" % (self.mode,))
        color, logits = _PlainColorFn.apply(self._engine, x, *self._engine.params())
        return (color, logits) if self.if_blending else color


# ------------------------------------------------------------------------------------------
class _NerfFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, pts4, rays_d, S, *params):
        ctx.set_materialize_grads(False)
        P = pts4.shape[0]
        need = any(ctx.needs_input_grad)
        sigma, rgb, st = engine.forward(pts4.detach().contiguous(), rays_d.detach().contiguous(), S, P, keep_state=need)
        ctx.engine, ctx.st = engine, st
        return sigma, rgb

    @staticmethod
    def backward(ctx, d_sigma, d_rgb):
        P = ctx.st["P"]
        dev = ctx.st["VIN"].device
        if d_sigma is None:
            d_sigma = torch.zeros(P, 1, device=dev)
        if d_rgb is None:
            d_rgb = torch.zeros(P, 3, device=dev)
        grads = ctx.engine.backward(ctx.st, d_sigma.contiguous(), d_rgb.contiguous())
        ctx.st = None
        return (None, None, None, None) + tuple(grads)


class NeRF(nn.Module):
    def __init__(self, D=8, W=256, d_in=3, d_in_view=3, multires=0, multires_view=0, output_ch=4, skips=[4],
                 use_viewdirs=False, occupancy=True):
        super().__init__()
        self.D, self.W, self.d_in, self.d_in_view = D, W, d_in, d_in_view
        self.input_ch, self.input_ch_view = 3, 3
        self.multires, self.multires_view = multires, multires_view
        self.embed_fn = self.embed_fn_view = None
        self.occupancy = occupancy
        if multires > 0:
            self.embed_fn, self.input_ch = get_embedder(multires, input_dims=d_in)
        if multires_view > 0:
            self.embed_fn_view, self.input_ch_view = get_embedder(multires_view, input_dims=d_in_view)
        self.skips = list(skips)
        self.use_viewdirs = use_viewdirs
        self.pts_linears = nn.ModuleList(
            [nn.Linear(self.input_ch, W)] +
            [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.input_ch, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(self.input_ch_view + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        self._engine = None

    def engine(self):
        if self._engine is None:
            self._engine = mlp.NerfEngine(self)
        return self._engine

    def invalidate(self):
        """drop the packed-weight caches after an in-place write through `.data` (which does not bump the
        parameters' version counters, the cache key): see mlp.PackedLinear.invalidate."""
        if self._engine is not None:
            self._engine.invalidate()

    def evaluate(self, pts4, rays_d, S):
        eng = self.engine()
        return _NerfFn.apply(eng, pts4, rays_d, S, *eng.params())

    def forward(self, input_pts, input_views):
        """fields.py:599-630.  `use_viewdirs=False` ends in `assert False` in the reference's own forward (:629-630): the
        same here (the constructor still registers `output_linear`, so state dicts interchange).  `input_views=None`
        returns the density only (:614-617): the colour branch is evaluated on zero directions and dropped."""
        assert self.use_viewdirs, "NeRF(use_viewdirs=False): the reference's forward asserts False here (fields.py:629-630)"
        if input_views is None:
            zeros = torch.zeros(input_pts.shape[0], self.d_in_view, device=input_pts.device)
            return self.evaluate(input_pts, zeros, 1)[0]
        return self.evaluate(input_pts, input_views.contiguous(), 1)


# ------------------------------------------------------------------------------------------
class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val, requires_grad=True):
        super().__init__()
        self.variance = nn.Parameter(torch.Tensor([init_val]), requires_grad=requires_grad)

    def set_trainable(self):
        self.variance.requires_grad = True

    def forward(self, x):
        # 1-element parameter transform (fields.py:654-655): host-side plumbing, not per-sample work
        return torch.ones([len(x), 1], device=x.device) * torch.exp(self.variance * 10.0)


class BetaNetwork(nn.Module):
    def __init__(self, init_var_beta=0.1, init_var_gamma=0.1, init_var_zeta=0.05, beta_min=0.00005,
                 requires_grad_beta=True, requires_grad_gamma=True, requires_grad_zeta=True):
        super().__init__()
        self.beta = nn.Parameter(torch.Tensor([init_var_beta]), requires_grad=requires_grad_beta)
        self.gamma = nn.Parameter(torch.Tensor([init_var_gamma]), requires_grad=requires_grad_gamma)
        self.zeta = nn.Parameter(torch.Tensor([init_var_zeta]), requires_grad=requires_grad_zeta)
        self.beta_min = beta_min

    def get_beta(self):
        return torch.exp(self.beta * 10).clip(0, 1. / self.beta_min)

    def get_gamma(self):
        return torch.exp(self.gamma * 10)

    def get_zeta(self):
        return self.zeta.abs()

    def set_beta_trainable(self):
        self.beta.requires_grad = True

    @torch.no_grad()
    def set_gamma(self, x):
        self.gamma = nn.Parameter(torch.Tensor([x]), requires_grad=self.gamma.requires_grad).to(self.gamma.device)

    def forward(self):
        return self.get_beta(), self.get_gamma(), self.get_zeta()
