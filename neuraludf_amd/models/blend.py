"""Image-based blending branch of render_core (models/udf_renderer_blending.py:431-480, 503-524):
pixel warp + view softmax + composite, and the fused patch warp / blend / composite, on the
`nudf_pixel_*` / `nudf_patch_*` HIP kernels.  Host code builds the 8 tiny camera matrices (3x3 / 4x4
inverses and products on [V,4,4] tensors, as the reference does at patch_projector.py:78-97) and wires
autograd; gradients reach the blending logits and the compositing weights."""
from __future__ import annotations

import torch

from .._lib import PatchBlend, PixelBlend, PixelComposite, call, ptr
from ..mlp import call_timed


def _c(t):
    return None if t is None else t.detach().float().contiguous()


def _img(t):
    """source images [V,3,H,W] -> (tensor whose memory the kernel reads, its [V,3,H,W] shape, layout flag).
    The reference hands over `images[src_idx].permute(0, 3, 1, 2)` (dataset.py:147-149), i.e. channel-interleaved
    memory behind an NCHW view: the kernels read that layout directly (one 12-byte load per texel, no 100 MB
    re-layout copy per iteration); a contiguous NCHW tensor is read as three planes."""
    t = t.detach().float()
    if not t.is_contiguous():
        hwc = t.permute(0, 2, 3, 1)
        if hwc.is_contiguous():
            return hwc, tuple(t.shape), 1
        t = t.contiguous()
    return t, tuple(t.shape), 0


class _PixelBlendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, logits, proj, imgs):
        pts, logits, proj = _c(pts), _c(logits), _c(proj)
        imgs, (V, _, H, W), layout = _img(imgs)
        P = pts.shape[0]
        a = PixelBlend()
        a.pts, a.logits, a.nl, a.proj, a.imgs = ptr(pts), ptr(logits), logits.shape[1], ptr(proj), ptr(imgs)
        a.P, a.V, a.H, a.W, a.img_layout = P, V, H, W, layout
        pix = torch.empty(P, 3, device=pts.device)
        a.pix = ptr(pix)
        # algorithmic bytes of the gathers: 4 texels x 12 B per (point, view) tap
        call_timed("pixel_blend", "pixel_blend_fwd P=%d V=%d" % (P, V), 48.0 * P * V, "nudf_pixel_blend_fwd", a, units=P * V)
        ctx.save_for_backward(pts, logits, proj, imgs)
        ctx.img = (V, H, W, layout)
        return pix

    @staticmethod
    def backward(ctx, d_pix):
        pts, logits, proj, imgs = ctx.saved_tensors
        P = pts.shape[0]
        V, H, W, layout = ctx.img
        a = PixelBlend()
        a.pts, a.logits, a.nl, a.proj, a.imgs = ptr(pts), ptr(logits), logits.shape[1], ptr(proj), ptr(imgs)
        a.P, a.V, a.H, a.W, a.img_layout = P, V, H, W, layout
        d_logits = torch.empty_like(logits)
        call_timed("pixel_blend", "pixel_blend_bwd P=%d V=%d" % (P, V), 48.0 * P * V, "nudf_pixel_blend_bwd", a,
                   ptr(d_pix.contiguous()), ptr(d_logits), units=P * V)
        return None, d_logits, None, None


class _PixelCompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, pix, pts, bg_in, bg_tail):
        w, pix, pts, bg_in, bg_tail = _c(w), _c(pix), _c(pts), _c(bg_in), _c(bg_tail)
        N, S = pix.shape[0], pix.shape[1]
        n_out = w.shape[1] - S
        a = PixelComposite()
        a.w, a.pix, a.pts, a.bg_in, a.bg_tail = ptr(w), ptr(pix), ptr(pts), ptr(bg_in), ptr(bg_tail)
        a.N, a.S, a.n_out = N, S, n_out
        out = torch.empty(N, 3, device=w.device)
        a.out = ptr(out)
        call("nudf_pixel_composite_fwd", a)
        ctx.save_for_backward(w, pix, pts, bg_in, bg_tail)
        return out

    @staticmethod
    def backward(ctx, d_out):
        w, pix, pts, bg_in, bg_tail = ctx.saved_tensors
        N, S = pix.shape[0], pix.shape[1]
        a = PixelComposite()
        a.w, a.pix, a.pts, a.bg_in, a.bg_tail = ptr(w), ptr(pix), ptr(pts), ptr(bg_in), ptr(bg_tail)
        a.N, a.S, a.n_out = N, S, w.shape[1] - S
        d_w = torch.empty_like(w)
        d_pix = torch.empty_like(pix)
        d_in = torch.empty_like(bg_in) if bg_in is not None else None
        d_tail = torch.empty_like(bg_tail) if bg_tail is not None else None
        call("nudf_pixel_composite_bwd", a, ptr(d_out.contiguous()), ptr(d_w), ptr(d_pix), ptr(d_in), ptr(d_tail))
        return d_w, d_pix, None, d_in, d_tail


def _fill_patch(a, pts, grad, rays_d, uv, logits, w, ref_cam, src_cam, imgs, hps, img):
    N, S = pts.shape[0], pts.shape[1]
    V, H, W, a.img_layout = img
    a.pts, a.grad, a.rays_d, a.uv = ptr(pts), ptr(grad), ptr(rays_d), ptr(uv)
    a.logits, a.nl, a.w, a.ldw = ptr(logits), logits.shape[-1], ptr(w), w.shape[1]
    a.ref_cam, a.src_cam, a.imgs = ptr(ref_cam), ptr(src_cam), ptr(imgs)
    a.N, a.S, a.V, a.H, a.W, a.hps = N, S, V, H, W, hps


class _PatchBlendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, grad, rays_d, uv, logits, w, ref_cam, src_cam, imgs, hps):
        ctx.set_materialize_grads(False)
        pts, grad, rays_d, uv, logits, w, ref_cam, src_cam = map(_c, (pts, grad, rays_d, uv, logits, w, ref_cam, src_cam))
        imgs, (V, _, H, W), layout = _img(imgs)
        ctx.img = (V, H, W, layout)
        N = pts.shape[0]
        npx = (2 * hps + 1) ** 2
        a = PatchBlend()
        _fill_patch(a, pts, grad, rays_d, uv, logits, w, ref_cam, src_cam, imgs, hps, ctx.img)
        pc = torch.empty(N, npx, 3, device=pts.device)
        pm = torch.empty(N, device=pts.device)
        a.patch_colors, a.patch_mask = ptr(pc), ptr(pm)
        S = pts.shape[1]
        taps = float(N) * S * V * npx          # one bilinear tap = 4 texels x 12 B
        call_timed("patch_blend", "patch_blend_fwd N=%d S=%d V=%d Npx=%d" % (N, S, V, npx), 48.0 * taps, "nudf_patch_blend_fwd", a,
                   units=taps)
        ctx.hps = hps
        ctx.save_for_backward(pts, grad, rays_d, uv, logits, w, ref_cam, src_cam, imgs)
        ctx.mark_non_differentiable(pm)      # only ever thresholded by the caller (exp_runner_blending.py:314)
        return pc, pm

    @staticmethod
    def backward(ctx, d_pc, _d_pm):
        if d_pc is None:
            return (None,) * 10
        pts, grad, rays_d, uv, logits, w, ref_cam, src_cam, imgs = ctx.saved_tensors
        N, S = pts.shape[0], pts.shape[1]
        a = PatchBlend()
        _fill_patch(a, pts, grad, rays_d, uv, logits, w, ref_cam, src_cam, imgs, ctx.hps, ctx.img)
        d_logits = torch.empty_like(logits)
        d_ws = torch.empty(N, S, device=pts.device)
        V, npx = ctx.img[0], (2 * ctx.hps + 1) ** 2
        taps = float(N) * S * V * npx          # the backward recomputes the gathers
        call_timed("patch_blend", "patch_blend_bwd N=%d S=%d V=%d Npx=%d" % (N, S, V, npx), 48.0 * taps, "nudf_patch_blend_bwd", a,
                   ptr(d_pc.contiguous()), ptr(d_logits), ptr(d_ws), units=taps)
        if w.shape[1] == S:                    # no outside samples: the inside weights ARE the weights (no fill + copy)
            d_w = d_ws
        else:
            d_w = torch.zeros_like(w)
            d_w[:, :S] = d_ws
        return None, None, None, None, d_logits, d_w, None, None, None, None


_uv_scale = {}       # (W, H, device) -> [W - 1, H - 1] on the device (filled on the first eager step, before any capture)


def patch_cameras(ref_intrinsic, src_intrinsics, ref_c2w, src_c2ws):
    """the per-view constants of PatchProjector.patch_warp (patch_projector.py:78-97)."""
    K_ref_inv = torch.inverse(ref_intrinsic[:3, :3])
    K_src = src_intrinsics[:, :3, :3]
    inv_ref_pose = torch.inverse(ref_c2w)
    rel = torch.inverse(src_c2ws) @ ref_c2w
    R_rel, t_rel = rel[:, :3, :3], rel[:, :3, 3:]
    c2 = (-R_rel.transpose(1, 2) @ t_rel)[..., 0]
    ref_cam = torch.cat([K_ref_inv.reshape(-1), inv_ref_pose[:3, :3].reshape(-1), inv_ref_pose[:3, 3].reshape(-1),
                         ref_c2w[:3, 3].reshape(-1)])
    V = K_src.shape[0]
    src_cam = torch.cat([K_src.reshape(V, 9), R_rel.reshape(V, 9), t_rel.reshape(V, 3), c2.reshape(V, 3)], dim=1)
    return ref_cam.float().contiguous(), src_cam.float().contiguous()


def blend_and_composite(hps, pts, logits, weights, grad, rays_d, color_maps, w2cs, intrinsics, query_c2w, rays_uv,
                        bg_in=None, bg_tail=None, patch_cams=None):
    """-> (color_pixel [N,3], patch_colors [N,Npx,3] | None, patch_mask [N] | None).
    `patch_cams` = patch_cameras(...) computed by the caller (a graph-captured step: the four small matrix inverses of the
    camera constants run through a solver library that cannot be captured, so train.GraphedStep computes them outside)."""
    N, S = pts.shape[0], pts.shape[1]
    V, _, H, W = color_maps.shape
    proj = torch.matmul(intrinsics[:, :3, :3], w2cs[:, :3, :]).reshape(V, 12)      # projector_utils.py:69-70
    pix = _PixelBlendFn.apply(pts.reshape(-1, 3), logits.reshape(N * S, -1), proj, color_maps)
    color_pixel = _PixelCompositeFn.apply(weights, pix.reshape(N, S, 3), pts, bg_in, bg_tail)
    patch_colors = patch_mask = None
    if rays_uv is not None:
        # the reference rescales the caller's uv tensor in place from (-1,1) to pixels (patch_projector.py:75-76)
        # (the same three roundings per element -- + 1, / 2, * (size - 1) -- on both columns at once: 4 launches, not 8)
        key = (W, H, str(rays_uv.device))
        if key not in _uv_scale:
            # device-side fills (no pageable host-to-device copy: the first use may sit inside a HIP-graph capture)
            sc = torch.empty(2, dtype=torch.float32, device=rays_uv.device)
            sc[0:1].fill_(float(W - 1))
            sc[1:2].fill_(float(H - 1))
            _uv_scale[key] = sc
        rays_uv.copy_((rays_uv + 1) / 2. * _uv_scale[key])
        ref_cam, src_cam = patch_cams if patch_cams is not None else patch_cameras(intrinsics[0], intrinsics, query_c2w,
                                                                                   torch.inverse(w2cs))
        patch_colors, patch_mask = _PatchBlendFn.apply(pts, grad, rays_d, rays_uv, logits, weights, ref_cam, src_cam,
                                                       color_maps, hps)
    return color_pixel, patch_colors, patch_mask
