"""Drop-in for models/patch_projector.py: `PatchProjector(h_patch_size)` with the reference's two public methods,
`pixel_warp` (:21-43) and `patch_warp` (:45-164), on the forward-only `nudf_pixel_warp` / `nudf_patch_warp` kernels
(csrc/blend.hip).  They return the PER-VIEW samples the reference's projector interface promises; the training path
(`UDFRendererBlending.render_core`) does not go through them -- it uses the fused blending kernels of models/blend.py,
which never materialise the [V, N*S*Npx, 3] tensors."""
import torch

from .._lib import PatchWarp, PixelBlend, call, ptr
from .blend import _img, patch_cameras


def build_patch_offset(h_patch_size):
    """(dx, dy) offsets of the (2h+1)^2 patch, row by row (patch_projector.py:211-214)."""
    o = torch.arange(-h_patch_size, h_patch_size + 1)
    yy, xx = torch.meshgrid(o, o, indexing="ij")
    return torch.stack([xx, yy], dim=-1).view(1, -1, 2)


def _c(t):
    return t.detach().float().contiguous()


class PatchProjector:
    def __init__(self, patch_size):
        self.h_patch_size = patch_size
        self.offsets = build_patch_offset(patch_size)
        self.z_axis = torch.tensor([0, 0, 1]).float()
        self.plane_dist_thresh = 0.001

    @torch.no_grad()
    def pixel_warp(self, pts, imgs, intrinsics, w2cs, img_wh=None):
        """pts [N_rays, n_samples, 3], imgs [V,3,H,W], intrinsics / w2cs [V,4,4]
        -> (colours [N_rays, n_samples, V, 3], valid mask [N_rays, n_samples, V] bool)   (patch_projector.py:21-43)."""
        N, S = pts.shape[0], pts.shape[1]
        if img_wh is not None and (img_wh[0] != imgs.shape[3] or img_wh[1] != imgs.shape[2]):
            raise NotImplementedError("img_wh different from the images' own size is unused by the reference")
        im, (V, _, H, W), layout = _img(imgs)
        proj = torch.matmul(intrinsics[:, :3, :3], w2cs[:, :3, :]).reshape(V, 12)      # projector_utils.py:69-70
        p3 = _c(pts).reshape(-1, 3)
        a = PixelBlend()
        a.pts, a.proj, a.imgs = ptr(p3), ptr(_c(proj)), ptr(im)
        a.P, a.V, a.H, a.W, a.img_layout, a.nl = N * S, V, H, W, layout, V
        colors = torch.empty(N * S, V, 3, device=p3.device)
        mask = torch.empty(N * S, V, device=p3.device)
        call("nudf_pixel_warp", a, ptr(colors), ptr(mask))
        return colors.view(N, S, V, 3), (mask > 0.5).view(N, S, V)

    @torch.no_grad()
    def patch_warp(self, pts, uv, normals, src_imgs, ref_intrinsic, src_intrinsics, ref_c2w, src_c2ws, img_wh=None,
                   detach_normal=False):
        """-> (sampled rgb [N_rays, n_samples, N_src, Npx, 3], warp mask [N_rays, n_samples, N_src, Npx] bool)
        (patch_projector.py:45-164).  Like the reference, rescales the caller's `uv` IN PLACE from (-1, 1) to pixels."""
        N, S = pts.shape[0], pts.shape[1]
        im, (V, _, H, W), layout = _img(src_imgs)
        sizeW, sizeH = (img_wh[0], img_wh[1]) if img_wh is not None else (W, H)
        if (sizeW, sizeH) != (W, H):
            raise NotImplementedError("img_wh different from the images' own size is unused by the reference")
        uv[:, 0] = (uv[:, 0] + 1) / 2. * (sizeW - 1)
        uv[:, 1] = (uv[:, 1] + 1) / 2. * (sizeH - 1)
        ref_cam, src_cam = patch_cameras(ref_intrinsic, src_intrinsics, ref_c2w, src_c2ws)
        npx = (2 * self.h_patch_size + 1) ** 2
        a = PatchWarp()
        keep = [_c(pts), _c(normals), _c(uv), ref_cam, src_cam]
        a.pts, a.normals, a.uv, a.ref_cam, a.src_cam, a.imgs = [ptr(t) for t in keep] + [ptr(im)]
        a.N, a.S, a.V, a.H, a.W, a.hps, a.img_layout = N, S, V, H, W, self.h_patch_size, layout
        colors = torch.empty(N, S, V, npx, 3, device=keep[0].device)
        mask = torch.empty(N, S, V, npx, device=keep[0].device)
        a.colors, a.mask = ptr(colors), ptr(mask)
        call("nudf_patch_warp", a)
        return colors, mask > 0.5
