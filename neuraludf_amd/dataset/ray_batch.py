"""GPU-resident training-batch generation: the per-iteration half of the reference's ``Dataset``
(dataset/dataset.py), i.e. ``gen_random_rays_patches_at`` (:228-294), ``near_far_from_sphere`` (:329-335) and
``get_ref_src_info`` (:141-149).  Image decoding / camera files (dataset.py:40-140, cv2 / glob) stay the caller's job:
this class is constructed from the tensors that ``Dataset.__init__`` ends up with.

MI355X-first difference: the reference keeps ``images`` / ``masks`` on the host and, every iteration, uploads the
whole reference image for ``F.grid_sample`` (23 MB for a DTU view) plus the gathered colours; here the full image
stack lives in HBM (49 DTU views = 1.1 GB of 288 GB) and one launch of ``nudf_gen_ray_batch`` writes the ray record,
the ndc uv, the camera-space direction, near / far and the ground-truth patches.
"""
from __future__ import annotations

import os

import torch

from .. import _lib


def build_patch_offset(h_patch_size: int, device=None) -> torch.Tensor:
    """dataset.py:342-344: [1, (2h+1)^2, 2] integer (dx, dy) offsets, dx fastest."""
    o = torch.arange(-h_patch_size, h_patch_size + 1, device=device)
    dy, dx = torch.meshgrid(o, o, indexing="ij")
    return torch.stack([dx, dy], dim=-1).view(1, -1, 2)


class RayBatchSource:
    """Holds ``images`` [n, H, W, 3], ``masks`` [n, H, W, 3], ``intrinsics_all`` [n, 4, 4], ``pose_all`` [n, 4, 4]
    (same meaning as the reference attributes of those names, dataset.py:96-112) on the GPU."""

    def __init__(self, images, masks, intrinsics_all, pose_all, device="cuda"):
        dev = torch.device(device)
        f = dict(device=dev, dtype=torch.float32)
        self.device = dev
        self.images = torch.as_tensor(images).to(**f).contiguous()
        self.masks = None if masks is None else torch.as_tensor(masks).to(**f).contiguous()
        self.intrinsics_all = torch.as_tensor(intrinsics_all).to(**f).contiguous()
        self.intrinsics_all_inv = torch.inverse(self.intrinsics_all).contiguous()      # dataset.py:107
        self.pose_all = torch.as_tensor(pose_all).to(**f).contiguous()
        self.n_images, self.H, self.W = self.images.shape[0], self.images.shape[1], self.images.shape[2]
        if self.masks is None:
            self.masks = torch.ones_like(self.images)                                    # dataset.py:86-88
        self._valid = {}

    @classmethod
    def from_idr(cls, camera_dict, images, masks=None, object_camera_dict=None, downsample_factor=1.0, device="cuda"):
        """the numeric part of Dataset.__init__ (dataset.py:59-127): `camera_dict` = the loaded `cameras*.npz`
        (`world_mat_i`, `scale_mat_i`), `images` / `masks` = the decoded [n, H, W, 3] arrays already divided by 256
        (decoding with cv2 / PIL stays the caller's job).  Also sets scale_mats_np and object_bbox_min / _max."""
        import numpy as np
        from . import cameras
        n = len(images)
        intr, poses, scale_mats = cameras.load_idr_cameras(camera_dict, n, downsample_factor)
        self = cls(images, masks, intr, poses, device=device)
        if downsample_factor != 1:              # dataset.py:99-108
            import torch.nn.functional as F
            def rs(t):
                return F.interpolate(t.permute(0, 3, 1, 2).contiguous(), size=None, scale_factor=downsample_factor,
                                     mode="bilinear").permute(0, 2, 3, 1).contiguous()
            self.images, self.masks = rs(self.images), rs(self.masks)
            self.H, self.W = self.images.shape[1], self.images.shape[2]
        self.scale_mats_np = scale_mats
        obj = np.asarray((object_camera_dict or camera_dict)["scale_mat_0"])
        self.object_bbox_min, self.object_bbox_max = cameras.object_bbox(scale_mats[0], obj)
        self.focal = self.intrinsics_all[0][0, 0]
        self.prepare_ref_src_pairs()
        return self

    @classmethod
    def from_directory(cls, data_dir, dataset_name="dtu", render_cameras_name="cameras.npz", object_cameras_name=None,
                       downsample_factor=1.0, device="cuda"):
        """Dataset.__init__ on a data directory (dataset.py:40-127): `image/*.png` + `mask/*.png` (or the BlendedMVS
        names) and the IDR camera files.  Missing masks mean all-ones, as at dataset.py:86-88."""
        import numpy as np
        from . import images as im
        img_files, mask_files = im.list_dataset_files(data_dir, dataset_name)
        images = im.load_image_stack(img_files)
        masks = im.load_image_stack(mask_files) if mask_files else None
        cams = np.load(os.path.join(data_dir, render_cameras_name))
        obj = np.load(os.path.join(data_dir, object_cameras_name)) if object_cameras_name else None
        self = cls.from_idr(cams, images, masks, obj, downsample_factor, device)
        self.images_lis, self.masks_lis, self.data_dir = img_files, mask_files, data_dir
        return self

    # -- pixel draws: torch's generator, same call order as dataset.py:235-252 -------------------------------------
    def _valid_pixels(self, img_idx: int) -> torch.Tensor:
        v = self._valid.get(img_idx)
        if v is None:      # row-major list of pixels with mask > 0; the reference rebuilds it every call (:242-245)
            v = torch.nonzero(self.masks[img_idx][:, :, 0] > 0)          # [num, 2] = (y, x)
            self._valid[img_idx] = v
        return v

    def draw_pixels(self, img_idx: int, batch_size: int, importance_sample: bool = False, generator=None):
        kw = dict(device=self.device, generator=generator)
        if not importance_sample:
            return (torch.randint(0, self.W, [batch_size], **kw), torch.randint(0, self.H, [batch_size], **kw))
        x1 = torch.randint(0, self.W, [batch_size // 4], **kw)
        y1 = torch.randint(0, self.H, [batch_size // 4], **kw)
        valid = self._valid_pixels(img_idx)
        sel = valid[torch.randint(0, valid.shape[0], [batch_size // 4 * 3], **kw)]
        return torch.cat([x1, sel[:, 1]]), torch.cat([y1, sel[:, 0]])

    # -- the batch -------------------------------------------------------------------------------------------------
    def rays_at_pixels(self, img_idx: int, pixels_x, pixels_y, h_patch_size: int = 3, crop_patch: bool = False,
                       with_near_far: bool = False):
        if not (self.images.is_cuda):
            raise RuntimeError("RayBatchSource needs a GPU (no CPU fallback)")
        px = pixels_x.to(device=self.device, dtype=torch.int64).contiguous()
        py = pixels_y.to(device=self.device, dtype=torch.int64).contiguous()
        N = px.numel()
        f = dict(device=self.device, dtype=torch.float32)
        rays = torch.empty(N, 10, **f)
        uv = torch.empty(N, 2, **f)
        xyz = torch.empty(N, 3, **f)
        near = far = None
        if with_near_far:
            near, far = torch.empty(N, 1, **f), torch.empty(N, 1, **f)
        patch_color = patch_mask = None
        if crop_patch:
            npx = (2 * h_patch_size + 1) ** 2
            patch_color = torch.empty(N, npx, 3, **f)
            patch_mask = torch.empty(N, 1, device=self.device, dtype=torch.bool)
        a = _lib.RayBatch()
        a.image, a.mask = self.images[img_idx].data_ptr(), self.masks[img_idx].data_ptr()
        a.intrinsics_inv, a.pose = self.intrinsics_all_inv[img_idx].data_ptr(), self.pose_all[img_idx].data_ptr()
        a.pixels_x, a.pixels_y = px.data_ptr(), py.data_ptr()
        a.N, a.H, a.W, a.h_patch_size = N, self.H, self.W, int(h_patch_size)
        a.rays, a.ndc_uv, a.xyz_cam = rays.data_ptr(), uv.data_ptr(), xyz.data_ptr()
        a.near = near.data_ptr() if with_near_far else None
        a.far = far.data_ptr() if with_near_far else None
        a.patch_color = patch_color.data_ptr() if crop_patch else None
        a.patch_mask = patch_mask.data_ptr() if crop_patch else None
        _lib.call("nudf_gen_ray_batch", a)
        sample = {"rays": rays, "rays_ndc_uv": uv, "rays_norm_XYZ_cam": xyz, "rays_patch_color": patch_color,
                  "rays_patch_mask": patch_mask}
        if with_near_far:
            sample["near"], sample["far"] = near, far
        return sample

    def gen_random_rays_patches_at(self, img_idx, batch_size, importance_sample=False, h_patch_size=3,
                                   crop_patch=False, generator=None, with_near_far=False):
        """Same arguments and sample dict as dataset.py:228-294 (+ optional near / far from the same launch)."""
        px, py = self.draw_pixels(int(img_idx), batch_size, importance_sample, generator)
        return self.rays_at_pixels(int(img_idx), px, py, h_patch_size, crop_patch, with_near_far)

    def gen_random_rays_at(self, img_idx, batch_size, importance_sample=False, generator=None):
        """dataset.py:196-226: the [N, 10] ray record only (no patches, no uv)."""
        px, py = self.draw_pixels(int(img_idx), batch_size, importance_sample, generator)
        return self.rays_at_pixels(int(img_idx), px, py)["rays"]

    # -- whole-image rays for validation / novel views (not per-iteration: plain device tensor ops) ----------------
    def _pixel_dirs(self, cam_idx, resolution_level):
        l = resolution_level
        tx = torch.linspace(0, self.W - 1, self.W // l, device=self.device)
        ty = torch.linspace(0, self.H - 1, self.H // l, device=self.device)
        pixels_x, pixels_y = torch.meshgrid(tx, ty, indexing="ij")
        p = torch.stack([pixels_x, pixels_y, torch.ones_like(pixels_y)], dim=-1)             # W, H, 3
        p = torch.matmul(self.intrinsics_all_inv[cam_idx, None, None, :3, :3], p[:, :, :, None]).squeeze(-1)
        return p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)

    def gen_rays_at(self, img_idx, resolution_level=1):
        """dataset.py:151-164 -> (rays_o, rays_v), each [H // l, W // l, 3]."""
        v = self._pixel_dirs(img_idx, resolution_level)
        rays_v = torch.matmul(self.pose_all[img_idx, None, None, :3, :3], v[:, :, :, None]).squeeze(-1)
        rays_o = self.pose_all[img_idx, None, None, :3, 3].expand(rays_v.shape)
        return rays_o.transpose(0, 1), rays_v.transpose(0, 1)

    def gen_rays_between(self, idx_0, idx_1, ratio, resolution_level=1):
        """dataset.py:296-327: camera interpolated between two views (rotation slerp of the world-to-camera
        rotations, linear translation), intrinsics of view 0."""
        import numpy as np
        from scipy.spatial.transform import Rotation as Rot
        from scipy.spatial.transform import Slerp
        v = self._pixel_dirs(0, resolution_level)
        pose_0 = np.linalg.inv(self.pose_all[idx_0].detach().cpu().numpy())
        pose_1 = np.linalg.inv(self.pose_all[idx_1].detach().cpu().numpy())
        rot = Slerp([0, 1], Rot.from_matrix(np.stack([pose_0[:3, :3], pose_1[:3, :3]])))(ratio)
        pose = np.diag([1.0, 1.0, 1.0, 1.0]).astype(np.float32)
        pose[:3, :3] = rot.as_matrix()
        pose[:3, 3] = ((1.0 - ratio) * pose_0 + ratio * pose_1)[:3, 3]
        pose = np.linalg.inv(pose)
        rot_t = torch.from_numpy(pose[:3, :3]).to(self.device)
        trans = torch.from_numpy(pose[:3, 3]).to(self.device)
        rays_v = torch.matmul(rot_t[None, None, :3, :3], v[:, :, :, None]).squeeze(-1)
        rays_o = trans[None, None, :3].expand(rays_v.shape)
        return rays_o.transpose(0, 1), rays_v.transpose(0, 1)

    def near_far_from_sphere(self, rays_o, rays_d):
        """dataset.py:329-335 on already generated rays (the fused path is ``with_near_far=True``)."""
        a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
        b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
        mid = 0.5 * (-b) / a
        return mid - 1.0, mid + 1.0

    def prepare_ref_src_pairs(self):
        """dataset.py:129-139: for every view the 9 nearest other cameras (by camera-centre distance)."""
        cam_loc = self.pose_all[:, :3, 3]
        pair_dist = torch.cdist(cam_loc[None], cam_loc[None], p=2.0)
        _, indices = torch.sort(pair_dist, descending=False, dim=2)
        self.ref_src_pair = {i: indices[0, i][1:10] for i in range(self.n_images)}
        return self.ref_src_pair

    def get_ref_src_info(self, img_idx, num=8):
        """dataset.py:141-149: (ref c2w, src c2ws, src intrinsics [n,4,4], src images [n,3,H,W], [W, H]);
        everything is already resident, so the `.cuda()` uploads of the reference are index selects."""
        if isinstance(img_idx, torch.Tensor):
            img_idx = int(img_idx.item())
        if not hasattr(self, "ref_src_pair"):
            self.prepare_ref_src_pairs()
        src_idx = self.ref_src_pair[img_idx][:num]
        return (self.pose_all[img_idx], self.pose_all[src_idx], self.intrinsics_all[src_idx],
                self.images[src_idx].permute(0, 3, 1, 2), [self.W, self.H])

    def src_w2cs(self, img_idx, num=8):
        """world-to-camera matrices of the `num` source views of image `img_idx`: the runner calls
        ``torch.inverse(src_c2ws)`` every iteration (exp_runner_blending.py:283-285, flagged "very slow" there); the poses
        are constants, so all of them are inverted ONCE (same torch.inverse, same values) and the per-iteration work is
        an index select."""
        if isinstance(img_idx, torch.Tensor):
            img_idx = int(img_idx.item())
        if getattr(self, "_w2c_all", None) is None or self._w2c_all.device != self.pose_all.device:
            self._w2c_all = torch.inverse(self.pose_all)
        if not hasattr(self, "ref_src_pair"):
            self.prepare_ref_src_pairs()
        return self._w2c_all[self.ref_src_pair[img_idx][:num]]
