"""CPU restatement (numpy, float32) of the reference's per-iteration batch generation -- TEST INFRASTRUCTURE ONLY
(imported by tests/ and nothing else; the product path is neuraludf_amd/csrc/raybatch.hip).

Follows dataset/dataset.py:254-294 (gen_random_rays_patches_at, for given pixel coordinates), :329-335
(near_far_from_sphere) and :342-344 (build_patch_offset); ``grid_sample_zeros`` restates
torch.nn.functional.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False) as the reference calls
it at :262-265.  Pinned against the reference itself by tests/golden/ref_raybatch.npz (tests/golden/make_golden_rays.py
runs Dataset.gen_random_rays_patches_at with the pixel draws recorded)."""
import numpy as np

f32 = np.float32


def build_patch_offset(h):
    o = np.arange(-h, h + 1)
    dy, dx = np.meshgrid(o, o, indexing="ij")
    return np.stack([dx, dy], -1).reshape(1, -1, 2)


def grid_sample_zeros(img_hwc, u, v):
    """img [H, W, C] float32, u / v normalised coords (any shape) -> [..., C]."""
    H, W = img_hwc.shape[:2]
    ix = ((u.astype(f32) + f32(1)) * f32(W) - f32(1)) / f32(2)
    iy = ((v.astype(f32) + f32(1)) * f32(H) - f32(1)) / f32(2)
    x0 = np.floor(ix); y0 = np.floor(iy)
    tx = (ix - x0).astype(f32); ty = (iy - y0).astype(f32)
    out = np.zeros(u.shape + (img_hwc.shape[2],), f32)
    for j in (0, 1):
        for i in (0, 1):
            xx = (x0 + i).astype(np.int64); yy = (y0 + j).astype(np.int64)
            ok = (xx >= 0) & (yy >= 0) & (xx < W) & (yy < H)
            w = ((tx if i else f32(1) - tx) * (ty if j else f32(1) - ty)).astype(f32)
            val = img_hwc[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
            out += np.where(ok[..., None], w[..., None] * val, f32(0)).astype(f32)
    return out


def gen_rays_patches(image, mask, intrinsics_inv, pose, px, py, h_patch_size=3, crop_patch=False):
    """image / mask [H, W, 3], intrinsics_inv / pose [4, 4], px / py int64 [N] -> the reference's sample dict."""
    H, W = image.shape[:2]
    px = px.astype(np.int64); py = py.astype(np.int64)
    patch_color = patch_mask = None
    if crop_patch:                                                                   # :254-267
        grid = np.stack([px, py], -1).reshape(-1, 1, 2).astype(f32) + build_patch_offset(h_patch_size).astype(f32)
        patch_mask = ((px > h_patch_size) & (px < W - h_patch_size) & (py > h_patch_size)
                      & (py < H - h_patch_size)).reshape(-1, 1)
        gu = f32(2) * grid[:, :, 0] / f32(W - 1) - f32(1)
        gv = f32(2) * grid[:, :, 1] / f32(H - 1) - f32(1)
        patch_color = grid_sample_zeros(image, gu, gv)
    ndc = np.stack([(2 * px).astype(f32) / f32(W - 1) - f32(1), (2 * py).astype(f32) / f32(H - 1) - f32(1)], -1)
    color = image[py, px]                                                            # :275
    m = (mask[py, px] > 0).astype(f32)                                               # :276
    p = np.stack([px, py, np.ones_like(px)], -1).astype(f32)                         # :277
    p = (intrinsics_inv[None, :3, :3].astype(f32) @ p[:, :, None])[:, :, 0]          # :278
    v = p / np.sqrt((p * p).sum(-1, keepdims=True, dtype=f32))                       # :279
    v = (pose[None, :3, :3].astype(f32) @ v[:, :, None])[:, :, 0]                    # :280
    o = np.broadcast_to(pose[None, :3, 3].astype(f32), v.shape)                      # :281
    rays = np.concatenate([o, v, color, m[:, :1]], -1).astype(f32)                   # :283
    return {"rays": rays, "rays_ndc_uv": ndc.astype(f32), "rays_norm_XYZ_cam": p.astype(f32),
            "rays_patch_color": patch_color, "rays_patch_mask": patch_mask}


def near_far_from_sphere(rays_o, rays_d):                                            # :329-335
    a = (rays_d * rays_d).sum(-1, keepdims=True, dtype=f32)
    b = f32(2) * (rays_o * rays_d).sum(-1, keepdims=True, dtype=f32)
    mid = f32(0.5) * (-b) / a
    return mid - f32(1), mid + f32(1)
