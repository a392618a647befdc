"""CPU: the batch-generation oracle (oracle/rays_oracle.py) against the reference's own output
(tests/golden/ref_raybatch.npz, made by tests/golden/make_golden_rays.py from Dataset.gen_random_rays_patches_at)."""
import os

import numpy as np
import pytest

from oracle import rays_oracle as ro

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_raybatch.npz")
CASES = ["plain", "importance", "nopatch"]


def load_gold():
    g = dict(np.load(GOLD))
    g["images"] = g["images_u8"].astype(np.float32) / np.float32(256.0)
    g["masks"] = np.repeat(g["masks_u8"].astype(np.float32) / np.float32(256.0), 3, axis=-1)
    g["intrinsics_all_inv"] = np.linalg.inv(g["intrinsics_all"].astype(np.float64)).astype(np.float32)
    return g


def check_sample(g, name, s, near, far, atol=2e-6):
    """integer / gather outputs bit-exact; float arithmetic within a few ulp of the reference's fp32."""
    ref = lambda k: g[f"{name}.{k}"]  # noqa: E731
    np.testing.assert_array_equal(s["rays"][:, 6:10], ref("rays")[:, 6:10])          # colour gather, mask
    np.testing.assert_array_equal(s["rays"][:, 0:3], ref("rays")[:, 0:3])            # origin
    np.testing.assert_allclose(s["rays"][:, 3:6], ref("rays")[:, 3:6], rtol=0, atol=atol)
    np.testing.assert_allclose(s["rays_ndc_uv"], ref("rays_ndc_uv"), rtol=0, atol=2e-7)
    np.testing.assert_allclose(s["rays_norm_XYZ_cam"], ref("rays_norm_XYZ_cam"), rtol=0, atol=atol)
    np.testing.assert_allclose(near, ref("near"), rtol=0, atol=1e-5)
    np.testing.assert_allclose(far, ref("far"), rtol=0, atol=1e-5)
    if f"{name}.rays_patch_color" in g:
        np.testing.assert_array_equal(np.asarray(s["rays_patch_mask"]).astype(bool), ref("rays_patch_mask"))
        np.testing.assert_allclose(s["rays_patch_color"], ref("rays_patch_color"), rtol=0, atol=2e-5)
    else:
        assert s["rays_patch_color"] is None and s["rays_patch_mask"] is None


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    g = load_gold()
    i = int(g[f"{name}.img_idx"])
    crop = f"{name}.rays_patch_color" in g
    s = ro.gen_rays_patches(g["images"][i], g["masks"][i], g["intrinsics_all_inv"][i], g["pose_all"][i],
                            g[f"{name}.px"], g[f"{name}.py"], int(g[f"{name}.h"]), crop)
    near, far = ro.near_far_from_sphere(s["rays"][:, :3], s["rays"][:, 3:6])
    check_sample(g, name, s, near, far)


def test_patch_offsets_and_border():
    off = ro.build_patch_offset(1)[0]
    assert off.tolist() == [[-1, -1], [0, -1], [1, -1], [-1, 0], [0, 0], [1, 0], [-1, 1], [0, 1], [1, 1]]
    # a patch hanging over the image corner: out-of-image taps contribute zero (padding_mode='zeros')
    img = np.ones((8, 10, 3), np.float32)
    s = ro.gen_rays_patches(img, img, np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32),
                            np.array([0, 9]), np.array([0, 7]), 2, True)
    assert not s["rays_patch_mask"].any()
    assert s["rays_patch_color"][0, 0].max() == 0.0 and s["rays_patch_color"][1, -1].max() == 0.0
    # centre tap of the corner pixel: align_corners=False puts it at (-0.5, -0.5) -> one quarter of the 2x2 footprint
    assert abs(s["rays_patch_color"][0, 12, 0] - 0.25) < 1e-6


def test_whole_image_rays_host_logic():
    """gen_rays_at / gen_rays_between are plain tensor code (validation views, not per iteration): checked on the host
    against the reference Dataset's output; the per-iteration batch itself has no host path."""
    import torch
    from neuraludf_amd.dataset import RayBatchSource
    g = load_gold()
    src = RayBatchSource(g["images"], g["masks"], g["intrinsics_all"], g["pose_all"], device="cpu")
    ro, rv = src.gen_rays_at(2, resolution_level=4)
    np.testing.assert_allclose(ro.numpy(), g["rays_at.o"], atol=1e-6)
    np.testing.assert_allclose(rv.numpy(), g["rays_at.v"], atol=5e-6)
    ro, rv = src.gen_rays_between(1, 3, 0.3, resolution_level=8)
    np.testing.assert_allclose(ro.numpy(), g["rays_between.o"], atol=1e-5)
    np.testing.assert_allclose(rv.numpy(), g["rays_between.v"], atol=1e-5)
    with pytest.raises(RuntimeError):
        src.gen_random_rays_patches_at(0, 8)
    n, f = src.near_far_from_sphere(torch.from_numpy(g["plain.rays"][:, :3]), torch.from_numpy(g["plain.rays"][:, 3:6]))
    np.testing.assert_allclose(n.numpy(), g["plain.near"], atol=1e-5)


def test_from_idr_builds_the_dataset_tensors():
    """RayBatchSource.from_idr = Dataset.__init__'s numeric part: P = world_mat @ scale_mat decomposed (no OpenCV),
    bounding box, neighbour pairs; on the host (no kernels involved)."""
    from neuraludf_amd.dataset import RayBatchSource
    g = load_gold()
    n = g["images"].shape[0]
    cam = {}
    for i in range(n):
        K, c2w = g["intrinsics_all"][i].astype(np.float64), g["pose_all"][i].astype(np.float64)
        w2c = np.linalg.inv(c2w)
        S = np.diag([2.0, 2.0, 2.0, 1.0]); S[:3, 3] = [0.1, -0.2, 0.3]
        P = np.eye(4); P[:3, :4] = (K @ w2c)[:3, :4]
        cam[f"world_mat_{i}"], cam[f"scale_mat_{i}"] = P @ np.linalg.inv(S), S        # so that world_mat @ scale_mat = K [R|t]
    src = RayBatchSource.from_idr(cam, g["images"], g["masks"], device="cpu")
    np.testing.assert_allclose(src.intrinsics_all.numpy()[:, :3, :3], g["intrinsics_all"][:, :3, :3], rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(src.pose_all.numpy(), g["pose_all"], atol=2e-4)
    np.testing.assert_allclose(src.object_bbox_min, [-1.01] * 3, atol=1e-9)
    assert len(src.ref_src_pair) == n and len(src.scale_mats_np) == n


def test_from_directory_reads_the_reference_layout(tmp_path):
    """image/*.png + mask/*.png + cameras.npz -> resident tensors: BGR / 256 like cv.imread, sorted file order."""
    from PIL import Image
    from neuraludf_amd.dataset import RayBatchSource
    from neuraludf_amd.dataset import images as im
    g = load_gold()
    n, H, W = 3, 24, 32
    rng = np.random.default_rng(0)
    os.makedirs(tmp_path / "image"); os.makedirs(tmp_path / "mask")
    rgb = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    msk = (rng.random((n, H, W)) > 0.5).astype(np.uint8) * 255
    for i in (2, 0, 1):                                   # written out of order: the loader sorts
        Image.fromarray(rgb[i]).save(tmp_path / "image" / f"{i:03d}.png")
        Image.fromarray(msk[i]).save(tmp_path / "mask" / f"{i:03d}.png")        # single-channel mask file
    cam = {}
    for i in range(n):
        K, c2w = g["intrinsics_all"][i].astype(np.float64), g["pose_all"][i].astype(np.float64)
        P = np.eye(4); P[:3, :4] = (K @ np.linalg.inv(c2w))[:3, :4]
        cam[f"world_mat_{i}"], cam[f"scale_mat_{i}"] = P, np.eye(4)
    np.savez(tmp_path / "cameras.npz", **cam)
    src = RayBatchSource.from_directory(str(tmp_path), device="cpu")
    assert src.images.shape == (n, H, W, 3) and src.n_images == n
    np.testing.assert_array_equal(src.images.numpy(), rgb[..., ::-1].astype(np.float32) / np.float32(256.0))
    np.testing.assert_array_equal(src.masks.numpy(), np.repeat(msk[..., None], 3, -1).astype(np.float32) / np.float32(256.0))
    np.testing.assert_allclose(src.pose_all.numpy(), g["pose_all"][:n], atol=2e-4)
    assert [os.path.basename(p) for p in src.images_lis] == ["000.png", "001.png", "002.png"]
    assert im.read_bgr(str(tmp_path / "mask" / "000.png")).shape == (H, W, 3)
