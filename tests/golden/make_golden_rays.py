"""Golden vectors for the batch generator, produced by the REFERENCE's own Dataset methods (build container only):

    python tests/golden/make_golden_rays.py   ->  tests/golden/ref_raybatch.npz

The reference Dataset.__init__ needs cv2 + image files; the methods under test only read a handful of tensor
attributes, so an instance is made with object.__new__ and those attributes are filled from a seeded synthetic scene
(neuraludf_amd.synth).  torch.randint is wrapped to RECORD the pixel draws so that the fixture carries them; .cuda()
is the identity here (no GPU in the build container)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from neuraludf_amd import synth  # noqa: E402


def load_ref_dataset_module():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("_nudf_ref_dataset", "/root/reference/dataset/dataset.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_scene_tensors(seed=3):
    scene = synth.make_scene("tiny")
    g = torch.Generator().manual_seed(seed)
    n = 4
    # like dataset.py:84-97: 8-bit images / masks divided by 256
    images_u8 = torch.randint(0, 256, (n, scene.H, scene.W, 3), generator=g, dtype=torch.uint8)
    masks_u8 = (torch.rand(n, scene.H, scene.W, 1, generator=g) > 0.4).to(torch.uint8) * 255
    images = images_u8.float() / 256.0
    masks = (masks_u8.float() / 256.0).expand(-1, -1, -1, 3).contiguous()
    return scene, images, masks, scene.intrinsics[:n].clone(), scene.c2w[:n].clone(), images_u8, masks_u8


def main():
    mod = load_ref_dataset_module()
    torch.Tensor.cuda = lambda self, *a, **k: self
    scene, images, masks, intr, poses, images_u8, masks_u8 = make_scene_tensors()
    ds = object.__new__(mod.Dataset)
    ds.images, ds.masks, ds.H, ds.W = images, masks, scene.H, scene.W
    ds.intrinsics_all, ds.intrinsics_all_inv, ds.pose_all = intr, torch.inverse(intr), poses
    ds.n_images = images.shape[0]
    out = {"images_u8": images_u8.numpy(), "masks_u8": masks_u8.numpy(), "intrinsics_all": intr.numpy(),
           "pose_all": poses.numpy()}
    real_randint = torch.randint
    for name, kw in {"plain": dict(img_idx=1, batch_size=96, importance_sample=False, h_patch_size=3, crop_patch=True),
                     "importance": dict(img_idx=2, batch_size=64, importance_sample=True, h_patch_size=2,
                                        crop_patch=True),
                     "nopatch": dict(img_idx=0, batch_size=32, importance_sample=False, crop_patch=False)}.items():
        draws = []

        def rec(*a, **k):
            t = real_randint(*a, **k)
            draws.append(t.clone())
            return t
        torch.manual_seed(11)
        torch.randint = rec
        try:
            s = ds.gen_random_rays_patches_at(**kw)
        finally:
            torch.randint = real_randint
        near, far = ds.near_far_from_sphere(s["rays"][:, :3], s["rays"][:, 3:6])
        # recover the pixels from the ndc uv is lossy; recompute from the recorded draws the way the reference does
        if not kw["importance_sample"]:
            px, py = draws[0], draws[1]
        else:
            valid = torch.nonzero(masks[kw["img_idx"]][:, :, 0] > 0)
            sel = valid[draws[2]]
            px, py = torch.cat([draws[0], sel[:, 1]]), torch.cat([draws[1], sel[:, 0]])
        out[f"{name}.px"], out[f"{name}.py"] = px.numpy(), py.numpy()
        out[f"{name}.img_idx"] = np.int64(kw["img_idx"])
        out[f"{name}.h"] = np.int64(kw.get("h_patch_size", 3))
        for i, d in enumerate(draws):
            out[f"{name}.draw{i}"] = d.numpy()
        for k, v in s.items():
            if v is not None:
                out[f"{name}.{k}"] = v.numpy()
        out[f"{name}.near"], out[f"{name}.far"] = near.numpy(), far.numpy()
    ro, rv = ds.gen_rays_at(2, resolution_level=4)
    out["rays_at.o"], out["rays_at.v"] = ro.contiguous().numpy(), rv.contiguous().numpy()
    ro, rv = ds.gen_rays_between(1, 3, 0.3, resolution_level=8)
    out["rays_between.o"], out["rays_between.v"] = ro.contiguous().numpy(), rv.contiguous().numpy()
    draws = []

    def rec2(*a, **k):
        t = real_randint(*a, **k)
        draws.append(t.clone())
        return t
    torch.manual_seed(3)
    torch.randint = rec2
    try:
        out["random_rays_at"] = ds.gen_random_rays_at(3, 40).numpy()
    finally:
        torch.randint = real_randint
    out["random_rays_at.px"], out["random_rays_at.py"] = draws[0].numpy(), draws[1].numpy()
    ds.ref_src_pair = None
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        pairs = mod.Dataset.prepare_ref_src_pairs(ds)
    out["ref_src_pairs"] = np.stack([pairs[i].numpy() for i in range(ds.n_images)])
    np.savez_compressed(os.path.join(HERE, "ref_raybatch.npz"), **out)
    print("wrote ref_raybatch.npz", {k: v.shape for k, v in out.items() if k.startswith("plain")})


if __name__ == "__main__":
    main()
