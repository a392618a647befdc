"""Synthetic DTU- / DeepFashion3D-shaped inputs.

No dataset is on disk (and none is needed for the hot path), so benches and tests use
cameras placed like the datasets' (outside the unit sphere, looking at the origin) and
the ray/near/far construction of the reference loader:
  rays      dataset/dataset.py:228-294  (random integer pixels, K^-1 p, normalise, rotate)
  near/far  dataset/dataset.py:329-335  (mid -/+ 1 on the unit sphere)
Everything here is CPU torch + an explicit generator so the same seed gives the same
inputs in the build container and on the GPU box.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass
class Scene:
    H: int
    W: int
    intrinsics: torch.Tensor      # [V,4,4]
    c2w: torch.Tensor             # [V,4,4]


def _look_at(eye: torch.Tensor) -> torch.Tensor:
    """camera-to-world with +z looking at the origin, OpenCV convention (x right, y down)."""
    z = -eye / eye.norm()
    up = torch.tensor([0.0, 0.0, 1.0])
    x = torch.linalg.cross(z, up)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    m = torch.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, eye
    return m


def make_scene(kind: str = "dtu") -> Scene:
    if kind == "dtu":
        H, W, n_cam, radius = 1200, 1600, 49, 2.6
        K = torch.tensor([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1.0]])
    elif kind == "garment":
        H, W, n_cam, radius = 1024, 1024, 72, 2.2
        K = torch.tensor([[886.8, 0, 511.5], [0, 886.8, 511.5], [0, 0, 1.0]])
    elif kind == "tiny":                       # small images for fast CPU tests
        H, W, n_cam, radius = 96, 128, 12, 2.6
        K = torch.tensor([[230.0, 0, 63.5], [0, 230.0, 47.5], [0, 0, 1.0]])
    else:
        raise ValueError(kind)
    intr = torch.eye(4).repeat(n_cam, 1, 1)
    intr[:, :3, :3] = K
    poses = []
    for i in range(n_cam):
        a = 2 * math.pi * i / n_cam
        elev = 0.3 * (1 if i % 2 == 0 else -1)
        eye = radius * torch.tensor([math.cos(a) * math.cos(elev), math.sin(a) * math.cos(elev), math.sin(elev)])
        poses.append(_look_at(eye))
    return Scene(H, W, intr, torch.stack(poses))


def near_far_from_sphere(rays_o: torch.Tensor, rays_d: torch.Tensor):
    a = (rays_d ** 2).sum(-1, keepdim=True)
    b = 2.0 * (rays_o * rays_d).sum(-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return mid - 1.0, mid + 1.0


def make_rays(scene: Scene, img_idx: int, n_rays: int, seed: int = 1234, margin: int = 0):
    """-> dict(rays_o, rays_d [N,3], near, far [N,1], true_rgb [N,3], mask [N,1], rays_uv [N,2] ndc,
    pixels [N,2])."""
    g = torch.Generator().manual_seed(seed)
    px = torch.randint(margin, scene.W - margin, (n_rays,), generator=g)
    py = torch.randint(margin, scene.H - margin, (n_rays,), generator=g)
    p = torch.stack([px, py, torch.ones_like(py)], -1).float()
    Kinv = torch.inverse(scene.intrinsics[img_idx, :3, :3])
    p = (Kinv[None] @ p[:, :, None]).squeeze(-1)
    v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
    v = (scene.c2w[img_idx, None, :3, :3] @ v[:, :, None]).squeeze(-1)
    o = scene.c2w[img_idx, None, :3, 3].expand(v.shape).contiguous()
    near, far = near_far_from_sphere(o, v)
    rgb = torch.rand(n_rays, 3, generator=g)
    uv = torch.stack([2 * px / (scene.W - 1) - 1, 2 * py / (scene.H - 1) - 1], -1).float()
    return dict(rays_o=o, rays_d=v.contiguous(), near=near, far=far, true_rgb=rgb,
                mask=torch.ones(n_rays, 1), rays_uv=uv, pixels=torch.stack([px, py], -1))


def make_source_views(scene: Scene, img_idx: int, n_src: int = 8, seed: int = 77, hwc: bool = False):
    """8 nearest cameras + random source images (dataset/dataset.py:129-149 picks by distance).  hwc=True returns
    `color_maps` the way the reference's dataset does: an NCHW view of channel-interleaved memory (:147-149)."""
    g = torch.Generator().manual_seed(seed)
    c = scene.c2w[:, :3, 3]
    d = (c - c[img_idx]).norm(dim=-1)
    order = torch.argsort(d)[1:n_src + 1]
    imgs = torch.rand(n_src, 3, scene.H, scene.W, generator=g)
    if hwc:
        imgs = imgs.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return dict(color_maps=imgs, intrinsics=scene.intrinsics[order].contiguous(),
                src_c2ws=scene.c2w[order].contiguous(), w2cs=torch.inverse(scene.c2w[order]).contiguous(),
                query_c2w=scene.c2w[img_idx].contiguous())
