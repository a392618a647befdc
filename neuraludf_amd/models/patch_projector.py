"""Drop-in for models/patch_projector.py (`PatchProjector(h_patch_size)`); the warps themselves run in
the fused blending kernels (models/blend.py, csrc/blend.hip)."""
import torch


def build_patch_offset(h_patch_size):
    """(dx, dy) offsets of the (2h+1)^2 patch, row by row (patch_projector.py:211-214)."""
    o = torch.arange(-h_patch_size, h_patch_size + 1)
    yy, xx = torch.meshgrid(o, o, indexing="ij")
    return torch.stack([xx, yy], dim=-1).view(1, -1, 2)


class PatchProjector:
    def __init__(self, patch_size):
        self.h_patch_size = patch_size
        self.offsets = build_patch_offset(patch_size)
        self.plane_dist_thresh = 0.001
