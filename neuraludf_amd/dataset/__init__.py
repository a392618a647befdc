from .ray_batch import RayBatchSource, build_patch_offset  # noqa: F401
