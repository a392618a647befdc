"""Image / mask decoding for the dataset directory layout of the reference (dataset/dataset.py:59-97), with Pillow
instead of OpenCV (not in this image).  `cv.imread` returns 8-bit BGR; the reference divides by 256 and keeps that
channel order for the whole pipeline (networks trained by it predict BGR), so the stack returned here is BGR / 256 too.
PNG (DTU, DeepFashion3D) decodes bit-identically; JPEG (BlendedMVS) may differ from OpenCV's decoder by an 8-bit
level or two, as any two JPEG decoders do."""
from __future__ import annotations

import os
from glob import glob

import numpy as np


def read_bgr(path: str) -> np.ndarray:
    """one file -> uint8 [H, W, 3] in BGR order (grey / palette / alpha inputs are converted like cv.imread's default
    IMREAD_COLOR does: three channels, alpha dropped)."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return rgb[:, :, ::-1].copy()


def load_image_stack(paths) -> np.ndarray:
    """sorted file list -> float32 [n, H, W, 3] = BGR / 256  (dataset.py:84-85, 96-97)."""
    if not paths:
        raise FileNotFoundError("no image files")
    return (np.stack([read_bgr(p) for p in paths]) / 256.0).astype(np.float32)


def list_dataset_files(data_dir: str, dataset_name: str = "dtu"):
    """dataset.py:76-82: (image files, mask files), each sorted."""
    if dataset_name in ("dtu", "deepfashion3d"):
        images = sorted(glob(os.path.join(data_dir, "image/*.png")))
        masks = sorted(glob(os.path.join(data_dir, "mask/*.png")))
    elif dataset_name == "bmvs":
        images = sorted(glob(os.path.join(data_dir, "blended_images/*.jpg")))
        masks = sorted(glob(os.path.join(data_dir, "masks/*.jpg")))
    else:
        raise ValueError("unknown dataset_name %r" % (dataset_name,))
    return images, masks
