"""Stand-in for the handful of OpenCV calls the reference's Dataset / runner make (dataset/dataset.py:14-35, 70-71,
exp_runner_blending.py validation) for images without OpenCV: Pillow for the image files, neuraludf_amd.dataset for
the projection-matrix decomposition.  Only used by scripts/run_reference_runner.py when `cv2` cannot be imported."""
import numpy as np
from PIL import Image

from neuraludf_amd.dataset.cameras import decompose_projection_matrix as _decompose

INTER_LINEAR = 1
IMREAD_COLOR = 1


def imread(path, flags=IMREAD_COLOR):
    """BGR uint8 [H,W,3], like cv2.imread (None when the file cannot be read)."""
    try:
        im = np.asarray(Image.open(path).convert("RGB"))
    except (OSError, FileNotFoundError):
        return None
    return np.ascontiguousarray(im[:, :, ::-1])


def imwrite(path, img):
    a = np.asarray(img)
    if a.ndim == 3 and a.shape[2] == 3:
        a = a[:, :, ::-1]
    Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(path)
    return True


def resize(img, dsize=None, fx=None, fy=None, interpolation=INTER_LINEAR):
    a = np.asarray(img)
    if dsize is None:
        dsize = (int(round(a.shape[1] * fx)), int(round(a.shape[0] * fy)))
    return np.asarray(Image.fromarray(a.astype(np.uint8)).resize((int(dsize[0]), int(dsize[1])), Image.BILINEAR))


def decomposeProjectionMatrix(P):
    """-> (K, R, t_homogeneous [4,1], ...) with cv2's conventions (dataset.py:22-25 uses the first three)."""
    K, R, c = _decompose(np.asarray(P, dtype=np.float64))
    return K, R, c.reshape(4, 1), None, None, None, None


def circle(img, center, radius, color, thickness=-1):
    return img
