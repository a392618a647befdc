"""IDR-format camera files (``cameras*.npz`` with ``world_mat_i`` / ``scale_mat_i``) -> intrinsics / poses, i.e. the
numeric part of ``Dataset.__init__`` (dataset/dataset.py:59-127) without OpenCV.

``load_K_Rt_from_P`` (dataset.py:14-35) calls ``cv2.decomposeProjectionMatrix``; OpenCV is not in this image, so the
decomposition is restated here from its documented behaviour: RQ-factor ``M = P[:, :3] = K R`` with an upper
triangular K whose first two diagonal entries are positive (OpenCV resolves the sign ambiguity by 180-degree
rotations, so det R = +1 and K[2,2] carries the sign of det M) and return the homogeneous camera centre (right null
vector of P).  Parity with OpenCV itself is unpinned (no cv2 here); the tests pin the defining properties instead:
K upper triangular, R a rotation, P ~ K [R | -R C], exact recovery of a synthetic (K, R, C)."""
from __future__ import annotations

import numpy as np
import scipy.linalg


def decompose_projection_matrix(P):
    """-> (K [3,3], R [3,3] world-to-camera rotation, C_h [4] homogeneous camera centre)."""
    P = np.asarray(P, dtype=np.float64)
    M = P[:3, :3]
    K, Q = scipy.linalg.rq(M)
    s = np.sign(np.diag(K)).copy()
    s[s == 0] = 1.0
    s[2] = 1.0
    D = np.diag(s)
    if np.linalg.det(D @ Q) < 0:
        D[2, 2] = -1.0
    K, R = K @ D, D @ Q                      # D D = I, so K R is unchanged
    _, _, vt = np.linalg.svd(P[:3, :4])
    return K, R, vt[-1]


def load_K_Rt_from_P(filename, P=None):
    """dataset.py:14-35 (same name, same return): 4x4 intrinsics (K / K[2,2]) and the 4x4 camera-to-world pose."""
    if P is None:
        lines = open(filename).read().splitlines()
        if len(lines) == 4:
            lines = lines[1:]
        P = np.asarray([ln.split(" ")[:4] for ln in lines]).astype(np.float32).squeeze()
    K, R, c = decompose_projection_matrix(P)
    K = K / K[2, 2]
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.transpose()
    pose[:3, 3] = (c[:3] / c[3])
    return intrinsics, pose


def load_idr_cameras(camera_dict, n_images, downsample_factor=1.0):
    """dataset.py:69-90: per view P = world_mat @ scale_mat -> (intrinsics_all [n,4,4], pose_all [n,4,4],
    scale_mats list), float32.  ``camera_dict`` is the npz (or any mapping)."""
    intr, poses, scale_mats = [], [], []
    for idx in range(n_images):
        world_mat = np.asarray(camera_dict["world_mat_%d" % idx]).astype(np.float32)
        scale_mat = np.asarray(camera_dict["scale_mat_%d" % idx]).astype(np.float32)
        K4, pose = load_K_Rt_from_P(None, (world_mat @ scale_mat)[:3, :4])
        K4[:2] *= downsample_factor
        intr.append(K4.astype(np.float32))
        poses.append(pose.astype(np.float32))
        scale_mats.append(scale_mat)
    return np.stack(intr), np.stack(poses), scale_mats


def object_bbox(scale_mat_0, object_scale_mat):
    """dataset.py:113-123: region of interest for mesh extraction in the normalised frame."""
    lo = np.array([-1.01, -1.01, -1.01, 1.0])
    hi = np.array([1.01, 1.01, 1.01, 1.0])
    T = np.linalg.inv(scale_mat_0) @ object_scale_mat
    return (T @ lo[:, None])[:3, 0], (T @ hi[:, None])[:3, 0]
