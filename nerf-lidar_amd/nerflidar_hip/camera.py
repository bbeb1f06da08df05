"""Pinhole camera -> ray batch (BASELINE config 3: 1024x768 novel-view render).

Restates, for the perspective / no-distortion / no-NDC case the nuScenes path uses, the host-side numpy of
ZI/camera_utils.py:454-564 (`pixels_to_rays`) and :567-617 (`cast_ray_batch`): rays through pixel centres, the
OpenCV->OpenGL flip, radii from the distance to the neighbouring pixel rays (x 2/sqrt(12)), and the two image-plane
basis vectors `base_x`, `base_y` that cast_rays spreads the multisamples along (ZI/render.py:157-163).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import synth
from .lidar import seeded_rotation


def pixels_to_rays(pix_x_int, pix_y_int, pixtocam, camtoworld):
    """Returns origins, directions, viewdirs, radii, imageplane, base_x, base_y (camera_utils.py:454-564)."""
    def pix_to_dir(x, y):
        return np.stack([x + .5, y + .5, np.ones_like(x)], axis=-1)
    stacked = np.stack([pix_to_dir(pix_x_int, pix_y_int), pix_to_dir(pix_x_int + 1, pix_y_int),
                        pix_to_dir(pix_x_int, pix_y_int + 1)], axis=0)
    mat_vec = lambda A, b: np.matmul(A, b[..., None])[..., 0]
    cam_dirs = mat_vec(pixtocam, stacked)
    cam_dirs = np.matmul(cam_dirs, np.diag(np.array([1., -1., -1.])))  # OpenCV -> OpenGL
    imageplane = cam_dirs[0, ..., :2]
    directions, dx, dy = mat_vec(camtoworld[..., :3, :3], cam_dirs)
    origins = np.broadcast_to(camtoworld[..., :3, -1], directions.shape)
    viewdirs = directions / np.linalg.norm(directions, axis=-1, keepdims=True)
    pix_x, pix_y = dx - directions, dy - directions
    dx_norm, dy_norm = np.linalg.norm(pix_x, axis=-1), np.linalg.norm(pix_y, axis=-1)
    base_x = pix_x / np.linalg.norm(pix_x, axis=-1, keepdims=True)
    base_y = pix_y / np.linalg.norm(pix_y, axis=-1, keepdims=True)
    radii = (0.5 * (dx_norm + dy_norm))[..., None] * 2 / np.sqrt(12)
    return origins, directions, viewdirs, radii, imageplane, base_x, base_y


def synthetic_camera_batch(width: int = 1024, height: int = 768, focal: float = 800.0, seed: int = 0,
                           near: float = 0.008, far: float = 2.0, rows=None) -> Dict[str, np.ndarray]:
    """Full-image ray batch [H*W, k] (float32) of one seeded pinhole camera."""
    K = np.array([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1.0]])
    pixtocam = np.linalg.inv(K)
    c2w = np.concatenate([seeded_rotation(seed + 17),
                          synth.uniform(seed, 9300, (3, 1), -0.02, 0.02).astype(np.float64)], axis=1)
    ys = np.arange(height) if rows is None else np.asarray(rows)
    px, py = np.meshgrid(np.arange(width), ys, indexing="xy")
    o, d, v, r, ip, bx, by = pixels_to_rays(px.reshape(-1).astype(np.float64), py.reshape(-1).astype(np.float64), pixtocam, c2w)
    n = o.shape[0]
    bs = lambda x: np.broadcast_to(np.asarray(x, np.float64), (n, 1))
    batch = dict(origins=o, directions=d, viewdirs=v, radii=r, imageplane=ip, base_x=bx, base_y=by,
                 near=bs(near), far=bs(far), lossmult=bs(1.0), cam_idx=bs(0))
    return {k: np.ascontiguousarray(val, dtype=np.float32) for k, val in batch.items()}
