"""LiDAR sweep -> ray batch (the batch contract `Model.forward` consumes).

Restates, for synthetic sweeps, what the reference does on the host in numpy:
  - beam table and azimuth grid: ZI/lidar_utils.py:122-124, 133 (32 sorted nuScenes elevations,
    azimuth linspace(270 deg, -90 deg, W));
  - direction formula `[cos(th)sin(ph), cos(th)cos(ph), sin(th)]`, beam-major order
    (idx = beam*W + az): ZI/lidar_utils.py:559-568;
  - ray batch: ZI/lidar_utils.py:8-33 (`cast_lidar_ray_batch`) and ZI/datasets.py:640-705
    (`_make_simu_lidar_ray_batch`), including the quirks that change numbers:
      * viewdirs = directions / ||directions||_F over the WHOLE [N,3] array (lidar_utils.py:12);
      * base_x = base_y = directions (lidar_utils.py:17-18); radii = 5e-4 (lidar_utils.py:14).
Everything is produced in float64 numpy and cast to float32 at the end (datasets.py:705).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import synth

# ZI/lidar_utils.py:122-123 (sorted at :124)
LIDAR_ANGLES = sorted([-30.67, -9.33, -29.33, -8.00, -28.00, -6.67, -26.67, -5.33, -25.33, -4.00, -24.00,
                       -2.67, -22.67, -1.33, -21.33, 0.00, -20.00, 1.33, -18.67, 2.67, -17.33, 4.00, -16.00,
                       5.33, -14.67, 6.67, -13.33, 8.00, -12.00, 9.33, -10.67, 10.67])


def get_directions(vertical_angles, horizontal_angles) -> np.ndarray:
    """[len(v)*len(h), 3] float32, beam-major (ZI/lidar_utils.py:559-568)."""
    theta = (np.asarray(vertical_angles, np.float64) / 180 * np.pi)[:, None]
    phi = np.asarray(horizontal_angles, np.float64)[None, :]
    d = np.stack([np.cos(theta) * np.sin(phi), np.cos(theta) * np.cos(phi),
                  np.broadcast_to(np.sin(theta), (theta.shape[0], phi.shape[1]))], axis=-1)
    return d.reshape(-1, 3).astype(np.float32)


def seeded_rotation(seed: int) -> np.ndarray:
    """Fixed SO(3) standing in for lidar2cam @ c2w (ZI/lidar_utils.py:138-142)."""
    a = synth.uniform(seed, 9001, (3, 3), -1.0, 1.0).astype(np.float64)
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))[None, :]
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def cast_lidar_ray_batch(origins: np.ndarray, directions: np.ndarray, near: float, far: float) -> Dict[str, np.ndarray]:
    """ZI/lidar_utils.py:8-33 + ZI/datasets.py:669-705 -> dict of float32 [N,k] arrays."""
    n = origins.shape[0]
    bs = lambda x: np.broadcast_to(np.asarray(x, np.float64), (n, 1))
    batch = dict(
        origins=origins,
        directions=directions,
        viewdirs=directions / np.linalg.norm(directions),  # Frobenius norm of the whole array (sic)
        radii=np.ones(n).reshape(-1, 1) * 0.0005,
        imageplane=np.zeros_like(origins)[:, :2],
        lossmult=bs(1.0), near=bs(near), far=bs(far), cam_idx=bs(-1),
        base_x=directions, base_y=directions,
        rgb=np.zeros_like(origins),
        semantic=np.ones(n) * 255,
        mask=np.ones(n),
    )
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in batch.items()}


def synthetic_sweep(width: int = 1024, seed: int = 0, scale_factor: float = 1.0 / 250.0,
                    sweep_idx: int = 0, beams=None) -> Dict[str, np.ndarray]:
    """One nuScenes-shaped sweep (SURVEY section 8d "Synthetic inputs").

    near = 2*scale_factor, far = 500*scale_factor (ZI/datasets.py:1233-1234); origin = o0 +
    U(-0.01,0.01)^3 per sweep; directions rotated by a seeded SO(3).
    """
    beams = LIDAR_ANGLES if beams is None else beams
    az = np.linspace(270, -90, width) / 180 * np.pi  # lidar_utils.py:133
    d = get_directions(beams, az).astype(np.float64) @ seeded_rotation(seed).T
    o0 = synth.uniform(seed, 9100 + sweep_idx, (3,), -0.01, 0.01).astype(np.float64)
    o = np.broadcast_to(o0[None, :], d.shape)
    return cast_lidar_ray_batch(np.array(o), d, 2.0 * scale_factor, 500.0 * scale_factor)


def azimuth_sector(batch: Dict[str, np.ndarray], height: int, width: int, rank: int, world: int):
    """Columns [rank*W/P, (rank+1)*W/P) of every beam.  Rays are beam-major, so the
    sector is a strided gather; W is padded up to a multiple of `world` by repeating the last
    column (padded columns are stripped after the all-gather)."""
    wp = -(-width // world)
    cols = np.minimum(np.arange(rank * wp, (rank + 1) * wp), width - 1)
    idx = (np.arange(height)[:, None] * width + cols[None, :]).reshape(-1)
    return {k: (v[idx] if v.shape[0] == height * width else v) for k, v in batch.items()}, wp
