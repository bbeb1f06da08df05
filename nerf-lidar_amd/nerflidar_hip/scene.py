"""An analytic street scene and its LiDAR ground truth (supervision for `training.training_step`; no dataset ships with the reference).

The reference trains on nuScenes sweeps: per LiDAR ray a range, an intensity and a class id (ZI/datasets.py:640-705, keys `depth`,
`intensity`, `semantic`, `lidar_mask`).  This module supplies the same keys for rays cast into a closed analytic scene - a ground
plane with road / sidewalk / terrain bands, four boundary walls, and axis-aligned boxes standing in for cars, a truck, poles,
vegetation and building blocks - so that a model can be TRAINED in this image and the whole chain (train -> checkpoint in the
reference's format -> fused render) runs on a field that looks like the reference's use: smooth, concentrated density, not
white-noise tables.

Geometry is stated in metres in the SENSOR-aligned frame (z up, the frame `lidar.get_directions` builds directions in);
`lidar.synthetic_sweep` rotates directions by `lidar.seeded_rotation(seed)` and works in scene units (metres * scale_factor), so
`cast` takes the rotation and the scale.  Everything is torch and runs on whatever device the rays live on.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import lidar as nlidar
from . import synth

# class ids of the 19-class label space the reference's semantic head uses (Cityscapes train ids; sky = 10 and car = 13 are the
# two the ray-drop stage singles out, RD/drop_simulation_rays.py:120-150)
ROAD, SIDEWALK, BUILDING, WALL, POLE, VEGETATION, TERRAIN, CAR, TRUCK = 0, 1, 2, 3, 5, 8, 9, 13, 14

GROUND_Z = -6.2                       # below every sensor position: sweep origins move by up to 0.01 * sqrt(3) scene units = 4.3 m
WALL_X, WALL_Y = 60.0, 35.0           # boundary walls: the scene is closed, every ray returns
# boxes: (centre xyz, size xyz, class)
_BOXES = [
    ((8.0, 3.2, GROUND_Z + 0.8), (4.6, 1.9, 1.6), CAR),
    ((-12.0, -3.4, GROUND_Z + 0.8), (4.4, 1.8, 1.6), CAR),
    ((22.0, -3.0, GROUND_Z + 0.75), (4.5, 1.9, 1.5), CAR),
    ((-30.0, 3.1, GROUND_Z + 0.8), (4.7, 2.0, 1.6), CAR),
    ((35.0, 3.5, GROUND_Z + 1.7), (9.0, 2.6, 3.4), TRUCK),
    ((5.0, 9.5, GROUND_Z + 3.0), (0.4, 0.4, 6.0), POLE),
    ((-18.0, -9.5, GROUND_Z + 3.0), (0.4, 0.4, 6.0), POLE),
    ((28.0, 9.5, GROUND_Z + 3.0), (0.4, 0.4, 6.0), POLE),
    ((-6.0, 14.0, GROUND_Z + 2.0), (5.0, 4.0, 4.0), VEGETATION),
    ((16.0, -15.0, GROUND_Z + 2.5), (6.0, 5.0, 5.0), VEGETATION),
    ((-40.0, 22.0, GROUND_Z + 7.0), (18.0, 14.0, 14.0), BUILDING),
    ((30.0, -26.0, GROUND_Z + 5.0), (22.0, 12.0, 10.0), BUILDING),
    ((-8.0, 27.0, GROUND_Z + 6.0), (16.0, 10.0, 12.0), BUILDING),
]
# per class: LiDAR reflectivity and an albedo for the colour target
_REFLECT = {ROAD: 0.18, SIDEWALK: 0.32, BUILDING: 0.55, WALL: 0.45, POLE: 0.85, VEGETATION: 0.38, TERRAIN: 0.25, CAR: 0.7, TRUCK: 0.6}
_ALBEDO = {ROAD: (0.25, 0.25, 0.27), SIDEWALK: (0.55, 0.53, 0.5), BUILDING: (0.7, 0.6, 0.5), WALL: (0.6, 0.6, 0.62), POLE: (0.3, 0.3, 0.3),
           VEGETATION: (0.2, 0.5, 0.2), TERRAIN: (0.45, 0.5, 0.3), CAR: (0.7, 0.15, 0.15), TRUCK: (0.85, 0.85, 0.9)}


def _tables(device):
    refl = torch.zeros(19, device=device)
    alb = torch.zeros(19, 3, device=device)
    for k, v in _REFLECT.items():
        refl[k] = v
    for k, v in _ALBEDO.items():
        alb[k] = torch.tensor(v, device=device)
    return refl, alb


def cast(origins: torch.Tensor, directions: torch.Tensor, rotation=None, scale_factor: float = 1.0 / 250.0) -> Dict[str, torch.Tensor]:
    """First hit of every ray.  origins / directions [N,3] in scene units / the rotated frame of `lidar.synthetic_sweep`
    (`rotation` = `lidar.seeded_rotation(seed)`, None for the sensor frame itself).  Returns
    depth [N] (scene units along the unit direction: what `renderings[-1]['depth']` is compared with, train.py:334),
    semantic [N] int64, intensity [N] in [0, 1], rgb [N,3], normal [N,3] (sensor frame)."""
    dev = origins.device
    o = origins.double() / scale_factor
    d = directions.double()
    if rotation is not None:
        R = torch.as_tensor(np.asarray(rotation), dtype=torch.float64, device=dev)
        o, d = o @ R, d @ R                       # d_world = d_sensor @ R^T  =>  d_sensor = d_world @ R
    n = o.shape[0]
    inf = torch.full((n,), float("inf"), dtype=torch.float64, device=dev)
    best_t, best_cls = inf.clone(), torch.full((n,), -1, dtype=torch.int64, device=dev)
    best_n = torch.zeros(n, 3, dtype=torch.float64, device=dev)

    def take(t, cls, normal):
        nonlocal best_t, best_cls, best_n
        ok = (t > 1e-6) & (t < best_t)
        best_t = torch.where(ok, t, best_t)
        best_cls = torch.where(ok, cls if isinstance(cls, torch.Tensor) else torch.full_like(best_cls, cls), best_cls)
        best_n = torch.where(ok[:, None], normal, best_n)

    safe = lambda x: torch.where(x.abs() < 1e-12, torch.full_like(x, 1e-12), x)
    # ground: class by lateral band of the hit point
    tg = (GROUND_Z - o[:, 2]) / safe(d[:, 2])
    yg = (o[:, 1] + tg * d[:, 1]).abs()
    cls_g = torch.where(yg < 7.0, ROAD, torch.where(yg < 12.0, SIDEWALK, TERRAIN))
    take(tg, cls_g, torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64, device=dev).expand(n, 3))
    # boundary walls
    for axis, pos, cls in ((0, WALL_X, WALL), (0, -WALL_X, WALL), (1, WALL_Y, BUILDING), (1, -WALL_Y, BUILDING)):
        t = (pos - o[:, axis]) / safe(d[:, axis])
        nrm = torch.zeros(3, dtype=torch.float64, device=dev)
        nrm[axis] = -np.sign(pos)
        take(t, cls, nrm.expand(n, 3))
    # boxes: slab test, all boxes at once ([N, B, 3])
    key = str(dev)
    if key not in _BOX_T:
        _BOX_T[key] = (torch.tensor([b[0] for b in _BOXES], dtype=torch.float64, device=dev),
                       torch.tensor([b[1] for b in _BOXES], dtype=torch.float64, device=dev) / 2,
                       torch.tensor([b[2] for b in _BOXES], dtype=torch.int64, device=dev))
    bc, bh, bcls = _BOX_T[key]
    inv = (1.0 / safe(d))[:, None, :]
    t0, t1 = (bc - bh - o[:, None, :]) * inv, (bc + bh - o[:, None, :]) * inv
    tn, ax = torch.minimum(t0, t1).max(dim=2)                       # entry parameter and the axis of the face entered
    tf = torch.maximum(t0, t1).min(dim=2).values
    tn = torch.where((tn < tf) & (tn > 1e-6), tn, inf[:, None])
    tb, ib = tn.min(dim=1)                                           # nearest box per ray
    axb = torch.gather(ax, 1, ib[:, None])
    nrm = torch.zeros(n, 3, dtype=torch.float64, device=dev)
    nrm.scatter_(1, axb, -torch.sign(torch.gather(d, 1, axb)))
    take(tb, bcls[ib], nrm)
    assert bool((best_cls >= 0).all()), "the scene is closed: every ray must hit something"
    refl, alb = _tables(dev)
    cosi = (-(d / d.norm(dim=-1, keepdim=True)) * best_n).sum(-1).clamp(0.0, 1.0)
    shade = (0.4 + 0.6 * cosi).float()
    # the ray is o + t * d with d as given (the renderer's t runs along `directions`, unit or not): t in metres -> scene units
    return dict(depth=(best_t * scale_factor).float(), semantic=best_cls, intensity=refl[best_cls] * shade, rgb=alb[best_cls] * shade[:, None], normal=best_n.float())


_POSITIONS: Dict[tuple, tuple] = {}
_BOX_T: Dict[str, tuple] = {}


def random_lidar_rays(n: int, seed: int, step: int, device, rot_seed: int = 0, scale_factor: float = 1.0 / 250.0,
                      origin_range: float = 0.01) -> Dict[str, torch.Tensor]:
    """A training batch of `n` LiDAR rays with the batch contract of `lidar.cast_lidar_ray_batch` (ZI/lidar_utils.py:8-33, incl. the
    Frobenius-norm `viewdirs` and `base_x = base_y = directions`): origins drawn from the sensor positions of
    `lidar.synthetic_sweep(sweep_idx = 0..63)`, beams from the nuScenes table, azimuths uniform.  Built on `device` (a few small
    kernels; on the host the same arithmetic costs more than the training step).  Deterministic in (seed, step) per device type."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed * 1000003 + step)
    key = (rot_seed, origin_range, str(dev))
    if key not in _POSITIONS:
        pos = np.stack([synth.uniform(rot_seed, 9100 + i, (3,), -origin_range, origin_range) for i in range(64)]).astype(np.float64)
        _POSITIONS[key] = (torch.from_numpy(pos).to(dev), torch.tensor(nlidar.LIDAR_ANGLES, dtype=torch.float64, device=dev) / 180 * np.pi,
                           torch.from_numpy(nlidar.seeded_rotation(rot_seed)).to(dev))
    pos, beams, R = _POSITIONS[key]
    o = pos[torch.randint(0, 64, (n,), generator=g, device=dev)]
    th = beams[torch.randint(0, 32, (n,), generator=g, device=dev)]
    ph = torch.rand(n, generator=g, dtype=torch.float64, device=dev) * 2 * np.pi
    d = torch.stack([torch.cos(th) * torch.sin(ph), torch.cos(th) * torch.cos(ph), torch.sin(th)], -1).float().double() @ R.T
    f32 = lambda t: t.float().contiguous()
    col = lambda v: torch.full((n, 1), float(v), device=dev)
    d32 = f32(d)
    return dict(origins=f32(o), directions=d32, viewdirs=f32(d / torch.linalg.norm(d)),  # Frobenius norm of the whole array (sic)
                radii=col(0.0005), imageplane=torch.zeros(n, 2, device=dev), lossmult=col(1.0), near=col(2.0 * scale_factor),
                far=col(500.0 * scale_factor), cam_idx=col(-1.0), base_x=d32, base_y=d32, rgb=torch.zeros(n, 3, device=dev),
                semantic=torch.full((n,), 255.0, device=dev), mask=torch.ones(n, device=dev))


def supervise(batch: Dict[str, torch.Tensor], rot_seed: int = 0, scale_factor: float = 1.0 / 250.0) -> Dict[str, torch.Tensor]:
    """Adds the supervision keys train.py:283-424 reads for LiDAR rays (`rgb`, `depth`, `semantic`, `intensity`, and the masks
    `losses.total_loss` takes) to a ray batch, from the analytic scene."""
    n = batch["origins"].shape[0]
    gt = cast(batch["origins"].reshape(n, 3), batch["directions"].reshape(n, 3), nlidar.seeded_rotation(rot_seed), scale_factor)
    out = dict(batch)
    ones = torch.ones(n, dtype=torch.bool, device=gt["depth"].device)
    out.update(rgb=gt["rgb"], depth=gt["depth"], semantic=gt["semantic"], intensity=gt["intensity"], mask_rgb=ones, depth_mask=ones,
               sem_mask=ones, lidar_mask=ones)
    return out
