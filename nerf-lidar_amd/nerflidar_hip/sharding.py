"""Azimuth-sector sharding of a LiDAR sweep over the GPUs of one node.

Replaces the reference's inference parallelism (ZI/models.py:1425-1437 rank slices of each chunk and
ZI/models.py:1454-1457 one `accelerator.gather` per output key per chunk, ~12 small all_gathers) by:
GPU p renders columns [p*W/P, (p+1)*W/P) of every beam, packs its outputs into ONE tile
[H, W/P, C_pack] and a single all-gather (RCCL over xGMI; `torch.distributed` backend "nccl")
reassembles the [H, W, C_pack] range image on every rank.  Rays are independent, so the gathered image
equals the single-GPU image bit for bit.  `render_fn` is injected so the partition / pack / gather /
reassembly logic is testable on CPU with the gloo backend.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import lidar

PACK_KEYS = ("depth", "intensity", "acc", "rgb", "labels")  # 1 + 1 + 1 + 3 + 1 = 7 floats per ray


def pack_tile(r: Dict[str, torch.Tensor], height: int, wp: int) -> torch.Tensor:
    """[H*wp] outputs -> [H, wp, 7] float32 tile (labels travel as float; exact for class ids < 2^24)."""
    n = height * wp
    cols = []
    for k in PACK_KEYS:
        t = r.get(k)
        if t is None:
            t = torch.zeros(n, 3 if k == "rgb" else 1, device=r["depth"].device)
        cols.append(t.reshape(n, -1).float())
    return torch.cat(cols, dim=1).reshape(height, wp, -1).contiguous()


def unpack_image(img: torch.Tensor) -> Dict[str, torch.Tensor]:
    return dict(depth=img[..., 0], intensity=img[..., 1], acc=img[..., 2], rgb=img[..., 3:6],
                labels=img[..., 6].round().to(torch.int32))


def render_sweep_sharded(render_fn: Callable[[Dict[str, torch.Tensor]], Dict[str, torch.Tensor]],
                         batch_np: Dict[str, np.ndarray], height: int, width: int, device,
                         group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Render this rank's azimuth sector and all-gather the range image.  Returns [H, W, 7] on every rank.

    batch_np is the FULL sweep's ray batch (host numpy, beam-major): the sector is sliced from it so that the
    LiDAR `viewdirs` keep the full-sweep Frobenius normalisation (ZI/lidar_utils.py:12) and a sector renders
    exactly what the same rays render inside a one-GPU sweep.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sec, wp = lidar.azimuth_sector(batch_np, height, width, rank, world)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in sec.items()}
    tile = pack_tile(render_fn(batch), height, wp)
    return gather_tiles(tile, width, group)


def gather_tiles(tile: torch.Tensor, width: int, group=None, force: bool = False) -> torch.Tensor:
    """ONE all-gather of the packed [H, W/P, C] tile -> [H, W, C] (pad columns stripped).  `force` runs the collective
    even with a single rank (rehearsal of the RCCL path on a one-GPU box)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
        return tile[:, :width]
    h, wp, c = tile.shape
    out = torch.empty(world * h, wp, c, device=tile.device, dtype=tile.dtype)  # rank-major concatenation
    dist.all_gather_into_tensor(out, tile, group=group)
    return out.view(world, h, wp, c).permute(1, 0, 2, 3).reshape(h, world * wp, c)[:, :width].contiguous()
