"""Azimuth-sector sharding of a LiDAR sweep over the GPUs of one node (BASELINE config C4).

Replaces the reference's inference parallelism (ZI/models.py:1425-1437 rank slices of each chunk and
ZI/models.py:1454-1457 one `accelerator.gather` per output key per chunk, ~12 small all_gathers) by:
GPU p renders columns [p*W/P, (p+1)*W/P) of every beam and ONE all-gather (RCCL over xGMI;
`torch.distributed` backend "nccl") puts the whole sweep on every rank.

Layout.  A rank's tile is AZIMUTH-MAJOR, `[W/P, H, 7]` (7 floats per ray: depth, intensity, acc, rgb[3], label),
written by the compositing kernel itself (`NlrOut.packed`).  The rank-major concatenation that an all-gather produces is
then the `[W, H, 7]` image of the sweep: no pack pass before the collective and no transpose after it.  `as_hw` gives the
`[H, W, 7]` view the ray-drop UNet side indexes.  Rays are independent, so the gathered image equals the one-GPU image
bit for bit.

Overlap.  `SweepGatherer` issues the collective on a side stream and double-buffers tile and image, so the all-gather of
sweep i runs under the resampling / proposal / encode kernels of sweep i+1 (the message is 0.9 MB per rank at 8 x 4096
rays: latency-bound, tens of microseconds).  `render_fn` is injected so that partition, gather and reassembly are
testable on CPU with the gloo backend.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import lidar

RECORD = 7  # depth, intensity, acc, rgb[3], label
PACK_KEYS = ("depth", "intensity", "acc", "rgb", "labels")


def _world(group=None):
    return (dist.get_world_size(group), dist.get_rank(group)) if dist.is_initialized() else (1, 0)


def pack_tile(r: Dict[str, torch.Tensor], height: int, wp: int) -> torch.Tensor:
    """Beam-major per-ray outputs [H*wp] -> azimuth-major tile [wp, H, 7] (labels travel as float; exact below 2^24).
    Host-side equivalent of what the compositing kernel writes into `NlrOut.packed`; used where the renderer is a stand-in."""
    n = height * wp
    tile = torch.zeros(n, RECORD, device=r["depth"].device)
    col = 0
    for k in PACK_KEYS:
        width = 3 if k == "rgb" else 1
        if r.get(k) is not None:
            tile[:, col:col + width] = r[k].reshape(n, width).float()
        col += width
    return tile.reshape(height, wp, RECORD).permute(1, 0, 2).contiguous()


def as_hw(img: torch.Tensor) -> torch.Tensor:
    """[W, H, 7] sweep image -> [H, W, 7] view (no copy)."""
    return img.permute(1, 0, 2)


def unpack_image(img: torch.Tensor) -> Dict[str, torch.Tensor]:
    """[W, H, 7] sweep image -> dict of [H, W(, 3)] views."""
    v = as_hw(img)
    return dict(depth=v[..., 0], intensity=v[..., 1], acc=v[..., 2], rgb=v[..., 3:6], labels=v[..., 6].round().to(torch.int32))


class SweepGatherer:
    """Double-buffered, side-stream all-gather of azimuth-major tiles.

        g = SweepGatherer(height, width, device)
        for i, sweep in enumerate(sweeps):
            tile = g.tile(i)            # [wp, H, 7] buffer the renderer writes (NlrOut.packed); safe to overwrite
            render(sweep, packed=tile)
            g.submit(i)                 # collective on the side stream, behind the render of sweep i
        img = g.image(i)                # [W, H, 7]; the current stream now waits for gather i

    With one rank (and `force=False`) the tile IS the image and nothing is launched.  On CPU (gloo tests) the collective
    is issued synchronously."""

    def __init__(self, height: int, width: int, device, group=None, force: bool = False):
        self.world, self.rank = _world(group)
        self.group = group
        self.height, self.width = height, width
        self.wp = -(-width // self.world)
        self.device = torch.device(device)
        self.collective = self.world > 1 or (force and dist.is_initialized())
        self.cuda = self.device.type == "cuda"
        mk = lambda *s: torch.zeros(*s, device=self.device)
        self.tiles = [mk(self.wp, height, RECORD) for _ in range(2)]
        self.images = [mk(self.world * self.wp, height, RECORD) for _ in range(2)] if self.collective else self.tiles
        self.side = torch.cuda.Stream(self.device) if (self.cuda and self.collective) else None
        self.done = [None, None]  # event: gather that last read tiles[b] / wrote images[b] has finished

    def tile(self, i: int) -> torch.Tensor:
        b = i & 1
        if self.side is not None and self.done[b] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.done[b])  # gather i-2 still reads this buffer
        return self.tiles[b]

    def submit(self, i: int) -> None:
        if not self.collective:
            return
        b = i & 1
        if self.side is None:
            dist.all_gather_into_tensor(self.images[b], self.tiles[b], group=self.group)
            return
        rendered = torch.cuda.Event()
        rendered.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(rendered)
            dist.all_gather_into_tensor(self.images[b], self.tiles[b], group=self.group)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.done[b] = ev

    def image(self, i: int) -> torch.Tensor:
        b = i & 1
        if self.side is not None and self.done[b] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.done[b])
        return self.images[b][:self.width]


def render_sweep_sharded(render_fn: Callable[..., Dict[str, torch.Tensor]], batch_np: Dict[str, np.ndarray], height: int,
                         width: int, device, group: Optional[dist.ProcessGroup] = None,
                         gatherer: Optional[SweepGatherer] = None, index: int = 0) -> torch.Tensor:
    """Render this rank's azimuth sector and all-gather the sweep.  Returns the [W, H, 7] image on every rank.

    batch_np is the FULL sweep's ray batch (host numpy, beam-major): the sector is sliced from it so that the LiDAR
    `viewdirs` keep the full-sweep Frobenius normalisation (ZI/lidar_utils.py:12) and a sector renders exactly what the
    same rays render inside a one-GPU sweep.  `render_fn(batch, packed=tile)` may fill the azimuth-major tile itself
    (the HIP renderer does) and return None / a dict with "packed"; if it returns per-ray outputs they are packed here.
    """
    g = gatherer or SweepGatherer(height, width, device, group)
    sec, wp = lidar.azimuth_sector(batch_np, height, width, g.rank, g.world)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in sec.items()}
    tile = g.tile(index)
    r = render_fn(batch, packed=tile)
    if isinstance(r, dict) and "packed" not in r:
        tile.copy_(pack_tile(r, height, wp))
    g.submit(index)
    return g.image(index)
