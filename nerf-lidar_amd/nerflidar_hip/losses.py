"""Loss terms of the training step (scope row f-3): what `train.py:326-446` sums before `accelerator.backward`.

Small elementwise / scan work on [rays, samples] arrays, written with torch ops so that autograd carries the gradients to
the stages that have HIP backward kernels (compositing, fused MLP, hash grid).  Each function states the reference lines it
follows; `tests/golden/fn_losses.npz` holds the reference's own values and gradients for the first four.

Formulations differ from the reference where that is cheaper and numerically at least as good:
  * interval look-ups use one `torch.searchsorted` per query instead of an [.., n, m] comparison cube
    (`stepfun.searchsorted`, `math.sorted_interp_quad`);
  * the pairwise term of the distortion loss is a running sum (O(S), every term non-negative) instead of an [.., S, S] array.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

_EPS = float(torch.finfo(torch.float32).eps)


def _bracket(knots: torch.Tensor, q: torch.Tensor):
    """For sorted `knots` [.., n] and queries `q` [.., m]: (lo, hi) with knots[lo] <= q < knots[hi]; lo = 0 when no knot is
    <= q, hi = n - 1 when none is > q (the index pair of `stepfun.searchsorted`, stepfun.py:6-26)."""
    n = knots.shape[-1]
    cnt = torch.searchsorted(knots.contiguous(), q.contiguous(), right=True)
    return (cnt - 1).clamp_min(0), cnt.clamp_max(n - 1)


# ---- mip-NeRF 360 interlevel loss (train_utils.py:120-131, stepfun.py:37-62) --------------------------------------------------
def outer_measure(t: torch.Tensor, t_env: torch.Tensor, w_env: torch.Tensor) -> torch.Tensor:
    """Mass of the envelope histogram (t_env, w_env) in every interval of `t`, counting each envelope bin the interval touches
    whole (`inner_outer`'s y0_outer)."""
    cum = torch.cat([torch.zeros_like(w_env[..., :1]), torch.cumsum(w_env, dim=-1)], dim=-1)
    lo, hi = _bracket(t_env, t)
    return torch.gather(cum, -1, hi[..., 1:]) - torch.gather(cum, -1, lo[..., :-1])


def interlevel_loss(ray_history: Sequence[Dict[str, torch.Tensor]], mult: float) -> torch.Tensor:
    c, w = ray_history[-1]["sdist"].detach(), ray_history[-1]["weights"].detach()
    total = 0.0
    for h in ray_history[:-1]:
        excess = (w - outer_measure(c, h["sdist"], h["weights"])).clamp_min(0)
        total = total + (excess ** 2 / (w + _EPS)).mean()
    return mult * total


# ---- Zip-NeRF anti-aliased interlevel loss (train_utils.py:134-172, stepfun.py:425-433, math.py:111-131) ----------------------
def blur_stepfun(x: torch.Tensor, y: torch.Tensor, r: float):
    """Step function (x [.., n+1], y [.., n]) convolved with a box of half-width r: a piecewise-linear function given by its
    2n + 2 knots and the values there.  Every edge of the step function turns into a ramp from x - r to x + r: sort the ramp
    ends, integrate the slope changes once for the slope and once more for the value."""
    jump = (torch.cat([y, torch.zeros_like(y[..., :1])], dim=-1) - torch.cat([torch.zeros_like(y[..., :1]), y], dim=-1)) / (2 * r)
    knots, order = torch.sort(torch.cat([x - r, x + r], dim=-1), dim=-1)
    slope = torch.cumsum(torch.gather(torch.cat([jump, -jump], dim=-1), -1, order[..., :-1]), dim=-1)
    val = torch.cumsum(torch.diff(knots, dim=-1) * slope, dim=-1).clamp_min(0)
    return knots, torch.cat([torch.zeros_like(val[..., :1]), val], dim=-1)


def _quad_cdf_at(q: torch.Tensor, knots: torch.Tensor, pdf: torch.Tensor, cdf: torch.Tensor) -> torch.Tensor:
    """`math.sorted_interp_quad`: the cdf of a piecewise-linear pdf at the queries.  The reference picks the pdf at the two ends
    of the bracketing interval as the max of pdf over the knots <= q and the min over the knots > q (its `find_interval` assumes
    a sorted array; the pdf is not one): reproduced with a running max / reverse running min."""
    lo, hi = _bracket(knots, q)
    run_max = torch.cummax(pdf, dim=-1).values
    rev_min = torch.flip(torch.cummin(torch.flip(pdf, dims=[-1]), dim=-1).values, dims=[-1])
    p0, p1 = torch.gather(run_max, -1, lo), torch.gather(rev_min, -1, hi)
    x0, x1 = torch.gather(knots, -1, lo), torch.gather(knots, -1, hi)
    # (a cdf and sorted knots are non-decreasing: their "max over the knots <= q" is the value at lo)
    c0 = torch.gather(cdf, -1, lo)
    frac = torch.nan_to_num((q - x0) / (x1 - x0), nan=0.0).clamp(0, 1)
    return c0 + (q - x0) * (p0 + p1 * frac + p0 * (1 - frac)) / 2


def anti_interlevel_loss(ray_history: Sequence[Dict[str, torch.Tensor]], mult: float, pulse_width: Sequence[float]) -> torch.Tensor:
    c, w = ray_history[-1]["sdist"].detach(), ray_history[-1]["weights"].detach()
    pdf = (w / (c[..., 1:] - c[..., :-1])).clamp_max(10)
    total = 0.0
    for i, h in enumerate(ray_history[:-1]):
        knots, val = blur_stepfun(c, pdf, pulse_width[i])
        area = 0.5 * (val[..., 1:] + val[..., :-1]) * torch.diff(knots, dim=-1)
        cdf = torch.cat([torch.zeros_like(area[..., :1]), torch.cumsum(area, dim=-1)], dim=-1)
        w_target = torch.diff(_quad_cdf_at(h["sdist"], knots, val, cdf), dim=-1)
        wp = h["weights"]
        term = (w_target - wp).clamp_min(0) ** 2 / (wp + 1e-5)
        if "obj_mask" in h:  # samples inside dynamic-object boxes are not held to the static envelope (train_utils.py:153-154)
            term = term[~h["obj_mask"]]
        total = total + term.mean()
    return mult * total


# ---- distortion loss (train_utils.py:175-181, stepfun.py:297-308) -------------------------------------------------------------
def distortion(t: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """iint w_i w_j |u_i - u_j| over the interval midpoints u + the within-interval term, per ray.
    D_i = sum_{j<i} w_j (u_i - u_j) obeys D_i = D_{i-1} + (sum_{j<i} w_j)(u_i - u_{i-1}): one scan, no cancellation."""
    u = 0.5 * (t[..., 1:] + t[..., :-1])
    before = torch.cumsum(w, dim=-1) - w
    du = torch.cat([torch.zeros_like(u[..., :1]), torch.diff(u, dim=-1)], dim=-1)
    d = torch.cumsum(before * du, dim=-1)
    return 2 * (w * d).sum(dim=-1) + (w ** 2 * (t[..., 1:] - t[..., :-1])).sum(dim=-1) / 3


def distortion_loss(ray_history: Sequence[Dict[str, torch.Tensor]], mult: float) -> torch.Tensor:
    return mult * distortion(ray_history[-1]["sdist"], ray_history[-1]["weights"]).mean()


# ---- data terms ----------------------------------------------------------------------------------------------------------------
def data_loss(renderings: Sequence[Dict[str, torch.Tensor]], rgb: torch.Tensor, ray_mask: Optional[torch.Tensor] = None,
              kind: str = "charb", charb_padding: float = 1e-3, coarse_mult: float = 0.0, mult: float = 1.0) -> torch.Tensor:
    """train_utils.py:55-117: per level the masked mean of the (Charbonnier | squared) colour residual; the proposal levels
    enter with `data_coarse_loss_mult`, the final one with `data_loss_mult`."""
    m = torch.ones_like(rgb[..., :1]) if ray_mask is None else ray_mask[..., None].to(rgb.dtype)
    m = m.expand_as(rgb[..., :3])
    denom = m.sum()
    per_level = []
    for r in renderings:
        sq = (r["rgb"] - rgb[..., :3]) ** 2
        if kind == "charb":
            e = torch.sqrt(sq + charb_padding ** 2)
        elif kind == "mse":
            e = sq
        else:
            raise ValueError(f"data_loss_type {kind!r} is not supported (charb, mse)")
        per_level.append((m * e).sum() / denom.clamp_min(1))        # (an empty mask: 0 / 1, as the reference's guard; no host read of denom)
    return coarse_mult * sum(per_level[:-1]) + mult * per_level[-1]


# The masked terms below select with `torch.where` and divide by the mask count instead of indexing with a boolean mask: `x[mask]` has a
# data-dependent shape, i.e. a device-to-host read in the middle of every training step.  Same values up to the order of a float32 sum.
def depth_loss(depth: torch.Tensor, target: torch.Tensor, mask: torch.Tensor, lam: float = 0.1) -> torch.Tensor:
    """train.py:330-341: log(|d| + 1) averaged over the residuals BELOW the 0.9 quantile of |d| (the comparison is on the signed
    residual, as in the reference: every negative residual stays in).  The quantile is torch.quantile's (linear interpolation between
    the two neighbouring order statistics, rank 0.9 (n - 1) in float32) taken from a sort with the unmasked rays pushed to +inf."""
    d = (depth - target).reshape(-1)
    mask = mask.reshape(-1)
    cnt = mask.sum()
    srt = torch.sort(torch.where(mask, d.detach().abs(), torch.full_like(d, float("inf")))).values
    pos = (cnt - 1).clamp_min(0).to(d.dtype) * 0.9                    # (a Python scalar: a device tensor made from one is a host copy)
    lo = pos.floor()
    pick = lambda i: srt.gather(0, i.long().reshape(1))[0]             # (srt[tensor_index] would read the index back to the host)
    thr = torch.lerp(pick(lo), pick(pos.ceil()), pos - lo)
    sel = mask & (d < thr)
    # no masked ray: 0 (the reference's `numel() == 0` guard); masked rays but none below the quantile: 0 / 0, as `mean()` of nothing
    n_sel = torch.where(cnt > 0, sel.sum(), torch.ones_like(cnt)).to(d.dtype)
    return lam * torch.where(sel, torch.log(d.abs() + 1), torch.zeros_like(d)).sum() / n_sel


def semantic_loss(prob: torch.Tensor, label: torch.Tensor, mask: torch.Tensor, lam: float = 0.01) -> torch.Tensor:
    """train.py:407-418: NLL of log(p + 1e-6) on the labelled rays."""
    K = prob.shape[-1]
    prob, mask = prob.reshape(-1, K), mask.reshape(-1)
    lab = torch.where(mask, label.reshape(-1).long(), torch.zeros_like(label.reshape(-1).long())).clamp(0, K - 1)
    lp = torch.log(prob + 1e-6).gather(1, lab[:, None])[:, 0]
    return -lam * torch.where(mask, lp, torch.zeros_like(lp)).sum() / mask.sum().clamp_min(1)


def intensity_loss(pred: torch.Tensor, target: torch.Tensor, lidar_mask: torch.Tensor) -> torch.Tensor:
    """train.py:419-424: mean squared error on the LiDAR rays, x 0.1."""
    d = pred.reshape(-1) - target.reshape(-1)
    m = lidar_mask.reshape(-1)
    return 0.1 * torch.where(m, d.pow(2), torch.zeros_like(d)).sum() / m.sum().clamp_min(1)


def nusc_masks(batch: Dict[str, torch.Tensor], lidar_supervision: bool = False, only_lidar_supervision: bool = False,
               instance_obj: bool = False, aug_road: bool = False) -> Dict[str, torch.Tensor]:
    """The per-ray masks of a nuScenes batch exactly as train.py:288-324 derives them (dataset_loader == 'nusc'), returned under the keys
    `total_loss` reads: mask_rgb, depth_mask, sem_mask, lidar_mask (bool).  Reference quirk kept: `batch['mask']` is first replaced by
    `mask == 0` (a bool) and the colour mask is then `that == 0`, i.e. colour is supervised where the ORIGINAL mask is non-zero."""
    m = batch["mask"] == 0                                           # train.py:288
    if instance_obj:
        m = torch.zeros_like(m)                                      # :290
    if aug_road:
        m = m.clone()
        m[batch["aug_mask"] == 1] = 1                                # :292
    patch = batch["patch_mask"] if "patch_mask" in batch else torch.zeros_like(batch["mask"])
    rgb = torch.logical_and(m == 0, patch == 0)                      # :311
    depth = torch.logical_and(batch["depth"] > 0, rgb)               # :313
    sem = torch.logical_and(batch["semantic"] != 255, rgb)           # :315
    lidar = (batch["lidar_mask"] == 1) if "lidar_mask" in batch else torch.zeros_like(rgb)
    if lidar_supervision:                                            # :317-323
        rgb, depth, sem = rgb.clone(), depth.clone(), sem.clone()
        rgb[lidar] = False
        depth[lidar] = True
        sem[lidar] = False
        if only_lidar_supervision:
            depth[~lidar] = False
    return dict(mask_rgb=rgb, depth_mask=depth, sem_mask=sem, lidar_mask=lidar)


def total_loss(renderings: List[Dict[str, torch.Tensor]], ray_history: List[Dict[str, torch.Tensor]], batch: Dict[str, torch.Tensor], *,
               data_kind: str = "charb", charb_padding: float = 1e-3, data_coarse_mult: float = 0.0, data_mult: float = 1.0,
               interlevel_mult: float = 0.0, anti_interlevel_mult: float = 0.01, pulse_width: Sequence[float] = (0.03, 0.003),
               distortion_mult: float = 0.005, depth_lam: float = 0.1, sem_lam: float = 0.01) -> Dict[str, torch.Tensor]:
    """The dictionary `losses` of train.py:326-446 for the terms this path covers (defaults = configs.py); the caller adds the
    hash-decay term (`training.hash_decay_loss`) and sums.  Batch keys (all optional except rgb): rgb [N,3], mask_rgb [N],
    depth [N] + depth_mask [N], semantic [N] + sem_mask [N], intensity [N] + lidar_mask [N]."""
    out = {"data": data_loss(renderings, batch["rgb"], batch.get("mask_rgb"), data_kind, charb_padding, data_coarse_mult, data_mult)}
    last = renderings[-1]
    if "depth" in batch:
        out["depth"] = depth_loss(last["depth"], batch["depth"], batch.get("depth_mask", batch["depth"] > 0), depth_lam)
    if "semantic" in batch and "semantic" in last:
        out["sem"] = semantic_loss(last["semantic"], batch["semantic"], batch.get("sem_mask", batch["semantic"] != 255), sem_lam)
    if "intensity" in batch and "intensity" in last:
        out["int"] = intensity_loss(last["intensity"], batch["intensity"], batch.get("lidar_mask", torch.ones_like(batch["intensity"], dtype=torch.bool)))
    if len(ray_history) > 1:
        if interlevel_mult > 0:
            out["interlevel"] = interlevel_loss(ray_history, interlevel_mult)
        if anti_interlevel_mult > 0:  # (same key as the reference: the Zip-NeRF term replaces the mip-NeRF 360 one)
            out["interlevel"] = anti_interlevel_loss(ray_history, anti_interlevel_mult, pulse_width)
    if distortion_mult > 0:
        out["distortion"] = distortion_loss(ray_history, distortion_mult)
    return out
