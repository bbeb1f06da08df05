"""`render_lidar`-equivalent driver (SURVEY 8b "who calls it"): sweeps -> fused render -> the files the reference writes.

Restates Z/render_lidar.py:106-162 around `nerflidar_hip.models.render_image`: per sweep, render the LiDAR ray batch with
`image=False`, then `points = (origins + depth * directions) / scale_factor`, `labels = argmax(semantic)` and save
`points_%04d.npy`, `points_semantic_%04d.npy`, `points_rgb_%04d.npy` under `<render_dir>/lidar_replay` (names at :157-162).
Optionally continues into the ray-drop stage without the `.npy` round trip (rows f-2 / f-4): GPU range projection ->
UNet feature stack -> UNet -> drop mask -> KITTI `.bin/.label`.

Inputs are either a reference checkpoint (`--ckpt`, nerflidar_hip.checkpoints) or seeded synthetic weights of a named
workload (`--workload`, there is no dataset or trained checkpoint in this image).  Sweep poses are synthetic
(nerflidar_hip.lidar.synthetic_sweep); a dataset loader is outside the hot path.
"""
from __future__ import annotations

import argparse
import os
import time
from typing import Dict, Optional

import numpy as np
import torch

from . import config as nconfig
from . import lidar as nlidar
from . import weights as nweights
from .models import Model, render_image


def render_sweep(model: Model, batch: Dict[str, np.ndarray], scale_factor: float, accelerator=None) -> Dict[str, torch.Tensor]:
    """One sweep through `render_image(..., image=False)` + the LiDAR post-step (render_lidar.py:128,142-161).
    Returns CUDA tensors: points [N,3] (metres), labels [N] int64, rgb [N,3], depth [N], semantic [N,K], intensity [N]."""
    dev = model.device
    tb = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in batch.items()}
    r = render_image(model, accelerator, tb, False, model.config, image=False)
    depth = r["depth"].reshape(-1)
    out = dict(depth=depth, rgb=r["rgb"], semantic=r["semantic"],
               points=(tb["origins"] + depth[:, None] * tb["directions"]) / scale_factor,
               labels=torch.argmax(r["semantic"], dim=-1))
    if "intensity" in r:
        out["intensity"] = r["intensity"].reshape(-1)
    return out


def save_sweep(out_dir: str, idx: int, res: Dict[str, torch.Tensor]) -> None:
    """File names and contents of render_lidar.py:157-162."""
    os.makedirs(out_dir, exist_ok=True)
    npy = lambda t: t.detach().cpu().numpy()
    np.save(os.path.join(out_dir, "points_{:04d}.npy".format(idx)), npy(res["points"]))
    np.save(os.path.join(out_dir, "points_semantic_{:04d}.npy".format(idx)), npy(res["labels"]))
    np.save(os.path.join(out_dir, "points_rgb_{:04d}.npy".format(idx)), npy(res["rgb"]))


def to_lidar_frame(points: torch.Tensor, origin_m: torch.Tensor, lidar2world: torch.Tensor) -> torch.Tensor:
    """World metres -> sensor frame (what NeRF_Lidar_code/src/nerf2world.py:22-38 does with the stored lidar2global pose):
    directions were built as d_world = d_lidar @ R^T, so p_lidar = (p_world - origin) @ R."""
    return (points - origin_m[None, :]) @ lidar2world


def drop_rays(res: Dict[str, torch.Tensor], points_lidar: torch.Tensor, unet, mask_thre: float = 0.5, place_car: bool = False,
              width: int = 1024):
    """Range image -> UNet features -> logits -> kept points/labels (rows f-2 + f-4), all on the GPU.  The range image is
    32 x 1024 whatever the sweep's azimuth count (RD/lidar_utils.py:23)."""
    from . import raydrop
    proj = raydrop.range_projection(points_lidar.double(), semantic=res["labels"].float(), rgb=res["rgb"], H=32, W=width)
    feats = raydrop.unet_features(proj, var=unet.n_channels == 6)
    with torch.no_grad():
        logits = unet(feats)
    logits = logits[0] if isinstance(logits, tuple) else logits
    return raydrop.apply_ray_drop(proj, logits[0], mask_thre=mask_thre, place_car=place_car), proj


def render_sweep_device(model: Model, batch: Dict[str, torch.Tensor], scale_factor: float) -> Dict[str, torch.Tensor]:
    """One sweep through ONE `nlr_render_rays` call (no chunk loop, nothing leaves the device): the LiDAR post-step outputs `points`
    (metres), `labels`, `rgb`, `depth`, `intensity` of render_lidar.py:142-161, written by the compositing kernel itself."""
    r, _ = model.render_rays(batch, compute_extras=False, scale_factor=scale_factor)
    return r


def sweep_unet_input(res: Dict[str, torch.Tensor], origin_m: torch.Tensor, lidar2world: torch.Tensor, var: bool = True, width: int = 1024):
    """Rendered sweep -> the UNet's input image (Generate_feature.py:144-167 without the .npy round trip): sensor frame, spherical range
    projection (RD/lidar_utils.py:215-282 on the GPU), feature stack.  Returns ([1, F, 32, width], the projection dict)."""
    from . import raydrop
    pts = to_lidar_frame(res["points"], origin_m, lidar2world)
    proj = raydrop.range_projection(pts.double(), semantic=res["labels"].float(), rgb=res["rgb"], H=32, W=width)
    return raydrop.unet_features(proj, var=var), proj


def analytic_drop_truth(batch: Dict[str, torch.Tensor], origin_m: torch.Tensor, lidar2world, scale_factor: float, seed: int, sweep_idx: int,
                        width: int = 1024):
    """What stands in for the REAL sweep the reference trains the ray-drop UNet against (ray_drop_train.py:80-90: `mask` and `range` of a
    recorded LiDAR frame): the analytic scene's first hits, with a return kept when its received power - reflectivity x incidence shading
    x (30 m / range)^2, the `intensity` of nerflidar_hip.scene over range squared - exceeds a per-ray noisy threshold.  Projected with the
    same range projection as the rendered sweep (the keep flag travels in the semantic channel).  -> gt_mask [32, width] int64,
    gt_range [32, width] (normalised log range where kept, 0 elsewhere)."""
    from . import raydrop, scene as nscene
    rot = nlidar.seeded_rotation(seed)
    gt = nscene.cast(batch["origins"], batch["directions"], rot, scale_factor)
    rng_m = gt["depth"] / scale_factor
    power = gt["intensity"] * (30.0 / rng_m.clamp_min(1.0)) ** 2
    g = torch.Generator(device=power.device).manual_seed(seed * 7919 + sweep_idx)
    keep = power > 0.08 * (0.5 + torch.rand(power.shape, device=power.device, generator=g))
    pts = (batch["origins"] + gt["depth"][:, None] * batch["directions"]) / scale_factor
    proj = raydrop.range_projection(to_lidar_frame(pts, origin_m, lidar2world).double(), semantic=keep.float(), H=32, W=width)
    mask = ((proj["proj_semantic"] == 1) & (proj["proj_mask"] == 1)).long()
    real = proj["proj_range"]
    lr = torch.clamp(torch.log2(torch.where(real < 0, torch.zeros_like(real), real) + 0.0001 + 1) / 6.5, 0, 1)
    return mask, lr * mask


def raydrop_batch(model: Model, sweep_ids, scale_factor: float = 1.0 / 250.0, seed: int = 0, width: int = 1024, var: bool = True,
                  batches=None, truth=None):
    """BASELINE config 5's input, end to end on the device: render the sweeps, project, stack -> img [B, F, 32, width], gt_mask [B, 32, width],
    gt_range [B, 32, width] for `raydrop.train_step` (B = len(sweep_ids)), plus the per-sweep projections for `raydrop.apply_ray_drop`.
    batches: the sweeps' ray batches already on the device (a replay keeps them there); truth: (gt_mask, gt_range) from an earlier call -
    the recorded frames do not change between epochs, only the rendering does."""
    dev = model.device
    rot = torch.from_numpy(nlidar.seeded_rotation(seed)).float().to(dev)
    if batches is None:
        batches = [{k: torch.from_numpy(v).to(dev) for k, v in nlidar.synthetic_sweep(width=width, seed=seed, scale_factor=scale_factor, sweep_idx=idx).items()}
                   for idx in sweep_ids]
    imgs, masks, ranges, projs = [], [], [], []
    for idx, b in zip(sweep_ids, batches):
        origin_m = b["origins"][0] / scale_factor
        res = render_sweep_device(model, b, scale_factor)
        img, proj = sweep_unet_input(res, origin_m, rot, var=var, width=width)
        imgs.append(img[0]); projs.append(proj)
        if truth is None:
            m, r = analytic_drop_truth(b, origin_m, rot, scale_factor, seed, idx, width)
            masks.append(m); ranges.append(r)
    gt_mask, gt_range = (torch.stack(masks), torch.stack(ranges)) if truth is None else truth
    return torch.stack(imgs), gt_mask, gt_range, projs


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    src = ap.add_mutually_exclusive_group()
    src.add_argument("--ckpt", help="reference checkpoint directory or file (checkpoint_<step>.ckpt)")
    src.add_argument("--workload", default="C2", help="synthetic weights of a named workload (REF, C1, C2, C2S)")
    ap.add_argument("--log2-hashmap", type=int, default=None, help="override the hash-map size of synthetic tables")
    ap.add_argument("--num-nerf-samples", type=int, default=None)
    ap.add_argument("--width", type=int, default=1024, help="azimuth columns per sweep (reference sweeps use 1100)")
    ap.add_argument("--sweeps", type=int, default=1)
    ap.add_argument("--scale-factor", type=float, default=1.0 / 250.0, help="scene_scale.npy of the reference (ZI/datasets.py:1233)")
    ap.add_argument("--render-dir", default="render_out")
    ap.add_argument("--precision", type=int, default=2, help="0 f32, 1 mixed, 2 fast (split-bf16 heads, bf16 view MLP)")
    ap.add_argument("--raydrop-unet", default=None, help="UNet .pth (reference state_dict keys); enables the ray-drop stage")
    ap.add_argument("--mask-thre", type=float, default=0.5)
    ap.add_argument("--place-car", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tracks", default=None, help=".npy [N_obj, T, 9] box tracks (center3, theta_z, wlh3, timestamp, id): renders with "
                    "the dynamic-object branch (Config.instance_obj=True); needs --track-classes")
    ap.add_argument("--track-classes", default=None, help="comma-separated nuScenes category per track, e.g. vehicle.car,vehicle.truck")
    ap.add_argument("--synthetic-tracks", type=int, default=0, help="N seeded boxes placed on rays of the sweep (no dataset in this image)")
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("render_lidar needs a GPU: the fused path has no CPU fallback")
    dynamic = bool(a.tracks or a.synthetic_tracks)
    tracks = classes = None
    if dynamic:
        from . import objects as nobj
        if a.tracks:
            tracks = np.load(a.tracks)
            classes = (a.track_classes or "").split(",")
        else:
            probe = nlidar.synthetic_sweep(width=a.width, seed=a.seed, scale_factor=a.scale_factor)
            tracks = nobj.synthetic_tracks(probe, a.synthetic_tracks, 20, a.seed, size=(0.02, 0.01, 0.008), depth=(0.02, 0.2))
            classes = (["vehicle.car", "vehicle.truck", "vehicle.bus.rigid"] * a.synthetic_tracks)[: a.synthetic_tracks]
    if a.ckpt and dynamic:
        from .checkpoints import dynamic_model_from_checkpoint
        base = nconfig.ModelConfig()
        if a.num_nerf_samples:
            base.num_nerf_samples = a.num_nerf_samples
        model, step, ignored = dynamic_model_from_checkpoint(a.ckpt, tracks, classes, base=base, precision=a.precision)
        print(f"restored step {step} with {len(classes)} tracks; {len(ignored)} keys ignored")
    elif a.ckpt:
        from .checkpoints import model_from_checkpoint
        base = nconfig.ModelConfig()
        if a.num_nerf_samples:
            base.num_nerf_samples = a.num_nerf_samples
        model, step, ignored = model_from_checkpoint(a.ckpt, base=base, precision=a.precision)
        print(f"restored step {step}; {len(ignored)} keys outside the fused path ignored")
    else:
        mc = nconfig.workload(a.workload, a.log2_hashmap)
        if a.num_nerf_samples:
            mc.num_nerf_samples = a.num_nerf_samples
        sd = nweights.synth_state_dict(mc, seed=a.seed, trained_like=True)
        if dynamic:
            cids = sorted({nobj.query_class(c) for c in classes})
            lg = a.log2_hashmap or 21
            sd.update(nweights.synth_object_state_dict({c: nconfig.obj_mlp_config(c, mc.config.latent_size, lg) for c in cids},
                                                       len(classes), seed=a.seed))
            mc.config.instance_obj = True
            model = nobj.DynamicModel(mc, sd, tracks, classes, precision=a.precision, obj_log2_hashmap=lg)
        else:
            model = Model(mc, sd, precision=a.precision)
    unet = None
    if a.raydrop_unet:
        from . import raydrop
        sd = torch.load(a.raydrop_unet, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        n_ch = sd["inc.double_conv.0.weight"].shape[1]
        unet = raydrop.UNet(n_ch, 2, bilinear="up1.up.weight" not in sd, regression="outr.conv.weight" in sd).to(model.device)
        unet.load_state_dict(sd)
        unet.eval()
    out_dir = os.path.join(a.render_dir, "lidar_replay")
    rot = torch.from_numpy(nlidar.seeded_rotation(a.seed)).float().to(model.device)
    for idx in range(a.sweeps):
        batch = nlidar.synthetic_sweep(width=a.width, seed=a.seed, scale_factor=a.scale_factor, sweep_idx=idx)
        if dynamic:  # one sweep = one instant (ZI/datasets.py: per-ray timestamps of a sweep are its capture time)
            batch["timestamp"] = np.full((batch["origins"].shape[0], 1), idx / max(a.sweeps - 1, 1), np.float32)
        t0 = time.time()
        res = render_sweep(model, batch, a.scale_factor)
        torch.cuda.synchronize()
        dt = time.time() - t0
        save_sweep(out_dir, idx, res)
        n = res["points"].shape[0]
        print(f"sweep {idx + 1}/{a.sweeps}: {n} rays in {dt:0.3f}s ({n / dt:,.0f} rays/s), {int(res['labels'].unique().numel())} labels")
        if unet is not None:
            from . import raydrop
            origin_m = torch.from_numpy(batch["origins"][0]).to(model.device) / a.scale_factor
            (pts, lab), _ = drop_rays(res, to_lidar_frame(res["points"], origin_m, rot), unet, a.mask_thre, a.place_car)
            raydrop.write_points_and_labels(idx, os.path.join(a.render_dir, "raydrop"), pts, lab)
            print(f"  ray-drop kept {pts.shape[0]} of {n} points")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
