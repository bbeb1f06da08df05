"""Checkpoint importer (scope row f-4): the reference's on-disk format -> reference-keyed `state_dict` -> `Model`.

On-disk format (ZI/checkpoints.py:26-82, Z/train.py:559-566): `<ckpt_dir>/checkpoint_<step>.ckpt`, a `torch.save`d dict
`{'step': int, 'state_dict': Model.state_dict(), 'optimizer': ...}`; the newest file is the one whose trailing number is
largest.  `nerflidar_hip.models.Model` already consumes reference-keyed parameters (`nerf_mlp.* / prop_mlp_{i}.*`,
nerflidar_hip/weights.py), so importing is: read the file, keep the keys the fused path uses, derive the architecture
from the tensor shapes (no gin file needed) and report everything else (`obj_mlp_*` of the dynamic-object branch,
GLO vectors) as ignored instead of silently dropping it.
"""
from __future__ import annotations

import dataclasses
import glob
import os
import re
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .config import Config, MLPConfig, ModelConfig
from .weights import grid_layout, mlp_names, mlp_param_shapes


def _checkpoint_path(ckpt_dir: str, step, prefix: str = "checkpoint_") -> str:
    return os.path.join(ckpt_dir, f"{prefix}{step}.ckpt")


def latest_checkpoint(ckpt_dir, prefix: str = "checkpoint_") -> Optional[str]:
    """Newest `<prefix><step>.ckpt` by numeric step (ZI/checkpoints.py:12-23)."""
    files = glob.glob(os.path.join(os.fspath(ckpt_dir), f"{prefix}*"))
    files = sorted(files, key=lambda s: float(s[:-5].split("_")[-1]))
    return files[-1] if files else None


def save_checkpoint(ckpt_dir, state_dict: Dict[str, object], step: int, optimizer_state=None, prefix: str = "checkpoint_") -> str:
    """Writes the reference's format (ZI/checkpoints.py:58-82) from a reference-keyed parameter dict."""
    os.makedirs(ckpt_dir, exist_ok=True)
    sd = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in state_dict.items()}
    blob = {"step": int(step), "state_dict": sd}
    if optimizer_state is not None:
        blob["optimizer"] = optimizer_state
    path = _checkpoint_path(os.fspath(ckpt_dir), step, prefix)
    torch.save(blob, path)
    return path


def load_checkpoint(ckpt_dir_or_file, step=None, prefix: str = "checkpoint_") -> Tuple[Dict[str, np.ndarray], int]:
    """Path resolution of ZI/checkpoints.py:26-50; returns (state_dict as float32/int numpy, step).
    Unlike the reference (which prints and returns step 0) a missing checkpoint is an error here: rendering random
    weights by accident is never what the caller wants."""
    p = os.fspath(ckpt_dir_or_file)
    if step is not None:
        path = _checkpoint_path(p, step, prefix)
        if not os.path.exists(path):
            raise ValueError(f"Matching checkpoint not found: {path}")
    elif os.path.isdir(p):
        path = latest_checkpoint(p, prefix)
        if not path:
            raise FileNotFoundError(f"no checkpoint files in {p} with prefix {prefix}")
    elif os.path.exists(p):
        path = p
    else:
        raise FileNotFoundError(f"no checkpoint directory or file at {p}")
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if "state_dict" not in ckpt:
        raise ValueError(f"{path}: not a reference checkpoint (no 'state_dict' entry)")
    sd = {k: v.detach().cpu().numpy() for k, v in ckpt["state_dict"].items() if isinstance(v, torch.Tensor)}
    return sd, int(ckpt.get("step", 0))


def split_state_dict(sd: Dict[str, np.ndarray]) -> Tuple[Dict[str, np.ndarray], List[str]]:
    """(parameters of the fused path, keys that belong to branches outside it)."""
    keep, ignored = {}, []
    for k, v in sd.items():
        if k.startswith("module."):  # DDP-wrapped save
            k = k[len("module."):]
        if not re.match(r"^(nerf_mlp|prop_mlp_\d+)\.", k):
            ignored.append(k)  # obj_mlp_* / latent vectors (dynamic-object branch), GLO
        elif re.search(r"\.encoder\.(idx|grid_sizes|offsets)$", k):
            continue  # buffers re-derived from the config (Z/gridencoder/grid.py:137-142)
        else:
            keep[k] = v
    return keep, ignored


def _infer_mlp(sd, prefix: str, base: MLPConfig) -> MLPConfig:
    w0 = sd[f"{prefix}.density_layer.0.weight"]
    w2 = sd[f"{prefix}.density_layer.2.weight"]
    emb = sd[f"{prefix}.encoder.embeddings"]
    C = int(emb.shape[1])
    L = int(w0.shape[1]) // C
    if L * C != w0.shape[1]:
        raise ValueError(f"{prefix}: density_layer.0 takes {w0.shape[1]} features, not a multiple of level_dim {C}")
    cfg = dataclasses.replace(base, grid_level_dim=C,
                              grid_disired_resolution=base.grid_base_resolution * base.grid_level_interval ** (L - 1))
    # the hash-map size is whichever reproduces the table's row count (grid.py:122-135)
    for log2 in range(8, 29):
        cand = dataclasses.replace(cfg, grid_log2_hashmap_size=log2)
        if int(grid_layout(cand)[0][-1]) == emb.shape[0]:
            cfg = cand
            break
    else:
        raise ValueError(f"{prefix}: no log2_hashmap_size gives a {emb.shape[0]}-row table for {L} levels")
    cfg.disable_rgb = w2.shape[0] == 1 and f"{prefix}.rgb_layer.weight" not in sd
    if cfg.disable_rgb:
        return cfg
    cfg.bottleneck_width = int(w2.shape[0])
    depth = 0
    while f"{prefix}.lin_second_stage_{depth}.weight" in sd:
        depth += 1
    cfg.net_depth_viewdirs = depth
    cfg.net_width_viewdirs = int(sd[f"{prefix}.lin_second_stage_0.weight"].shape[0])
    cfg.deg_view = (int(sd[f"{prefix}.lin_second_stage_0.weight"].shape[1]) - cfg.bottleneck_width - 3) // 6
    cfg.use_semantic = f"{prefix}.sem_layer.0.weight" in sd or base.use_semantic
    cfg.no_sem_layer = f"{prefix}.sem_layer.0.weight" not in sd
    if not cfg.no_sem_layer:
        cfg.class_num = int(sd[f"{prefix}.sem_layer.2.weight"].shape[0])
    cfg.use_intensity = f"{prefix}.intensity_layer.0.weight" in sd
    # the skip connection after layer `skip_layer_dir` widens the NEXT layer's input (models.py:939-950)
    for i in range(1, depth):
        if sd[f"{prefix}.lin_second_stage_{i}.weight"].shape[1] > cfg.net_width_viewdirs:
            cfg.skip_layer_dir = i - 1
            break
    return cfg


def infer_model_config(sd: Dict[str, np.ndarray], base: Optional[ModelConfig] = None) -> ModelConfig:
    """Architecture from tensor shapes; sampling hyper-parameters (`num_prop_samples`, `num_nerf_samples`, dilation, ...) are
    not stored in a checkpoint, so they come from `base` (default: the shipped gin values, nerflidar_hip.config)."""
    base = base or ModelConfig()
    n_prop = 0
    while f"prop_mlp_{n_prop}.density_layer.0.weight" in sd:
        n_prop += 1
    if "nerf_mlp.density_layer.0.weight" not in sd:
        raise ValueError("state_dict has no nerf_mlp.* parameters")
    nerf = _infer_mlp(sd, "nerf_mlp", base.nerf_mlp)
    props = [_infer_mlp(sd, f"prop_mlp_{i}", base.prop_mlp) for i in range(n_prop)]
    prop_samples = tuple(base.num_prop_samples[:n_prop]) if len(base.num_prop_samples) >= n_prop else (64,) * n_prop
    cfg = dataclasses.replace(base.config, use_semantic=nerf.use_semantic, use_intensity=nerf.use_intensity, no_sem_layer=nerf.no_sem_layer)
    mc = ModelConfig(num_prop_samples=prop_samples, num_nerf_samples=base.num_nerf_samples, num_levels=n_prop + 1,
                     bg_intensity_range=base.bg_intensity_range, anneal_slope=base.anneal_slope, raydist_fn=base.raydist_fn,
                     dilation_multiplier=base.dilation_multiplier, dilation_bias=base.dilation_bias,
                     resample_padding=base.resample_padding, opaque_background=base.opaque_background,
                     power_lambda=base.power_lambda, std_scale=base.std_scale,
                     prop_desired_grid_size=tuple(p.grid_disired_resolution for p in props), config=cfg, nerf_mlp=nerf,
                     prop_mlp=(dataclasses.replace(props[0]) if props else base.prop_mlp))
    # every parameter the fused path will read must be present with the shape the inferred config implies
    for prefix, c in mlp_names(mc):
        for name, shape, _ in mlp_param_shapes(c):
            for suffix, want in ((".weight", shape), (".bias", (shape[0],))):
                key = f"{prefix}.{name}{suffix}"
                if key not in sd:
                    raise KeyError(f"checkpoint lacks {key}")
                if tuple(sd[key].shape) != tuple(want):
                    raise ValueError(f"{key}: shape {tuple(sd[key].shape)}, expected {tuple(want)}")
    return mc


def model_from_checkpoint(ckpt_dir_or_file, step=None, base: Optional[ModelConfig] = None, device="cuda:0", **model_kw):
    """Reference checkpoint -> ready `Model` (weights packed by `nlr_model_create`).  Returns (model, step, ignored_keys)."""
    from .models import Model
    sd, step = load_checkpoint(ckpt_dir_or_file, step)
    keep, ignored = split_state_dict(sd)
    mc = infer_model_config(keep, base)
    if any(k.startswith("obj_mlp") for k in ignored) and base is not None and base.config.instance_obj:
        raise NotImplementedError("checkpoint carries obj_mlp_* (dynamic-object branch, SURVEY 8f-1): outside the fused path")
    return Model(mc, keep, device=device, **model_kw), step, ignored


def dynamic_model_from_checkpoint(ckpt_dir_or_file, tracks, class_names, step=None, base: Optional[ModelConfig] = None, device="cuda:0",
                                  **model_kw):
    """Checkpoint of a `Config.instance_obj=True` run (the shipped gin) -> `nerflidar_hip.objects.DynamicModel`.
    The static field is imported as in `model_from_checkpoint`; `obj_mlp_<class id>.*` and `latent_vector_dict.obj_latent_<track>`
    (ZI/models.py:150-173) feed the object branch, whose latent size and hash-map size are read off the tensors.
    tracks [N_obj, T, 9] / class_names come from the dataset (`dataset.bboxes`), they are not stored in a checkpoint."""
    from .config import obj_mlp_config
    from .objects import DynamicModel, query_class
    sd, step = load_checkpoint(ckpt_dir_or_file, step)
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    keep, ignored = split_state_dict(sd)
    mc = infer_model_config(keep, base)
    lat_keys = [k for k in sd if k.startswith("latent_vector_dict.obj_latent_")]
    mc.config = dataclasses.replace(mc.config, instance_obj=True, latent_size=int(sd[lat_keys[0]].shape[0]) if lat_keys else 0)
    mc.__post_init__()
    cids = sorted({query_class(c) for c in class_names})
    log2 = None
    for cid in cids:
        key = f"obj_mlp_{cid}.encoder.embeddings"
        if key not in sd:
            raise KeyError(f"checkpoint lacks {key} (class id {cid} of the given tracks)")
        for cand in range(8, 29):
            if int(grid_layout(obj_mlp_config(cid, mc.config.latent_size, cand))[0][-1]) == sd[key].shape[0]:
                log2 = cand if log2 is None else log2
                if cand != log2:
                    raise ValueError("object networks with different hash-map sizes are not supported")
                break
        else:
            raise ValueError(f"{key}: {sd[key].shape[0]} rows match no log2_hashmap_size")
    used = dict(keep)
    used.update({k: v for k, v in sd.items() if k.startswith(("obj_mlp_", "latent_vector_dict."))
                 and not k.endswith((".encoder.idx", ".encoder.offsets", ".encoder.grid_sizes"))})
    left = [k for k in ignored if not k.startswith(("obj_mlp_", "latent_vector_dict."))]
    return DynamicModel(mc, used, tracks, class_names, device=device, obj_log2_hashmap=log2, **model_kw), step, left
