"""Parameter tables for the render path, keyed exactly like the reference's `state_dict`.

Key names follow `Model.state_dict()` of ZI/models.py (`nerf_mlp.encoder.embeddings`,
`nerf_mlp.density_layer.0.weight`, `nerf_mlp.lin_second_stage_3.bias`, `prop_mlp_0.encoder...`),
so a released checkpoint's `state_dict` can be passed to `Model.load_state_dict` unchanged
(ZI/checkpoints.py:26-55).  Values are float32 numpy arrays.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import synth
from .config import MLPConfig, ModelConfig


def level_table(num_levels: int, base_resolution: int, log2_hashmap_size: int, per_level_scale: float = 2.0,
                desired_resolution=None, input_dim: int = 3, align_corners: bool = False):
    """Rows of the multi-resolution table, level by level (Z/gridencoder/grid.py:105-106,122-142).

    Level l has G_l = ceil(H * s^l) (+1 unless align_corners) grid points per axis and stores min(2^log2_hashmap_size, G_l^D)
    rows rounded up to a multiple of 8; `desired_resolution` (if given) fixes s so that the finest level reaches it.
    Returns (offsets int32 [L+1], grid_sizes int32 [L], per_level_scale float).
    """
    if desired_resolution is not None:
        per_level_scale = float(np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))) if num_levels > 1 else 1.0
    grow = 0 if align_corners else 1
    sizes = [int(np.ceil(base_resolution * per_level_scale ** l)) + grow for l in range(num_levels)]
    cap = 2 ** log2_hashmap_size
    rows = [-(-min(cap, g ** input_dim) // 8) * 8 for g in sizes]
    offsets = np.concatenate([[0], np.cumsum(rows, dtype=np.int64)])
    return offsets.astype(np.int32), np.asarray(sizes, np.int32), float(per_level_scale)


def grid_layout(cfg: MLPConfig, input_dim: int = 3, align_corners: bool = False):
    """`level_table` for an MLP's gin-configured grid (ZI/models.py:867-880)."""
    return level_table(cfg.grid_num_levels, cfg.grid_base_resolution, cfg.grid_log2_hashmap_size, 2.0,
                       cfg.grid_disired_resolution, input_dim, align_corners)


def mlp_param_shapes(cfg: MLPConfig) -> List[Tuple[str, Tuple[int, int], bool]]:
    """(name, (out,in), kaiming) for every Linear of ZI/models.py:MLP.__init__ on the path."""
    feat = cfg.grid_num_levels * cfg.grid_level_dim
    if cfg.latent_size > 0:  # models.py:881-885: the (shape half of the) latent code rides beside the grid features
        feat += cfg.latent_size // 2 if cfg.split_latent else cfg.latent_size
    out = [("density_layer.0", (64, feat), False),
           ("density_layer.2", (1 if cfg.disable_rgb else cfg.bottleneck_width, 64), False)]
    if cfg.disable_rgb:
        return out
    in_rgb = cfg.bottleneck_width + cfg.dim_dir_enc  # models.py:920-926
    if cfg.split_latent:
        in_rgb += cfg.latent_size // 2  # texture half (models.py:924-925)
    last = in_rgb
    for i in range(cfg.net_depth_viewdirs):  # models.py:939-950
        out.append((f"lin_second_stage_{i}", (cfg.net_width_viewdirs, last), True))
        last = cfg.net_width_viewdirs
        if i == cfg.skip_layer_dir:
            last += in_rgb
    out.append(("rgb_layer", (cfg.num_rgb_channels, last), False))
    if not cfg.no_sem_layer and not cfg.fixed_semantic:  # models.py:954-957 (built whether or not use_semantic reads it)
        out += [("sem_layer.0", (64, cfg.bottleneck_width), False), ("sem_layer.2", (cfg.class_num, 64), False)]
    if cfg.use_intensity:  # models.py:958-961
        out += [("intensity_layer.0", (64, cfg.bottleneck_width), False), ("intensity_layer.2", (1, 64), False)]
    return out


def mlp_names(mc: ModelConfig) -> List[Tuple[str, MLPConfig]]:
    names = [(f"prop_mlp_{i}", mc.prop_cfg(i)) for i in range(mc.num_levels - 1)]
    names.append(("nerf_mlp", mc.nerf_mlp))
    return names


def synth_state_dict(mc: ModelConfig, seed: int = 0, table_std: float = 1e-4,
                     trained_like: bool = False) -> Dict[str, np.ndarray]:
    """Seeded random-init parameters with the reference's init ranges.

    table_std=1e-4 is the reference init (grid.py:101): with it every ray composites as uniform
    fog, which exercises little.  trained_like=True is SURVEY 8d's second weight set: tables
    U(-1,1) * 2^(-level/2) (fine levels carry less amplitude, as the reference's hash-decay
    regulariser enforces, ZI/models.py:203-223; with full-amplitude white noise at resolution 8192 a
    1-ulp change of a coordinate moves a feature by 5e-4, which measures libm differences, not the
    renderer) and a few heads rescaled so that the random scene has empty space, opaque surfaces
    (density up to a few hundred), rays that reach the opaque background, and >10 distinct
    semantic labels per sweep:
      density_layer.2 (NerfMLP, all rows) x8, row 0 (raw density) x1500 in every MLP with its
      bias shifted by -40, sem_layer.2.weight x8.
    """
    if trained_like:
        table_std = 1.0
    sd: Dict[str, np.ndarray] = {}
    for prefix, cfg in mlp_names(mc):
        offsets, sizes, _ = grid_layout(cfg)
        table = synth.table_init(seed, f"{prefix}.encoder.embeddings", int(offsets[-1]), cfg.grid_level_dim, table_std)
        if trained_like:
            for l in range(len(offsets) - 1):
                table[offsets[l]:offsets[l + 1]] *= np.float32(2.0 ** (-0.5 * l))
        sd[f"{prefix}.encoder.embeddings"] = table
        sd[f"{prefix}.encoder.offsets"] = offsets
        sd[f"{prefix}.encoder.grid_sizes"] = sizes
        for name, (o, i), kaiming in mlp_param_shapes(cfg):
            w, b = synth.linear_init(seed, f"{prefix}.{name}", o, i, kaiming)
            if trained_like and name == "density_layer.2":
                if not cfg.disable_rgb:
                    w = w * np.float32(8.0)
                    w[0] *= np.float32(1500.0 / 8.0)
                else:
                    w[0] *= np.float32(1500.0)
                b[0] += np.float32(-40.0)
            if trained_like and name == "sem_layer.2":
                w = w * np.float32(8.0)
            sd[f"{prefix}.{name}.weight"] = np.ascontiguousarray(w, np.float32)
            sd[f"{prefix}.{name}.bias"] = np.ascontiguousarray(b, np.float32)
    return sd


def synth_object_state_dict(obj_cfgs: Dict[int, MLPConfig], n_tracks: int, seed: int = 0, trained_like: bool = True) -> Dict[str, np.ndarray]:
    """Seeded parameters of the dynamic-object branch with the reference's key names (ZI/models.py:150-173): one
    `obj_mlp_<class_id>.*` per class (latent mode) and one `latent_vector_dict.obj_latent_<track>` per track
    (train_utils.py:459-471, normal init).  trained_like scales tables and the density row so that boxes contain both
    empty and opaque space."""
    sd: Dict[str, np.ndarray] = {}
    for cid, cfg in obj_cfgs.items():
        prefix = f"obj_mlp_{cid}"
        offsets, sizes, _ = grid_layout(cfg)
        table = synth.table_init(seed, f"{prefix}.encoder.embeddings", int(offsets[-1]), cfg.grid_level_dim, 1.0 if trained_like else 1e-4)
        if trained_like:
            for l in range(len(offsets) - 1):
                table[offsets[l]:offsets[l + 1]] *= np.float32(2.0 ** (-0.5 * l))
        sd[f"{prefix}.encoder.embeddings"] = table
        sd[f"{prefix}.encoder.offsets"] = offsets
        sd[f"{prefix}.encoder.grid_sizes"] = sizes
        for name, (o, i), kaiming in mlp_param_shapes(cfg):
            w, b = synth.linear_init(seed, f"{prefix}.{name}", o, i, kaiming)
            if trained_like and name == "density_layer.2":
                w[0] *= np.float32(300.0)
                b[0] += np.float32(-6.0)
            sd[f"{prefix}.{name}.weight"] = np.ascontiguousarray(w, np.float32)
            sd[f"{prefix}.{name}.bias"] = np.ascontiguousarray(b, np.float32)
    lat = max((c.latent_size for c in obj_cfgs.values()), default=0)
    for t in range(n_tracks):
        if lat:
            u = synth.uniform(seed, 7000 + t, (2, lat), 1e-7, 1.0).astype(np.float64)
            sd[f"latent_vector_dict.obj_latent_{t}"] = (np.sqrt(-2 * np.log(u[0])) * np.cos(2 * np.pi * u[1])).astype(np.float32)
    return sd


_PRIMES = (1, 2654435761, 805459861)  # gridencoder.cu:54


def inflate_hashmaps(sd: Dict[str, np.ndarray], mc: ModelConfig, log2_hashmap: int):
    """The SAME field on larger hash maps: (state_dict, ModelConfig) whose grids have 2^log2_hashmap rows per hashed level and evaluate,
    bit for bit, the function the given (smaller-map) parameters evaluate.

    A level that is hashed in both layouts reads row `hash(p) mod T` (gridencoder.cu:66-84); with T_small | T_big,
    (h mod T_big) mod T_small = h mod T_small, so tiling the small level T_big / T_small times reproduces every look-up.  A level that
    is hashed in the small layout but fits the big one densely gets row `x + G y + G^2 z` filled with the small level's row
    `hash(x, y, z) mod T_small`.  Dense-in-both levels are copied.  Purpose: a checkpoint trained with small maps (a few MB, committable)
    becomes a model with the memory footprint and the access pattern of the full-size configuration (229 MiB NerfMLP table) without
    changing one rendered value - the benchmark's trained-scene workload."""
    import dataclasses
    big = dataclasses.replace(mc, nerf_mlp=dataclasses.replace(mc.nerf_mlp, grid_log2_hashmap_size=log2_hashmap),
                              prop_mlp=dataclasses.replace(mc.prop_mlp, grid_log2_hashmap_size=log2_hashmap), config=dataclasses.replace(mc.config))
    out = {k: v for k, v in sd.items() if ".encoder." not in k}
    for (prefix, cs), (_, cb) in zip(mlp_names(mc), mlp_names(big)):
        if cb.grid_log2_hashmap_size < cs.grid_log2_hashmap_size:
            raise ValueError("inflate_hashmaps only grows the maps")
        off_s, sizes, _ = grid_layout(cs)
        off_b, sizes_b, _ = grid_layout(cb)
        small = np.asarray(sd[f"{prefix}.encoder.embeddings"])
        table = np.zeros((int(off_b[-1]), small.shape[1]), small.dtype)
        for l, G in enumerate(int(g) for g in sizes):
            ts, tb = int(off_s[l + 1] - off_s[l]), int(off_b[l + 1] - off_b[l])
            src, dst = small[off_s[l]:off_s[l + 1]], table[off_b[l]:off_b[l + 1]]
            dense_s, dense_b = G ** 3 <= ts, G ** 3 <= tb
            if dense_s:                      # dense in both (the big map is at least as large)
                dst[:ts] = src
            elif not dense_b:                # hashed in both
                if tb % ts:
                    raise ValueError(f"{prefix} level {l}: {tb} rows are not a multiple of {ts}")
                dst[:] = np.tile(src, (tb // ts, 1))
            else:                            # hashed -> dense
                ax = np.arange(G, dtype=np.uint64)
                h = ((ax * np.uint64(_PRIMES[0]))[None, None, :] & np.uint64(0xFFFFFFFF)) ^ \
                    ((ax * np.uint64(_PRIMES[1]))[None, :, None] & np.uint64(0xFFFFFFFF)) ^ \
                    ((ax * np.uint64(_PRIMES[2]))[:, None, None] & np.uint64(0xFFFFFFFF))        # [z, y, x], x fastest = dense row order
                dst[:G ** 3] = src[(h % np.uint64(ts)).reshape(-1).astype(np.int64)]
        out[f"{prefix}.encoder.embeddings"] = table
        out[f"{prefix}.encoder.offsets"], out[f"{prefix}.encoder.grid_sizes"] = off_b, sizes_b
    return out, big
