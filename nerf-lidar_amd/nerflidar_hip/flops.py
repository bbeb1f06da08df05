"""Algorithmic work of the render path: GEMM MACs only, transcendental and
elementwise work excluded.  Used by bench.py for the roofline numerator."""
from __future__ import annotations

from .config import MLPConfig, ModelConfig
from .weights import mlp_param_shapes


def macs_per_sample(cfg: MLPConfig) -> int:
    return sum(o * i for _, (o, i), _ in mlp_param_shapes(cfg))


def flops_per_ray(mc: ModelConfig) -> int:
    s = mc.level_samples()
    total = 0
    for li in range(mc.num_levels):
        cfg = mc.prop_cfg(li) if li < mc.num_levels - 1 else mc.nerf_mlp
        total += s[li] * macs_per_sample(cfg)
    return 2 * total


def gather_bytes_per_ray(mc: ModelConfig, sample_n: int = 7, table_bytes: int = 4) -> int:
    """sum_l S_l * n * L_l * 8 corners * C_l * sizeof  (algorithmic, before any caching)."""
    s = mc.level_samples()
    total = 0
    for li in range(mc.num_levels):
        cfg = mc.prop_cfg(li) if li < mc.num_levels - 1 else mc.nerf_mlp
        total += s[li] * sample_n * cfg.grid_num_levels * 8 * cfg.grid_level_dim * table_bytes
    return total
