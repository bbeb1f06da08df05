"""Configuration for the render hot path, keeping the reference's gin names.

The reference configures the path by gin class-attribute injection on `Model`, `NerfMLP`,
`PropMLP` and the `Config` dataclass (ZI/models.py:30-58, 796-846, 1266-1277; ZI/configs.py:22-211;
Z/configs/nuscenes_single.gin).  gin itself is not installed here, so this module keeps the
same attribute names in plain dataclasses and ships a tiny reader for `Scope.name = value`
lines (`parse_gin_bindings`), enough for the shipped .gin files and `--gin_bindings` strings.
Only the fields the hot path reads are kept ("Config / flags").
"""
from __future__ import annotations

import ast
import dataclasses
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple


@dataclass
class MLPConfig:
    """Attributes of ZI/models.py:MLP (lines 796-846) read on the inference path."""
    bottleneck_width: int = 256
    net_depth_viewdirs: int = 2
    net_width_viewdirs: int = 256
    skip_layer_dir: int = 0
    num_rgb_channels: int = 3
    deg_view: int = 4
    density_bias: float = -1.0
    rgb_premultiplier: float = 1.0
    rgb_bias: float = 0.0
    rgb_padding: float = 0.001
    disable_density_normals: bool = True
    disable_rgb: bool = False
    warp_fn: str = "contract"
    grid_level_interval: int = 2
    grid_level_dim: int = 4
    grid_base_resolution: int = 16
    grid_disired_resolution: int = 8192  # (sic) reference spelling, ZI/models.py:829
    grid_log2_hashmap_size: int = 21
    class_num: int = 19
    use_semantic: bool = False
    use_intensity: bool = False
    no_sem_layer: bool = True
    re_weights: bool = True
    # ObjMLP-only attributes (ZI/models.py:837-846); the defaults are those of MLP, i.e. inert for NerfMLP / PropMLP
    fixed_semantic: bool = False
    class_type: int = 255
    latent_size: int = 0
    split_latent: bool = False

    @property
    def grid_num_levels(self) -> int:
        # ZI/models.py:867
        return int(math.log(self.grid_disired_resolution / self.grid_base_resolution)
                   / math.log(self.grid_level_interval)) + 1

    @property
    def dim_dir_enc(self) -> int:
        # coord.pos_enc(min_deg=0, max_deg=deg_view, append_identity=True): 3 + 2*3*deg
        return 3 + 6 * self.deg_view


@dataclass
class Config:
    """The subset of ZI/configs.py:Config the path reads."""
    use_semantic: bool = True
    use_intensity: bool = False
    no_sem_layer: bool = False
    analytic_gradient: bool = True
    zero_glo: bool = True
    instance_obj: bool = False
    sem_detach: bool = True
    vis_num_rays: int = 16
    hash_decay_mults: float = 0.0
    symmetrize: bool = False
    render_chunk_size: int = 16384
    sample_n_test: int = 7
    sample_m_test: int = 3
    latent_size: int = 128  # nuscenes_single.gin:18
    near: float = 0.1
    far: float = 10.0


@dataclass
class ModelConfig:
    """Attributes of ZI/models.py:Model (lines 31-58) plus the three MLP scopes."""
    num_prop_samples: Tuple[int, ...] = (64, 64)
    num_nerf_samples: int = 32
    num_levels: int = 3
    bg_intensity_range: Tuple[float, float] = (1.0, 1.0)
    anneal_slope: float = 10.0
    use_viewdirs: bool = True
    raydist_fn: str = "power_transformation"
    single_jitter: bool = True
    dilation_multiplier: float = 0.5
    dilation_bias: float = 0.0025
    resample_padding: float = 0.0
    opaque_background: bool = True
    power_lambda: float = -1.5
    std_scale: float = 0.35
    prop_desired_grid_size: Tuple[int, ...] = (512, 2048)
    near_anneal_rate: Optional[float] = None
    config: Config = field(default_factory=Config)
    nerf_mlp: MLPConfig = field(default_factory=MLPConfig)
    prop_mlp: MLPConfig = field(default_factory=lambda: MLPConfig(
        disable_rgb=True, grid_level_dim=1))

    def __post_init__(self):
        # Model.__init__ passes these from Config into NerfMLP (ZI/models.py:68-73).
        self.nerf_mlp.use_semantic = self.config.use_semantic
        self.nerf_mlp.use_intensity = self.config.use_intensity
        self.nerf_mlp.no_sem_layer = self.config.no_sem_layer

    def prop_cfg(self, i: int) -> MLPConfig:
        # ZI/models.py:80 -- one PropMLP per proposal level with its own desired resolution.
        return dataclasses.replace(self.prop_mlp, grid_disired_resolution=self.prop_desired_grid_size[i])

    def level_samples(self) -> List[int]:
        return [int(s) for s in self.num_prop_samples[: self.num_levels - 1]] + [int(self.num_nerf_samples)]


def obj_mlp_config(class_type: int = 255, latent_size: int = 128, log2_hashmap: int = 21, use_semantic: bool = True) -> MLPConfig:
    """ObjMLP as `Model.__init__` builds it in latent mode (ZI/models.py:125-142) under the shipped gin
    (nuscenes_single.gin:36-44): L = 7 levels x 2 features up to resolution 1024, bottleneck 64, view width 32, deg_view 2,
    no ray warp, no erf re-weighting, one-hot semantic of the track's class, latent split into shape / texture halves."""
    return MLPConfig(bottleneck_width=64, net_depth_viewdirs=2, net_width_viewdirs=32, deg_view=2, grid_level_dim=2,
                     grid_base_resolution=16, grid_disired_resolution=1024, grid_log2_hashmap_size=log2_hashmap,
                     disable_density_normals=True, disable_rgb=False, warp_fn=None, re_weights=False, fixed_semantic=True,
                     class_type=class_type, use_semantic=use_semantic, use_intensity=False, no_sem_layer=True,
                     latent_size=latent_size, split_latent=True)


# Named workloads of BASELINE.json / SURVEY section 8 ---------------------------------------
def workload(name: str, log2_hashmap: Optional[int] = None) -> ModelConfig:
    """REF = shipped gin; C1 = 4x128, 64 samples, single level; C2 = 8x256, (64,64,128)+intensity."""
    if name == "REF":
        mc = ModelConfig()
    elif name == "REFI":  # the shipped architecture with the intensity head switched on (Config.use_intensity, configs.py:164)
        mc = ModelConfig(config=Config(use_intensity=True))
    elif name == "C1":
        mc = ModelConfig(num_prop_samples=(), num_nerf_samples=64, num_levels=1,
                         nerf_mlp=MLPConfig(net_depth_viewdirs=4, net_width_viewdirs=128))
    elif name == "C2":
        mc = ModelConfig(num_prop_samples=(64, 64), num_nerf_samples=128,
                         config=Config(use_intensity=True),
                         nerf_mlp=MLPConfig(net_depth_viewdirs=8, net_width_viewdirs=256))
    elif name == "C2S":  # C2 as a single uniform level (SURVEY 8d "(i)")
        mc = ModelConfig(num_prop_samples=(), num_nerf_samples=128, num_levels=1,
                         config=Config(use_intensity=True),
                         nerf_mlp=MLPConfig(net_depth_viewdirs=8, net_width_viewdirs=256))
    elif name == "C3":  # camera, hierarchical (64 + 128)
        mc = ModelConfig(num_prop_samples=(64,), num_nerf_samples=128, num_levels=2,
                         prop_desired_grid_size=(512,),
                         nerf_mlp=MLPConfig(net_depth_viewdirs=8, net_width_viewdirs=256))
    # --- parity cases for the parameter space of ZI/models.py:MLP beyond the named configurations
    elif name == "P_NOSEM":  # no semantic / intensity head at all (Config.use_semantic = False)
        mc = ModelConfig(config=Config(use_semantic=False))
    elif name == "P_NSL":  # semantic logits = bottleneck[1:1+K] (Config.no_sem_layer = True, models.py:1133), + intensity
        mc = ModelConfig(config=Config(no_sem_layer=True, use_intensity=True))
    elif name == "P_NSL0":  # the same without the intensity head
        mc = ModelConfig(config=Config(no_sem_layer=True))
    elif name == "P_W128I":  # width 128 with both heads
        mc = ModelConfig(num_prop_samples=(), num_nerf_samples=64, num_levels=1, config=Config(use_intensity=True),
                         nerf_mlp=MLPConfig(net_depth_viewdirs=4, net_width_viewdirs=128))
    elif name == "P_F32":  # the GridEncoder's own default grid (16 levels x 2 features = 32, Z/gridencoder/grid.py:96-110) under the NerfMLP
        mc = ModelConfig(nerf_mlp=MLPConfig(grid_level_dim=2, grid_disired_resolution=16 * 2 ** 15))
    elif name == "P_F20":  # NerfMLP.grid_level_dim = 2 alone: 10 levels x 2 = 20 features
        mc = ModelConfig(config=Config(use_intensity=True), nerf_mlp=MLPConfig(grid_level_dim=2))
    elif name == "P_D3":  # odd view depths: 3 x 256 ...
        mc = ModelConfig(nerf_mlp=MLPConfig(net_depth_viewdirs=3, net_width_viewdirs=256))
    elif name == "P_D5":  # ... and 5 x 128
        mc = ModelConfig(num_prop_samples=(), num_nerf_samples=64, num_levels=1,
                         nerf_mlp=MLPConfig(net_depth_viewdirs=5, net_width_viewdirs=128))
    else:
        raise ValueError(f"unknown workload {name!r}")
    if log2_hashmap is not None:
        mc.nerf_mlp.grid_log2_hashmap_size = log2_hashmap
        mc.prop_mlp.grid_log2_hashmap_size = log2_hashmap
    return mc


# gin-lite -----------------------------------------------------------------------------------
_SCOPES = {"Model": None, "NerfMLP": "nerf_mlp", "PropMLP": "prop_mlp", "Config": "config"}


def parse_gin_bindings(text: str, mc: Optional[ModelConfig] = None) -> ModelConfig:
    """Apply `Scope.attr = value` lines (gin syntax subset) to a ModelConfig.

    Unknown scopes (ObjMLP, ...) and attributes the hot path does not read are ignored, as a
    gin file for the full reference program carries many such lines.
    """
    mc = mc or ModelConfig()
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].strip()
        if not line or "=" not in line:
            continue
        lhs, rhs = [s.strip() for s in line.split("=", 1)]
        if "." not in lhs:
            continue
        scope, attr = lhs.rsplit(".", 1)
        scope = scope.split("/")[-1]
        if scope not in _SCOPES:
            continue
        target = mc if _SCOPES[scope] is None else getattr(mc, _SCOPES[scope])
        if not hasattr(target, attr):
            continue
        try:
            val = ast.literal_eval(rhs)
        except (ValueError, SyntaxError):
            val = rhs.strip("'\"")
        if isinstance(getattr(target, attr), tuple) and isinstance(val, list):
            val = tuple(val)
        setattr(target, attr, val)
    mc.__post_init__()
    return mc
