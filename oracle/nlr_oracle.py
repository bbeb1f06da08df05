"""oracle/nlr_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU (PyTorch fp32) restatement of the reference's render hot path, function by function.  It is
the checker for the HIP path and the `cpu_baseline` leg of bench.py; the product never imports
it (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may).

Pinning: every function here is checked in tests/test_oracle_golden.py against fixtures under
tests/golden/ that were produced by importing the reference's own Python
(/root/reference/NeRF_LiDAR/zipnerf/internal/{math,stepfun,render,coord,models}.py) in the build
container with tests/golden/make_golden.py.  The only piece without a runnable reference is the
CUDA hash-grid kernel; its restatement is oracle/grid_oracle.c (see the header there) with a
numpy twin below (`grid_encode_numpy`) used to cross-check the C.

All citations are relative to /root/reference/NeRF_LiDAR/zipnerf/ (ZI = internal/).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

EPS = float(torch.finfo(torch.float32).eps)
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


# --------------------------------------------------------------------------------------------
# a-7  grid encoder (gridencoder/src/gridencoder.cu:50-245, gridencoder/grid.py:122-174)
# --------------------------------------------------------------------------------------------
def _load_c():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libgrid_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.nlr_oracle_grid_forward.restype = None
        lib.nlr_oracle_grid_backward.restype = None
        lib.nlr_oracle_level_scale.restype = None
        lib.nlr_oracle_grid_tv.restype = None
        _LIB = lib
    return _LIB


def level_scale(L: int, S: float, H: int):
    """Per-level (scale, resolution) exactly as the host launcher computes them (cu:138-139)."""
    lib = _load_c()
    sc = np.zeros(L, np.float32)
    rs = np.zeros(L, np.uint32)
    lib.nlr_oracle_level_scale(ctypes.c_uint32(L), ctypes.c_float(S), ctypes.c_uint32(H),
                               sc.ctypes.data_as(ctypes.c_void_p), rs.ctypes.data_as(ctypes.c_void_p))
    return sc, rs


def grid_encode_c(x01: np.ndarray, table: np.ndarray, offsets: np.ndarray, S: float, H: int,
                  gridtype: int = 0, align_corners: bool = False, interp: int = 0,
                  want_dy_dx: bool = False):
    """x01 [B,D] f32 in [0,1] -> outputs [L,B,C] f32 (level-major like the kernel), dy_dx or None."""
    lib = _load_c()
    x01 = np.ascontiguousarray(x01, np.float32)
    table = np.ascontiguousarray(table, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    B, D = x01.shape
    C = table.shape[1]
    L = offsets.shape[0] - 1
    out = np.empty((L, B, C), np.float32)
    dy = np.empty((B, L * D * C), np.float32) if want_dy_dx else None
    lib.nlr_oracle_grid_forward(
        x01.ctypes.data_as(ctypes.c_void_p), table.ctypes.data_as(ctypes.c_void_p),
        offsets.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_uint32(B), ctypes.c_uint32(D), ctypes.c_uint32(C), ctypes.c_uint32(L),
        ctypes.c_float(S), ctypes.c_uint32(H),
        dy.ctypes.data_as(ctypes.c_void_p) if want_dy_dx else ctypes.c_void_p(0),
        ctypes.c_uint32(gridtype), ctypes.c_int(int(align_corners)), ctypes.c_uint32(interp))
    return out, dy


def grid_backward_c(grad: np.ndarray, x01: np.ndarray, offsets: np.ndarray, n_entries: int, C: int,
                    S: float, H: int, dy_dx: Optional[np.ndarray] = None, gridtype: int = 0,
                    align_corners: bool = False, interp: int = 0):
    """grad [L,B,C] -> (grad_table [sO,C], grad_inputs [B,D] or None)  (cu:248-369)."""
    lib = _load_c()
    grad = np.ascontiguousarray(grad, np.float32)
    x01 = np.ascontiguousarray(x01, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    B, D = x01.shape
    L = offsets.shape[0] - 1
    gt = np.zeros((n_entries, C), np.float32)
    gi = np.zeros((B, D), np.float32) if dy_dx is not None else None
    lib.nlr_oracle_grid_backward(
        grad.ctypes.data_as(ctypes.c_void_p), x01.ctypes.data_as(ctypes.c_void_p),
        offsets.ctypes.data_as(ctypes.c_void_p), gt.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_uint32(B), ctypes.c_uint32(D), ctypes.c_uint32(C), ctypes.c_uint32(L),
        ctypes.c_float(S), ctypes.c_uint32(H),
        dy_dx.ctypes.data_as(ctypes.c_void_p) if dy_dx is not None else ctypes.c_void_p(0),
        gi.ctypes.data_as(ctypes.c_void_p) if gi is not None else ctypes.c_void_p(0),
        ctypes.c_uint32(gridtype), ctypes.c_int(int(align_corners)), ctypes.c_uint32(interp))
    return gt, gi


def grid_tv_c(x01: np.ndarray, table: np.ndarray, grad: np.ndarray, offsets: np.ndarray, weight: float, S: float, H: int,
              gridtype: int = 0, align_corners: bool = False) -> np.ndarray:
    """grad_total_variation (cu:506-601): returns grad + the TV gradient at the cells of x01 [B,D] (in [0,1])."""
    lib = _load_c()
    x01 = np.ascontiguousarray(x01, np.float32)
    table = np.ascontiguousarray(table, np.float32)
    out = np.array(grad, np.float32, copy=True, order="C")
    offsets = np.ascontiguousarray(offsets, np.int32)
    B, D = x01.shape
    lib.nlr_oracle_grid_tv(x01.ctypes.data_as(ctypes.c_void_p), table.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                           offsets.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(weight), ctypes.c_uint32(B), ctypes.c_uint32(D),
                           ctypes.c_uint32(table.shape[1]), ctypes.c_uint32(offsets.shape[0] - 1), ctypes.c_float(S), ctypes.c_uint32(H),
                           ctypes.c_uint32(gridtype), ctypes.c_int(int(align_corners)))
    return out


def grid_encode_numpy(x01, table, offsets, S, H, gridtype=0, align_corners=False):
    """Vectorised numpy twin of kernel_grid (linear interp, D=3) used to cross-check the C.
    Uses float64 FMA emulation for `x*scale+0.5` and the accumulation, so it agrees with the C to
    ~1 ulp rather than bit for bit."""
    x01 = np.asarray(x01, np.float32)
    B, D = x01.shape
    assert D == 3
    C = table.shape[1]
    L = offsets.shape[0] - 1
    out = np.zeros((L, B, C), np.float32)
    oob = ((x01 < 0) | (x01 > 1)).any(-1)
    primes = np.array([1, 2654435761, 805459861], np.uint64)
    for l in range(L):
        scale = np.float32(np.exp2(np.float32(l * S)) * np.float32(H) - np.float32(1.0))
        res = np.uint64(int(np.ceil(scale)) + 1)
        hs = np.uint64(int(offsets[l + 1] - offsets[l]))
        pos = (x01.astype(np.float64) * np.float64(scale) + (0.0 if align_corners else 0.5)).astype(np.float32)
        pg = np.floor(pos).astype(np.int64)
        fr = (pos - pg.astype(np.float32)).astype(np.float32)
        acc = np.zeros((B, C), np.float64)
        for idx in range(8):
            w = np.ones(B, np.float32)
            pl = np.empty((B, 3), np.uint64)
            for d in range(3):
                if (idx >> d) & 1:
                    w = (w * fr[:, d]).astype(np.float32)
                    pl[:, d] = (pg[:, d] + 1).astype(np.uint64)
                else:
                    w = (w * (np.float32(1) - fr[:, d])).astype(np.float32)
                    pl[:, d] = pg[:, d].astype(np.uint64)
            stride = np.uint64(1)
            index = np.zeros(B, np.uint64)
            hashed = False
            step = res if align_corners else res + np.uint64(1)
            for d in range(3):
                if stride <= hs:
                    index = (index + pl[:, d] * stride) & np.uint64(0xFFFFFFFF)
                    stride = (stride * step) & np.uint64(0xFFFFFFFF)
            if gridtype == 0 and stride > hs:
                index = np.zeros(B, np.uint64)
                for d in range(3):
                    index ^= (pl[:, d] * primes[d]) & np.uint64(0xFFFFFFFF)
            index = (index % hs).astype(np.int64) + int(offsets[l])
            g = table[np.where(oob, 0, index)]
            acc = (acc + w[:, None].astype(np.float64) * g.astype(np.float64)).astype(np.float32).astype(np.float64)
        out[l] = np.where(oob[:, None], 0, acc.astype(np.float32))
    return out


class GridEncoder:
    """Stand-in for gridencoder.GridEncoder (grid.py:96-174) holding a numpy table."""

    def __init__(self, table: np.ndarray, offsets: np.ndarray, grid_sizes: np.ndarray,
                 per_level_scale: float, base_resolution: int, gridtype: int = 0,
                 align_corners: bool = False, interp: int = 0):
        self.table = np.ascontiguousarray(table, np.float32)
        self.offsets = np.ascontiguousarray(offsets, np.int32)
        self.grid_sizes = torch.from_numpy(np.asarray(grid_sizes, np.int32))
        self.num_levels = len(offsets) - 1
        self.level_dim = self.table.shape[1]
        self.output_dim = self.num_levels * self.level_dim
        self.S = float(np.log2(per_level_scale))  # grid.py:38
        self.H = int(base_resolution)
        self.gridtype, self.align_corners, self.interp = gridtype, align_corners, interp

    def __call__(self, inputs: torch.Tensor, bound: float = 1) -> torch.Tensor:
        x01 = (inputs + bound) / (2 * bound)  # grid.py:162
        prefix = list(x01.shape[:-1])
        flat = x01.reshape(-1, 3).contiguous().numpy()
        out, _ = grid_encode_c(flat, self.table, self.offsets, self.S, self.H, self.gridtype,
                               self.align_corners, self.interp)
        # grid.py:57 -- [L,B,C] -> [B, L*C]
        out = torch.from_numpy(out).permute(1, 0, 2).reshape(flat.shape[0], self.output_dim)
        return out.reshape(prefix + [self.output_dim])


# --------------------------------------------------------------------------------------------
# a-2  ray warp (ZI/coord.py:103-162)
# --------------------------------------------------------------------------------------------
def power_transformation(x, lam):
    lam_1 = abs(lam - 1)
    return lam_1 / lam * ((x / lam_1 + 1) ** lam - 1)


def inv_power_transformation(x, lam):
    lam_1 = abs(lam - 1)
    return ((x * lam / lam_1 + 1 + EPS) ** (1 / lam) - 1) * lam_1


def construct_ray_warps(t_near, t_far, lam):
    """'power_transformation' branch only (coord.py:144-146, 159-162); the gin value used."""
    fwd = lambda x: power_transformation(x * 2, lam)
    inv = lambda y: inv_power_transformation(y, lam) / 2
    s_near, s_far = fwd(t_near), fwd(t_far)
    t_to_s = lambda t: (fwd(t) - s_near) / (s_far - s_near)
    s_to_t = lambda s: inv(s * s_far + (1 - s) * s_near)
    return t_to_s, s_to_t


# --------------------------------------------------------------------------------------------
# a-3 / a-4  step-function algebra (ZI/stepfun.py, ZI/math.py:89-108)
# --------------------------------------------------------------------------------------------
def sorted_interp(x, xp, fp):
    """ZI/math.py:89-108 in index form.  For sorted xp/fp the masked max/min of the reference
    selects xp[i0], xp[i1] with i0 = last j: x >= xp_j (else 0), i1 = first j: x < xp_j (else
    last); searchsorted(right=True) gives exactly those."""
    n = xp.shape[-1]
    hi = torch.searchsorted(xp.contiguous(), x.contiguous(), right=True)  # count of xp <= x
    i0 = (hi - 1).clamp(0, n - 1)
    i1 = hi.clamp(0, n - 1)
    xp0, xp1 = torch.gather(xp, -1, i0), torch.gather(xp, -1, i1)
    fp0, fp1 = torch.gather(fp, -1, i0), torch.gather(fp, -1, i1)
    offset = torch.clip(torch.nan_to_num((x - xp0) / (xp1 - xp0), 0), 0, 1)
    return fp0 + offset * (fp1 - fp0)


def max_dilate(t, w, dilation, domain):
    """ZI/stepfun.py:75-88 (dense mask form kept: it defines the tie behaviour)."""
    t0 = t[..., :-1] - dilation
    t1 = t[..., 1:] + dilation
    t_dilate, _ = torch.sort(torch.cat([t, t0, t1], dim=-1), dim=-1)
    t_dilate = torch.clip(t_dilate, *domain)
    inside = (t0[..., None, :] <= t_dilate[..., None]) & (t1[..., None, :] > t_dilate[..., None])
    w_dilate = torch.where(inside, w[..., None, :], torch.zeros_like(w[..., None, :])).max(dim=-1).values[..., :-1]
    return t_dilate, w_dilate


def max_dilate_weights(t, w, dilation, domain, renormalize=True):
    """ZI/stepfun.py:91-105 with weight_to_pdf :64-67 and pdf_to_weight :70-72."""
    p = w / (t[..., 1:] - t[..., :-1]).clamp_min(EPS)
    t_dilate, p_dilate = max_dilate(t, p, dilation, domain)
    w_dilate = p_dilate * (t_dilate[..., 1:] - t_dilate[..., :-1])
    if renormalize:
        w_dilate = w_dilate / torch.sum(w_dilate, dim=-1, keepdim=True).clamp_min(EPS)
    return t_dilate, w_dilate


def integrate_weights(w):
    """ZI/stepfun.py:108-128."""
    cw = torch.cumsum(w[..., :-1], dim=-1).clamp_max(1)
    shape = cw.shape[:-1] + (1,)
    return torch.cat([torch.zeros(shape), cw, torch.ones(shape)], dim=-1)


def sample_u(num_samples: int, rand_u: Optional[torch.Tensor] = None):
    """ZI/stepfun.py:203-216 (deterministic_center=True, single_jitter=True).  rand_u: [N,1]
    uniform draws standing in for torch.rand, or None for the deterministic linspace."""
    if rand_u is None:
        pad = 1 / (2 * num_samples)
        return torch.linspace(pad, 1. - pad - EPS, num_samples)
    u_max = EPS + (1 - EPS) / num_samples
    max_jitter = (1 - u_max) / (num_samples - 1) - EPS
    return torch.linspace(0, 1 - u_max, num_samples) + rand_u * max_jitter


def sample_intervals(t, w_logits, num_samples, domain, rand_u=None):
    """ZI/stepfun.py:251-294 -> sample :175-218 -> invert_cdf :154-161."""
    u = sample_u(num_samples, rand_u)
    u = torch.broadcast_to(u, t.shape[:-1] + (num_samples,))
    w = torch.softmax(w_logits, dim=-1)
    cw = integrate_weights(w)
    centers = sorted_interp(u, cw, t)
    mid = (centers[..., 1:] + centers[..., :-1]) / 2
    first = (2 * centers[..., :1] - mid[..., :1]).clamp_min(domain[0])
    last = (2 * centers[..., -1:] - mid[..., -1:]).clamp_max(domain[1])
    return torch.cat([first, mid, last], dim=-1)


def weighted_percentile(t, w, ps):
    """ZI/stepfun.py:329-339."""
    cw = integrate_weights(w)
    x = torch.broadcast_to(torch.tensor(ps, dtype=torch.float32) / 100, cw.shape[:-1] + (len(ps),))
    return sorted_interp(x, cw, t)


# --------------------------------------------------------------------------------------------
# a-5  cast_rays (ZI/render.py:129-168)
# --------------------------------------------------------------------------------------------
def cast_rays(tdist, origins, directions, radii, base_x, base_y, n=7, m=3, std_scale=0.35, rand_deg=None):
    t0, t1 = tdist[..., :-1], tdist[..., 1:]
    j = torch.arange(n)
    t = t0[..., None] + (t1[..., None] - t0[..., None]) * (j + 0.5) / n
    deg = torch.broadcast_to(2 * torch.pi * m * j / n, t.shape)
    if rand_deg is not None:
        deg = deg + rand_deg * torch.pi * 2
    means = torch.stack([radii[..., None] * t * torch.cos(deg) / 2,
                         radii[..., None] * t * torch.sin(deg) / 2, t], dim=-1)
    stds = std_scale * radii[..., None] * t
    basis = torch.stack([base_x, base_y, directions], dim=-1)
    means = torch.matmul(means, basis[..., None, :, :].transpose(-1, -2))
    means = means + origins[..., None, None, :]
    return means, stds


# --------------------------------------------------------------------------------------------
# a-6  contraction (ZI/coord.py:51-63, 67-100)
# --------------------------------------------------------------------------------------------
def contract_mean_std(x, std):
    x_mag_sq = torch.sum(x ** 2, dim=-1, keepdim=True).clamp_min(EPS)
    x_mag_sqrt = torch.sqrt(x_mag_sq)
    mask = x_mag_sq <= 1
    z = torch.where(mask, x, ((2 * torch.sqrt(x_mag_sq) - 1) / x_mag_sq) * x)
    det = ((1 / x_mag_sq) * ((2 / x_mag_sqrt - 1 / x_mag_sq) ** 2))[..., 0]
    std = torch.where(mask[..., 0], std, (det ** (1 / x.shape[-1])) * std)
    return z, std


# --------------------------------------------------------------------------------------------
# a-10  pos_enc (ZI/coord.py:199-210)
# --------------------------------------------------------------------------------------------
def pos_enc(x, min_deg, max_deg):
    scales = 2 ** torch.arange(min_deg, max_deg)
    shape = x.shape[:-1] + (-1,)
    scaled_x = (x[..., None, :] * scales[:, None]).reshape(*shape)
    four_feat = torch.sin(torch.cat([scaled_x, scaled_x + 0.5 * torch.pi], dim=-1))
    return torch.cat([x, four_feat], dim=-1)


# --------------------------------------------------------------------------------------------
# a-8 .. a-12  MLP.forward at inference (ZI/models.py:965-1004, 1070-1263)
# --------------------------------------------------------------------------------------------
def _lin(sd, name, x):
    return torch.nn.functional.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def encode_features(enc: GridEncoder, means, stds, re_weights=True):
    """models.py:968-979: contract, /2, grid encode, erf re-weight, mean over multisamples."""
    pre = means.shape[:-1]
    m, s = contract_mean_std(means.reshape(-1, 3), stds.reshape(-1))
    m, s = m.reshape(*pre, 3), s.reshape(*pre)
    bound = 2
    m, s = m / bound, s / bound
    feats = enc(m, bound=1).unflatten(-1, (enc.num_levels, -1))
    if re_weights:
        w = torch.erf(1 / torch.clamp(torch.sqrt(8 * s[..., None] ** 2 * enc.grid_sizes ** 2), min=1e-10))
        feats = (feats * w[..., None]).mean(dim=-3).flatten(-2, -1)
    else:
        feats = feats.flatten(-2, -1)
    return feats


def mlp_forward(sd: Dict[str, torch.Tensor], prefix: str, cfg, enc: GridEncoder, means, stds, viewdirs):
    """Returns dict(density, rgb, semantic, intensity, features, bottleneck) for one level."""
    F = torch.nn.functional
    feats = encode_features(enc, means, stds, cfg.re_weights)
    x = _lin(sd, f"{prefix}.density_layer.2", F.relu(_lin(sd, f"{prefix}.density_layer.0", feats)))
    raw_density = x[..., 0]
    density = F.softplus(raw_density + cfg.density_bias)
    out = dict(density=density, features=feats, bottleneck=x, semantic=None, intensity=None)
    if cfg.disable_rgb:
        out["rgb"] = torch.zeros(density.shape + (3,))
        return out
    if cfg.use_semantic:
        if cfg.no_sem_layer:
            sem = x[..., 1:(1 + cfg.class_num)]
        else:
            sem = _lin(sd, f"{prefix}.sem_layer.2", F.relu(_lin(sd, f"{prefix}.sem_layer.0", x)))
        out["semantic"] = torch.softmax(sem, -1)
    if cfg.use_intensity:
        out["intensity"] = _lin(sd, f"{prefix}.intensity_layer.2", F.relu(_lin(sd, f"{prefix}.intensity_layer.0", x)))
    dir_enc = pos_enc(viewdirs, 0, cfg.deg_view)
    dir_enc = torch.broadcast_to(dir_enc[..., None, :], x.shape[:-1] + (dir_enc.shape[-1],))
    h = torch.cat([x, dir_enc], dim=-1)
    inputs = h
    for i in range(cfg.net_depth_viewdirs):
        h = F.relu(_lin(sd, f"{prefix}.lin_second_stage_{i}", h))
        if i == cfg.skip_layer_dir:
            h = torch.cat([h, inputs], dim=-1)
    rgb = torch.sigmoid(cfg.rgb_premultiplier * _lin(sd, f"{prefix}.rgb_layer", h) + cfg.rgb_bias)
    out["rgb"] = rgb * (1 + 2 * cfg.rgb_padding) - cfg.rgb_padding
    return out


# --------------------------------------------------------------------------------------------
# a-13 / a-14  compositing (ZI/render.py:170-284)
# --------------------------------------------------------------------------------------------
def compute_alpha_weights(density, tdist, dirs, opaque_background):
    t_delta = tdist[..., 1:] - tdist[..., :-1]
    delta = t_delta * torch.norm(dirs[..., None, :], dim=-1)
    dd = density * delta
    if opaque_background:
        dd = torch.cat([dd[..., :-1], torch.full_like(dd[..., -1:], torch.inf)], dim=-1)
    alpha = 1 - torch.exp(-dd)
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], dim=-1)], dim=-1))
    return alpha * trans


def volumetric_rendering(rgbs, weights, tdist, bg_rgbs, t_far, compute_extras, semantic=None, intensity=None):
    r = {}
    acc = weights.sum(dim=-1)
    bg_w = (1 - acc[..., None]).clamp_min(0.)
    r["rgb"] = (weights[..., None] * rgbs).sum(dim=-2) + bg_w * bg_rgbs
    t_mids = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
    r["depth"] = (weights * t_mids).sum(dim=-1) / acc.clamp_min(EPS)
    if semantic is not None:  # render.py:240-246, sem_detach=True: no gradient from the semantic loss into the density
        r["semantic"] = (weights.clone().detach()[..., None] * semantic).sum(dim=-2)
    if intensity is not None:  # render.py:248-252
        if intensity.shape != weights.shape:
            intensity = intensity.squeeze(-1)
        r["intensity"] = (weights.clone().detach() * intensity).sum(dim=-1)
    if compute_extras:
        r["acc"] = acc
        expectation = lambda x: (weights * x).sum(dim=-1) / acc.clamp_min(EPS)
        r["distance_mean"] = torch.clip(torch.nan_to_num(torch.exp(expectation(torch.log(t_mids))), torch.inf),
                                        tdist[..., 0], tdist[..., -1])
        t_aug = torch.cat([tdist, t_far], dim=-1)
        w_aug = torch.cat([weights, bg_w], dim=-1)
        pct = weighted_percentile(t_aug, w_aug, [5, 50, 95])
        r["distance_percentile_5"], r["distance_median"], r["distance_percentile_95"] = pct[..., 0], pct[..., 1], pct[..., 2]
    return r


# --------------------------------------------------------------------------------------------
# a-1  Model.forward (ZI/models.py:239-576), instance_obj=False, num_glo_features=0
# --------------------------------------------------------------------------------------------
def make_encoders(sd_np: Dict[str, np.ndarray], mc) -> Dict[str, GridEncoder]:
    from_cfg = {}
    names = [(f"prop_mlp_{i}", mc.prop_cfg(i)) for i in range(mc.num_levels - 1)] + [("nerf_mlp", mc.nerf_mlp)]
    for prefix, cfg in names:
        L = cfg.grid_num_levels
        pls = np.exp2(np.log2(cfg.grid_disired_resolution / cfg.grid_base_resolution) / (L - 1))
        from_cfg[prefix] = GridEncoder(sd_np[f"{prefix}.encoder.embeddings"], sd_np[f"{prefix}.encoder.offsets"],
                                       sd_np[f"{prefix}.encoder.grid_sizes"], pls, cfg.grid_base_resolution)
    return from_cfg


def to_torch_sd(sd_np):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()
            if not k.endswith(("encoder.embeddings", "encoder.offsets", "encoder.grid_sizes"))}


def model_forward(sd_np, mc, batch: Dict[str, torch.Tensor], train_frac=1.0, compute_extras=True,
                  sample_n=7, sample_m=3, rand_jitter: Optional[List[torch.Tensor]] = None,
                  rand_deg: Optional[List[torch.Tensor]] = None, encoders=None, sd_t=None, objects=None):
    """Returns (renderings, ray_history) like the reference.  rand_* are per-level uniform draws
    ([N,1] and [N,S,n]) replacing torch.rand for rand=True; None = deterministic.
    objects: None (Config.instance_obj=False) or the dict made by `make_objects` (dynamic-object branch, row f-1)."""
    encoders = encoders or make_encoders(sd_np, mc)
    sd = sd_t or to_torch_sd(sd_np)
    obj_pose = obj_get_pose(batch["timestamp"], objects["tracks"]) if objects is not None else None
    _, s_to_t = construct_ray_warps(batch["near"], batch["far"], mc.power_lambda)
    init_s_near, init_s_far = 0., 1.
    sdist = torch.cat([torch.full_like(batch["near"], init_s_near), torch.full_like(batch["far"], init_s_far)], dim=-1)
    weights = torch.ones_like(batch["near"])
    prod_num_samples = 1
    renderings, ray_history = [], []
    for i_level in range(mc.num_levels):
        is_prop = i_level < mc.num_levels - 1
        num_samples = mc.num_prop_samples[i_level] if is_prop else mc.num_nerf_samples
        dilation = mc.dilation_bias + mc.dilation_multiplier * (init_s_far - init_s_near) / prod_num_samples
        prod_num_samples *= num_samples
        if i_level > 0 and (mc.dilation_bias > 0 or mc.dilation_multiplier > 0):
            sdist, weights = max_dilate_weights(sdist, weights, dilation, (init_s_near, init_s_far), True)
            sdist, weights = sdist[..., 1:-1], weights[..., 1:-1]
        if mc.anneal_slope > 0:
            anneal = (mc.anneal_slope * train_frac) / ((mc.anneal_slope - 1) * train_frac + 1)
        else:
            anneal = 1.
        logits = torch.where(sdist[..., 1:] > sdist[..., :-1], anneal * torch.log(weights + mc.resample_padding),
                             torch.full_like(sdist[..., :-1], -torch.inf))
        sdist = sample_intervals(sdist, logits, num_samples, (init_s_near, init_s_far),
                                 None if rand_jitter is None else rand_jitter[i_level])
        tdist = s_to_t(sdist)
        means, stds = cast_rays(tdist, batch["origins"], batch["directions"], batch["radii"], batch["base_x"],
                                batch["base_y"], n=sample_n, m=sample_m, std_scale=mc.std_scale,
                                rand_deg=None if rand_deg is None else rand_deg[i_level])
        prefix = f"prop_mlp_{i_level}" if is_prop else "nerf_mlp"
        cfg = mc.prop_cfg(i_level) if is_prop else mc.nerf_mlp
        res = mlp_forward(sd, prefix, cfg, encoders[prefix], means, stds, batch["viewdirs"])
        if objects is not None:
            obj_merge(res, objects, sd, batch, tdist, obj_pose)
        weights = compute_alpha_weights(res["density"], tdist, batch["directions"], mc.opaque_background)
        bg = mc.bg_intensity_range[0] if mc.bg_intensity_range[0] == mc.bg_intensity_range[1] else \
            (mc.bg_intensity_range[0] + mc.bg_intensity_range[1]) / 2
        last = i_level == mc.num_levels - 1
        rendering = volumetric_rendering(res["rgb"], weights, tdist, bg, batch["far"], compute_extras,
                                         semantic=res["semantic"] if last and mc.config.use_semantic else None,
                                         intensity=res["intensity"] if last and mc.config.use_intensity else None)
        renderings.append(rendering)
        res.update(sdist=sdist.clone(), weights=weights.clone(), tdist=tdist.clone())
        ray_history.append(res)
    return renderings, ray_history


# --------------------------------------------------------------------------------------------
# f-1  dynamic-object branch (ZI/models.py:93-177,306-315,401-477; ZI/obj_utils.py:5-28,76-113,116-234,431-475)
# --------------------------------------------------------------------------------------------
def obj_get_pose(time, tracks):
    """obj_utils.py:431-475.  time [N,1]; tracks [N_obj, T, 9] = (center3, theta_z, wlh3, timestamp, track_id) -> [N, N_obj, 9].
    Per ray and track: the two recorded poses nearest in time (a = nearest, b = second nearest), blended as
    w a + (1 - w) b with w = clamp(|t - t_b| / (|t_a - t_b| + 1e-9), 0, 1); every one of the 9 columns is blended, the
    timestamp and the track id included."""
    rec_t = tracks[:, :, 7]                                            # [N_obj, T]
    order = torch.argsort((time[:, :, None] - rec_t[None]).abs(), dim=-1, stable=False)
    ia, ib = order[..., 0], order[..., 1]                              # [N, N_obj]
    obj = torch.arange(tracks.shape[0])[None, :].expand_as(ia)
    pa, pb = tracks[obj, ia], tracks[obj, ib]                          # [N, N_obj, 9]
    ta, tb = pa[..., 7:8], pb[..., 7:8]
    w = ((time[:, :, None] - tb).abs() / ((ta - tb).abs() + 1e-9)).clamp(0, 1)
    return w * pa + (1 - w) * pb


def obj_rotate_yaw_z(p, yaw):
    """obj_utils.py:76-113 with pitch=None.  (sic) p_y is computed from the ALREADY ROTATED p_x (line 106-107)."""
    c, sn = torch.cos(yaw), torch.sin(yaw)
    px = c * p[..., 0] - sn * p[..., 1]
    py = sn * px + c * p[..., 1]
    return torch.stack([px, py, p[..., 2]], dim=-1)


def obj_scale_frames(p, sc_factor):
    """obj_utils.py:5-28 (forward): p / (dim/2 + 1e-9), written as the reference multiplies."""
    dim = torch.tensor([1., 1., 1.]) * sc_factor
    return (1 / (dim / 2 + 1e-9)) * p


def obj_box_pts(pts, viewdirs, obj_pose):
    """obj_utils.py:203-234 + world2object :116-176 (inverse=False).  pts [N,S,3], viewdirs [N,3], obj_pose [N,N_obj,9] ->
    pts_o, dirs_o [N,S,N_obj,3] in box coordinates ([-1,1]^3 inside), intersection_map [N,S,N_obj] bool."""
    N, S = pts.shape[:2]
    center, theta_z, wlh = obj_pose[:, :, :3], obj_pose[:, :, 3], obj_pose[:, :, 4:7]
    pose = torch.repeat_interleave(center, S, dim=0)
    th = torch.repeat_interleave(theta_z, S, dim=0)
    dim = torch.repeat_interleave(wlh, S, dim=0)
    dirs = torch.repeat_interleave(viewdirs, S, dim=0)
    p = pts.reshape(-1, 3)
    t_w_o = obj_rotate_yaw_z(-pose, th)
    n_obj = th.shape[1]
    pts_w = torch.repeat_interleave(p.unsqueeze(1), n_obj, dim=1)
    dirs_w = torch.repeat_interleave(dirs.unsqueeze(1), n_obj, dim=1)
    pts_o = obj_scale_frames(obj_rotate_yaw_z(pts_w, th) + t_w_o, dim)
    dirs_o = obj_scale_frames(obj_rotate_yaw_z(dirs_w, th), dim)
    dirs_o = dirs_o / torch.norm(dirs_o, dim=-1, keepdim=True)
    imap = (pts_o[..., 0].abs() < 1) & (pts_o[..., 1].abs() < 1) & (pts_o[..., 2].abs() < 1)
    return pts_o.reshape(N, S, -1, 3), dirs_o.reshape(N, S, -1, 3), imap.reshape(N, S, -1)


def obj_mlp_forward(sd, prefix, cfg, enc, pts, viewdirs, latent):
    """MLP.forward (ZI/models.py:1036-1265) as configured for ObjMLP: points without a multisample axis and zero std
    (models.py:424-425), warp_fn None, re_weights False, latent split into shape / texture halves, fixed one-hot semantic."""
    F = torch.nn.functional
    feats = enc(pts, bound=1)  # [n, L*C]; unflatten/flatten of models.py:971,976 is the identity on values
    if latent is not None:
        feats = torch.cat([feats, latent[..., :cfg.latent_size // 2] if cfg.split_latent else latent], dim=-1)
    x = _lin(sd, f"{prefix}.density_layer.2", F.relu(_lin(sd, f"{prefix}.density_layer.0", feats)))
    out = dict(density=F.softplus(x[..., 0] + cfg.density_bias), semantic=None, intensity=None)
    if cfg.use_semantic:  # fixed_semantic (models.py:1125-1130)
        sem = torch.zeros(x.shape[:-1] + (cfg.class_num,))
        if cfg.class_type != 255:
            sem[..., cfg.class_type] = 1.
        out["semantic"] = sem
    h = [x, pos_enc(viewdirs, 0, cfg.deg_view)]
    if cfg.split_latent:
        h.append(latent[..., cfg.latent_size // 2:])
    h = torch.cat(h, dim=-1)
    inputs = h
    for i in range(cfg.net_depth_viewdirs):
        h = F.relu(_lin(sd, f"{prefix}.lin_second_stage_{i}", h))
        if i == cfg.skip_layer_dir:
            h = torch.cat([h, inputs], dim=-1)
    rgb = torch.sigmoid(cfg.rgb_premultiplier * _lin(sd, f"{prefix}.rgb_layer", h) + cfg.rgb_bias)
    out["rgb"] = rgb * (1 + 2 * cfg.rgb_padding) - cfg.rgb_padding
    return out


def make_objects(sd_np, tracks, class_ids, obj_cfgs):
    """tracks [N_obj,T,9] numpy; class_ids[track] -> class id (obj_utils.query_class); obj_cfgs {class id: MLPConfig}."""
    encs = {}
    for cid, cfg in obj_cfgs.items():
        L = cfg.grid_num_levels
        pls = np.exp2(np.log2(cfg.grid_disired_resolution / cfg.grid_base_resolution) / (L - 1))
        pre = f"obj_mlp_{cid}"
        encs[cid] = GridEncoder(sd_np[f"{pre}.encoder.embeddings"], sd_np[f"{pre}.encoder.offsets"], sd_np[f"{pre}.encoder.grid_sizes"],
                                pls, cfg.grid_base_resolution)
    return dict(tracks=torch.from_numpy(np.asarray(tracks, np.float32)), class_ids=list(class_ids), cfgs=obj_cfgs, encoders=encs)


def obj_merge(res, objects, sd, batch, tdist, obj_pose):
    """ZI/models.py:401-477 (latent mode, no symmetry): evaluate each track's ObjMLP on the samples inside its box and
    overwrite the static field's per-sample results there; later tracks win where boxes overlap."""
    t_mids = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
    pts_w = t_mids[..., None] * batch["directions"][:, None, :] + batch["origins"][:, None, :]
    pts_o, dirs_o, imap = obj_box_pts(pts_w, batch["viewdirs"], obj_pose)
    for track_id, cid in enumerate(objects["class_ids"]):
        m = imap[:, :, track_id]
        if m.sum() == 0:
            continue
        cfg = objects["cfgs"][cid]
        pts_k, dirs_k = pts_o[m][:, track_id, :], dirs_o[m][:, track_id, :]
        latent = sd[f"latent_vector_dict.obj_latent_{track_id}"][None, :].repeat(pts_k.shape[0], 1)
        o = obj_mlp_forward(sd, f"obj_mlp_{cid}", cfg, objects["encoders"][cid], pts_k, dirs_k, latent)
        for key in ("density", "rgb", "semantic", "intensity"):
            if res.get(key) is None:
                continue
            if o[key] is None:  # the reference assigns None into a tensor here (TypeError): an invalid configuration
                raise TypeError(f"static field predicts '{key}' but ObjMLP does not")
            tmp = torch.zeros_like(res[key])
            tmp[m] = o[key]
            mm = m if m.shape == res[key].shape else m[..., None].expand(res[key].shape)
            res[key] = torch.where(mm, tmp, res[key])
    res["obj_mask"] = imap.sum(-1) > 0


def hash_decay_loss(embeddings, offsets, mult=1.0):
    """ZI/models.py:203-223 for one encoder: segment_coo(param ** 2, idx, reduce='mean').mean() with idx = the level of each row
    (Z/gridencoder/grid.py:138-141), restated with per-level slices."""
    L = len(offsets) - 1
    per = [(embeddings[int(offsets[l]):int(offsets[l + 1])] ** 2).mean(dim=0) for l in range(L)]  # [C] per level
    return mult * torch.stack(per).mean()


def lidar_post(batch, rendering, scale_factor):
    """a-16: Z/render_lidar.py:142-161 -> (points [N,3], labels [N])."""
    depth = rendering["depth"].reshape(-1)
    points = (batch["origins"] + depth[..., None] * batch["directions"]) / scale_factor
    labels = torch.argmax(rendering["semantic"], dim=-1) if "semantic" in rendering else None
    return points, labels


def render_chunked(sd_np, mc, batch, chunk=4096, **kw):
    """Test-time driver (ZI/models.py:1379-1507 for one process): chunk, forward, concat."""
    encoders = make_encoders(sd_np, mc)
    sd_t = to_torch_sd(sd_np)
    n = batch["origins"].shape[0]
    outs = []
    for i in range(0, n, chunk):
        cb = {k: v[i:i + chunk] for k, v in batch.items()}
        r, _ = model_forward(sd_np, mc, cb, encoders=encoders, sd_t=sd_t, **kw)
        outs.append(r[-1])
    return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}


# --------------------------------------------------------------------------------------------
# f-2  point cloud -> range image (NeRF_Lidar_code/src/lidar_utils.py:215-282; Generate_feature.py:14-55;
#      real_to_var lidar_utils.py:348-363)
# --------------------------------------------------------------------------------------------
def range_projection(points, semantic=None, rgb=None, H=32, W=1024, fov_up=10.67, fov_down=-30.67):
    """numpy (float64) restatement of LaserScan.do_range_projection: far -> near scatter, nearest point wins."""
    points = np.asarray(points)
    n = points.shape[0]
    semantic = np.zeros(n, np.float32) if semantic is None else np.asarray(semantic)
    rgb = np.zeros((n, 3), np.float32) if rgb is None else np.asarray(rgb)
    fu, fd = fov_up / 180.0 * np.pi, fov_down / 180.0 * np.pi
    fov = abs(fd) + abs(fu)
    depth = np.linalg.norm(points, 2, axis=1)
    yaw = -np.arctan2(points[:, 1], points[:, 0])
    pitch = np.arcsin(points[:, 2] / depth)
    px = np.floor(0.5 * (yaw / np.pi + 1.0) * W)
    py = np.floor((1.0 - (pitch + abs(fd)) / fov) * H)
    px = np.maximum(0, np.minimum(W - 1, px)).astype(np.int32)
    py = np.maximum(0, np.minimum(H - 1, py)).astype(np.int32)
    order = np.argsort(depth, kind="stable")[::-1]
    out = dict(proj_range=np.full((H, W), -1, np.float32), proj_xyz=np.full((H, W, 3), -1, np.float32),
               proj_semantic=np.full((H, W), -1, np.float32), proj_rgb=np.zeros((H, W, 3), np.float32),
               proj_idx=np.full((H, W), -1, np.int32))
    out["proj_range"][py[order], px[order]] = depth[order]
    out["proj_xyz"][py[order], px[order]] = points[order]
    out["proj_semantic"][py[order], px[order]] = semantic[order]
    out["proj_rgb"][py[order], px[order]] = rgb[order]
    out["proj_idx"][py[order], px[order]] = np.arange(n)[order]
    out["proj_mask"] = (out["proj_idx"] > 0).astype(np.float32)
    return out


def log_range(real):
    """pcs2img(log=True), Generate_feature.py:44-49."""
    real = np.where(real < 0, 0, real) + 0.0001
    return np.clip(np.log2(real + 1) / 6.5, 0, 1)


def real_to_var(real, size=1):
    """lidar_utils.py:348-363 (sic: the window is range(-size, size), i.e. 2*size columns)."""
    return np.var(np.stack([np.roll(real, i, axis=1) for i in range(-size, size)], axis=-1), axis=-1)
