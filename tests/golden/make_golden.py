#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python.

Runs only in the build container (needs /root/reference; the GPU box never has it).  It imports
/root/reference/NeRF_LiDAR/zipnerf/internal/{math,stepfun,render,coord,models}.py unmodified,
with in-process stubs for the third-party modules that are not installed here (gin,
torch_scatter, pyquaternion, skimage, cv2, accelerate is present) and with `gridencoder`
provided by the CPU restatement of the CUDA kernel (oracle/grid_oracle.c) -- the one piece of the
path that has no runnable reference.  gin bindings are applied by hand as class attributes
(SURVEY Appendix B).

Only DATA is written: inputs, seeds/config names and the reference's outputs, as small .npz
files.  Weights are not stored: they are regenerated from (seed, config) by
nerflidar_hip.weights.synth_state_dict, which this script also uses to fill the reference model.

Usage:  python tests/golden/make_golden.py            (from the repo root)
"""
import os
import sys
import types

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")  # coord.contract_mean_std is @torch.compile
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/NeRF_LiDAR/zipnerf"
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.nn as nn

from oracle import nlr_oracle as orc
from nerflidar_hip import camera as ncamera, config as nconfig, lidar as nlidar, weights as nweights, synth


# ---------------------------------------------------------------------------- stubs
def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _configurable(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


gin = _stub("gin", configurable=_configurable, add_config_file_search_path=lambda *a, **k: None,
            config_scope=None, REQUIRED=None)
gin.config = _stub("gin.config", external_configurable=lambda *a, **k: None)
_stub("torch_scatter", segment_coo=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))
_stub("pyquaternion", Quaternion=object)
sk = _stub("skimage")
sk.metrics = _stub("skimage.metrics", structural_similarity=None, peak_signal_noise_ratio=None)
_stub("cv2")
absl = _stub("absl")
absl.flags = _stub("absl.flags", DEFINE_multi_string=lambda *a, **k: None, DEFINE_string=lambda *a, **k: None, FLAGS=types.SimpleNamespace())


class RefGridEncoder(nn.Module):
    """`gridencoder.GridEncoder` for the reference import: constructor/attrs of grid.py:96-149,
    forward through the CPU restatement of kernel_grid."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False,
                 interpolation='linear', init_std=1e-4):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = per_level_scale, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype_id = {'hash': 0, 'tiled': 1}[gridtype]
        self.interp_id = {'linear': 0, 'smoothstep': 1}[interpolation]
        self.align_corners, self.init_std = align_corners, init_std
        offsets, sizes, off = [], [], 0
        for i in range(num_levels):
            res = int(np.ceil(base_resolution * per_level_scale ** i))
            res = res if align_corners else res + 1
            n = min(2 ** log2_hashmap_size, res ** input_dim)
            n = int(np.ceil(n / 8) * 8)
            sizes.append(res); offsets.append(off); off += n
        offsets.append(off)
        self.register_buffer('offsets', torch.from_numpy(np.array(offsets, dtype=np.int32)))
        idx = torch.empty(off, dtype=torch.long)
        for i in range(num_levels):
            idx[offsets[i]:offsets[i + 1]] = i
        self.register_buffer('idx', idx)
        self.register_buffer('grid_sizes', torch.from_numpy(np.array(sizes, dtype=np.int32)))
        self.embeddings = nn.Parameter(torch.empty(off, level_dim).uniform_(-init_std, init_std))

    def forward(self, inputs, bound=1):
        x01 = (inputs + bound) / (2 * bound)
        prefix = list(x01.shape[:-1])
        flat = x01.reshape(-1, self.input_dim).detach().contiguous().numpy()
        out, _ = orc.grid_encode_c(flat, self.embeddings.detach().numpy(), self.offsets.numpy(),
                                   float(np.log2(self.per_level_scale)), self.base_resolution,
                                   self.gridtype_id, self.align_corners, self.interp_id)
        out = torch.from_numpy(out).permute(1, 0, 2).reshape(flat.shape[0], self.output_dim)
        return out.view(prefix + [self.output_dim])


_stub("gridencoder", GridEncoder=RefGridEncoder)
sys.path.insert(0, REF)
os.chdir("/tmp")
from internal import math as rmath, stepfun as rstep, render as rrender, coord as rcoord, models as rmodels  # noqa: E402
from internal import lidar_utils as rlidar  # noqa: E402
from internal import checkpoints as rcheckpoints  # noqa: E402


# ---------------------------------------------------------------------------- helpers
def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)")


def rnd(seed, stream, shape, lo=0.0, hi=1.0):
    return torch.from_numpy(synth.uniform(seed, stream, shape, lo, hi))


def build_ref_model(mc: nconfig.ModelConfig, sd_np):
    """Reference class tree configured like `mc` (gin bindings applied by hand) + our weights."""
    for cls, cfg in ((rmodels.NerfMLP, mc.nerf_mlp), (rmodels.PropMLP, mc.prop_mlp)):
        for f in ("bottleneck_width", "net_depth_viewdirs", "net_width_viewdirs", "skip_layer_dir", "deg_view",
                  "disable_density_normals", "disable_rgb", "grid_level_dim", "grid_base_resolution",
                  "grid_disired_resolution", "grid_log2_hashmap_size", "class_num"):
            setattr(cls, f, getattr(cfg, f))
    c = mc.config
    cfg_ns = types.SimpleNamespace(use_semantic=c.use_semantic, analytic_gradient=c.analytic_gradient,
                                   use_intensity=c.use_intensity, no_sem_layer=c.no_sem_layer, zero_glo=True,
                                   instance_obj=False, sem_detach=True, vis_num_rays=c.vis_num_rays,
                                   hash_decay_mults=0, symmetrize=False)
    model = rmodels.Model(config=cfg_ns, raydist_fn=mc.raydist_fn, opaque_background=mc.opaque_background,
                          num_prop_samples=tuple(mc.num_prop_samples), num_nerf_samples=mc.num_nerf_samples,
                          num_levels=mc.num_levels, prop_desired_grid_size=list(mc.prop_desired_grid_size))
    ref_sd = model.state_dict()
    new_sd = {}
    for k, v in ref_sd.items():
        if k.endswith(".idx"):
            continue
        assert k in sd_np, f"reference key {k} missing from synth_state_dict"
        t = torch.from_numpy(np.ascontiguousarray(sd_np[k]))
        assert tuple(t.shape) == tuple(v.shape), (k, t.shape, v.shape)
        if v.dtype in (torch.int32, torch.int64):
            assert torch.equal(t.to(v.dtype), v), f"layout mismatch for {k}"
        new_sd[k] = t.to(v.dtype)
    missing = set(sd_np) - set(ref_sd)
    assert not missing, f"synth keys unknown to the reference: {missing}"
    model.load_state_dict(new_sd, strict=False)
    model.eval()
    return model


# ---------------------------------------------------------------------------- per-function fixtures
def gen_stepfun():
    print("stepfun / math fixtures")
    for seed, (N, S) in enumerate([(7, 64), (33, 32)]):
        # monotone fenceposts in [0,1] with a few duplicated (zero-width) bins and weights incl. zeros
        raw = rnd(seed, 1, (N, S + 1))
        t = torch.sort(raw, dim=-1).values
        t[:, 0] = 0.0
        t[:, -1] = 1.0
        t[::3, 5] = t[::3, 4]  # zero-width bin
        w = rnd(seed, 2, (N, S)) ** 4
        w[1::4, 7] = 0.0
        w = w / w.sum(-1, keepdim=True)
        for d in (0.0025 + 0.5 / 64, 0.0025 + 0.5 / 4096):
            td, wd = rstep.max_dilate_weights(t, w, d, domain=(0., 1.), renormalize=True)
            save(f"fn_max_dilate_s{seed}_d{int(d * 1e6)}", t=t, w=w, dilation=np.float32(d), t_dilate=td, w_dilate=wd)
        logits = torch.where(t[..., 1:] > t[..., :-1], torch.log(w), torch.full_like(w, -torch.inf))
        for ns in (32, 64, 128):
            sd = rstep.sample_intervals(False, t, logits, ns, single_jitter=True, domain=(0., 1.))
            save(f"fn_sample_intervals_s{seed}_n{ns}", t=t, logits=logits, sdist=sd)
        # level-0 case: single interval [0,1], weight 1
        t0 = torch.tensor([[0., 1.]]).repeat(N, 1)
        sd0 = rstep.sample_intervals(False, t0, torch.zeros(N, 1), 64, single_jitter=True, domain=(0., 1.))
        save(f"fn_sample_intervals_level0_s{seed}", t=t0, logits=torch.zeros(N, 1), sdist=sd0)
        x = rnd(seed, 3, (N, 11))
        xs = torch.sort(rnd(seed, 4, (N, S + 1)), dim=-1).values
        fs = torch.sort(rnd(seed, 5, (N, S + 1)), dim=-1).values
        save(f"fn_sorted_interp_s{seed}", x=x, xp=xs, fp=fs, out=rmath.sorted_interp(x, xs, fs))
        wp = rnd(seed, 6, (N, S)) ** 2
        wp = wp / wp.sum(-1, keepdim=True)
        save(f"fn_weighted_percentile_s{seed}", t=t, w=wp, out=rstep.weighted_percentile(t, wp, [5, 50, 95]))


def gen_coord_render():
    print("coord / render fixtures")
    for seed, (N, S) in enumerate([(7, 64), (19, 32)]):
        near = torch.full((N, 1), 0.008)
        far = torch.full((N, 1), 2.0)
        _, s_to_t = rcoord.construct_ray_warps('power_transformation', near, far, -1.5)
        s = torch.sort(rnd(seed, 10, (N, S + 1)), dim=-1).values
        s[:, 0] = 0.0
        s[:, -1] = 1.0
        tdist = s_to_t(s)
        save(f"fn_ray_warp_s{seed}", near=near, far=far, s=s, t=tdist)
        o = rnd(seed, 11, (N, 3), -0.01, 0.01)
        d = rnd(seed, 12, (N, 3), -1, 1)
        d = d / d.norm(dim=-1, keepdim=True)
        radii = torch.full((N, 1), 5e-4)
        # LiDAR quirk (base = directions) and a camera-like orthonormal basis
        bx = torch.linalg.cross(d, rnd(seed, 13, (N, 3), -1, 1))
        bx = bx / bx.norm(dim=-1, keepdim=True)
        by = torch.linalg.cross(d, bx)
        for tag, (b0, b1, r) in {"lidar": (d, d, radii), "camera": (bx, by, radii * 4)}.items():
            means, stds = rrender.cast_rays(tdist, o, d, r, False, n=7, m=3, std_scale=0.35,
                                            batch=dict(base_x=b0, base_y=b1))
            save(f"fn_cast_rays_{tag}_s{seed}", tdist=tdist, origins=o, directions=d, radii=r, base_x=b0,
                 base_y=b1, means=means, stds=stds)
        pts = rnd(seed, 14, (N * 5, 3), -3, 3)
        pts[::7] *= 0.1
        st = rnd(seed, 15, (N * 5,), 1e-5, 1e-2)
        zm, zs = rcoord.track_linearize('contract', pts, st)
        save(f"fn_contract_s{seed}", x=pts, std=st, z=zm, zstd=zs)
        v = rnd(seed, 16, (N, 3), -1, 1)
        save(f"fn_pos_enc_s{seed}", x=v, out=rcoord.pos_enc(v, 0, 4, append_identity=True))
        dens = rnd(seed, 17, (N, S), 0, 30) ** 2 / 30
        wts = rrender.compute_alpha_weights(dens, tdist, d * 1.7, opaque_background=True)[0]
        wts_t = rrender.compute_alpha_weights(dens, tdist, d * 1.7, opaque_background=False)[0]
        rgbs = rnd(seed, 18, (N, S, 3))
        sem = torch.softmax(rnd(seed, 19, (N, S, 19), -2, 2), -1)
        inten = rnd(seed, 20, (N, S, 1))
        for tag, ww in (("opaque", wts), ("transparent", wts_t)):
            r = rrender.volumetric_rendering(rgbs, ww, tdist, 1.0, far, True, semantic=sem, intensity=inten,
                                             extras={}, sem_detach=True)
            save(f"fn_composite_{tag}_s{seed}", density=dens, tdist=tdist, dirs=d * 1.7, rgbs=rgbs, sem=sem,
                 intensity=inten, far=far, weights=ww, **{"out_" + k: v for k, v in r.items()})


def gen_lidar():
    print("lidar batch fixture")
    az = np.linspace(270, -90, 40) / 180 * np.pi
    d = rlidar.get_directions(nlidar.LIDAR_ANGLES[:5], az)
    o = np.broadcast_to(np.array([[0.1, -0.2, 0.05]]), d.shape)
    bs = lambda x: np.broadcast_to(x, (d.shape[0], 1))
    b = rlidar.cast_lidar_ray_batch(o, d, dict(near=bs(0.008), far=bs(2.0), lossmult=bs(1.), cam_idx=bs(-1)))
    save("fn_lidar_batch", az=az, beams=np.array(nlidar.LIDAR_ANGLES[:5]), origin=np.array([0.1, -0.2, 0.05]),
         **{"out_" + k: np.asarray(v, np.float32) for k, v in b.items() if v is not None})


# ---------------------------------------------------------------------------- MLP + whole forward
def gen_mlp_and_forward():
    cases = [
        # name, workload, log2_hashmap, N, sweep width, seed, trained_like
        ("REF_small", "REF", 15, 96, 24, 0, True),
        ("C1_small", "C1", 15, 96, 24, 1, True),
        ("C2_small", "C2", 15, 64, 16, 0, True),
        ("REF_full", "REF", None, 64, 16, 1, True),
        ("REF_init", "REF", 15, 64, 16, 0, False),
        # the parameter space of ZI/models.py:MLP beyond the named configurations (nerflidar_hip/config.py:workload)
        ("P_NOSEM", "P_NOSEM", 15, 64, 16, 2, True),
        ("P_NSL", "P_NSL", 15, 64, 16, 3, True),
        ("P_NSL0", "P_NSL0", 15, 64, 16, 4, True),
        ("P_W128I", "P_W128I", 15, 64, 16, 5, True),
        ("P_D3", "P_D3", 15, 64, 16, 6, True),
        ("P_D5", "P_D5", 15, 64, 16, 7, True),
        ("P_F32", "P_F32", 15, 64, 16, 8, True),
        ("P_F20", "P_F20", 15, 64, 16, 9, True),
    ]
    only = os.environ.get("NLR_GOLDEN_ONLY")  # regenerate a subset: comma-separated case names
    if only:
        cases = [c for c in cases if c[0] in only.split(",")]
    for name, wl, lg, N, width, seed, tl in cases:
        print(f"whole-forward fixture {name}")
        mc = nconfig.workload(wl, lg)
        sd_np = nweights.synth_state_dict(mc, seed=seed, trained_like=tl)
        model = build_ref_model(mc, sd_np)
        beams = nlidar.LIDAR_ANGLES[:: max(1, 32 // (N // width))][: N // width]
        batch_np = nlidar.synthetic_sweep(width=width, seed=seed, beams=beams)
        batch = {k: torch.from_numpy(v) for k, v in batch_np.items()}
        with torch.no_grad():
            rend, hist = model(False, batch, train_frac=1.0, compute_extras=True, zero_glo=True)
        out = {}
        for k, v in rend[-1].items():
            if not k.startswith("ray_"):
                out["out_" + k] = v
        K = 24  # per-sample history for the first K rays only (fixture size)
        for lvl, h in enumerate(hist):
            for k in ("sdist", "weights", "tdist", "density", "rgb", "semantic", "intensity"):
                if h.get(k) is not None and not (k == "rgb" and lvl < len(hist) - 1):
                    out[f"hist{lvl}_{k}"] = h[k][:K]
            out[f"lvl{lvl}_depth"] = rend[lvl]["depth"]
            if name.startswith("P_") and lvl < len(hist) - 1:  # the whole rendering dict of the proposal levels
                for k in ("rgb", "acc", "distance_mean", "distance_median", "distance_percentile_5", "distance_percentile_95"):
                    out[f"lvl{lvl}_{k}"] = rend[lvl][k]
        save(f"fwd_{name}", workload=np.array(wl), log2_hashmap=np.array(-1 if lg is None else lg),
             seed=np.array(seed), trained_like=np.array(int(tl)), width=np.array(width),
             beams=np.array(beams), **{"in_" + k: v for k, v in batch_np.items()
                                       if k in ("origins", "directions", "viewdirs", "radii", "near", "far")},
             **out)
        if name in ("REF_small", "C2_small"):
            # MLP-level fixture (rows a-6..a-12): reference MLP.forward on the final level's gaussians
            with torch.no_grad():
                KM = 4
                tdist = hist[-1]["tdist"][:KM]
                means, stds = rrender.cast_rays(tdist, batch["origins"][:KM], batch["directions"][:KM],
                                                batch["radii"][:KM], False, n=7, m=3, std_scale=0.35,
                                                batch=dict(base_x=batch["base_x"][:KM], base_y=batch["base_y"][:KM]))
                res = model.nerf_mlp(False, means, stds, viewdirs=batch["viewdirs"][:KM])
                pres = model.prop_mlp_0(False, means, stds, viewdirs=batch["viewdirs"][:KM])
                feats = model.nerf_mlp.encoder(rcoord.track_linearize('contract', means, stds)[0] / 2, bound=1)
            save(f"mlp_{name}", workload=np.array(wl), log2_hashmap=np.array(lg), seed=np.array(seed),
                 trained_like=np.array(int(tl)), means=means, stds=stds, viewdirs=batch["viewdirs"][:KM],
                 enc_raw=feats, density=res["density"], rgb=res["rgb"], semantic=res["semantic"],
                 **({"intensity": res["intensity"]} if res["intensity"] is not None else {}),
                 prop_density=pres["density"])


from unet_fill import unet_fill  # noqa: E402  (tests/golden/unet_fill.py)


def gen_camera():
    """Config C3: camera rays through the reference's pixels_to_rays, then the reference forward (64 + 128 samples)."""
    print("camera fixtures (C3)")
    from internal import camera_utils as rcam
    W, H, f = 64, 48, 50.0
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])
    c2w = np.concatenate([nlidar.seeded_rotation(17), synth.uniform(0, 9300, (3, 1), -0.02, 0.02).astype(np.float64)], axis=1)
    px, py = np.meshgrid(np.arange(W), np.arange(0, H, 8), indexing="xy")
    px, py = px.reshape(-1).astype(np.float64), py.reshape(-1).astype(np.float64)
    o, d, v, r, ip, bx, by = rcam.pixels_to_rays(px, py, np.linalg.inv(K), c2w)
    save("fn_camera_rays", pix_x=px, pix_y=py, K=K, c2w=c2w, origins=o, directions=d, viewdirs=v, radii=r, imageplane=ip,
         base_x=bx, base_y=by)
    mc = nconfig.workload("C3", 15)
    sd_np = nweights.synth_state_dict(mc, seed=2, trained_like=True)
    model = build_ref_model(mc, sd_np)
    batch_np = ncamera.synthetic_camera_batch(width=W, height=H, focal=f, seed=0, rows=np.arange(0, H, 8))
    for k_, ref_ in (("origins", o), ("directions", d), ("radii", r), ("base_x", bx), ("base_y", by)):
        assert np.allclose(batch_np[k_], ref_, atol=1e-6), k_
    batch = {k: torch.from_numpy(v) for k, v in batch_np.items()}
    with torch.no_grad():
        rend, hist = model(False, batch, train_frac=1.0, compute_extras=True, zero_glo=True)
    out = {"out_" + k: v for k, v in rend[-1].items() if not k.startswith("ray_")}
    for lvl, h in enumerate(hist):
        for k in ("sdist", "weights", "tdist"):
            out[f"hist{lvl}_{k}"] = h[k][:24]
        out[f"lvl{lvl}_depth"] = rend[lvl]["depth"]
    save("camfwd_C3", workload=np.array("C3"), log2_hashmap=np.array(15), seed=np.array(2), trained_like=np.array(1),
         cam=np.array([W, H, f]), rows=np.arange(0, H, 8), **out)


def gen_range_image():
    """f-2: the reference's LaserScan projection + pcs2img normalisation + real_to_var on a synthetic point cloud."""
    print("range image fixture (reference RD/lidar_utils, Generate_feature)")
    sys.path.insert(0, "/root/reference/NeRF_LiDAR/NeRF_Lidar_code/src")
    import lidar_utils as rdl
    n = 6000
    d = nlidar.get_directions(nlidar.LIDAR_ANGLES, np.linspace(270, -90, n // 32) / 180 * np.pi)[:n].astype(np.float64)
    rr = synth.uniform(3, 50, (d.shape[0],), 1.0, 60.0).astype(np.float64)
    pts = d * rr[:, None] + synth.uniform(3, 51, d.shape, -0.01, 0.01)
    pts = np.concatenate([pts, pts[:500] * 1.7, pts[100:300] * 0.5], 0)   # several points per pixel, nearer + farther
    sem = np.floor(synth.uniform(3, 52, (pts.shape[0],), 0, 19)).astype(np.float32)
    rgb = synth.uniform(3, 53, (pts.shape[0], 3), 0, 1)
    real, psem, pmask, prgb, pxyz = rdl.point_cloud_to_range_image(pts, True, H=32, W=256, semantic=sem, rgb=rgb, return_semantic=True,
                                                                   return_mask=True, return_points=True)
    scan = rdl.LaserScan(H=32, W=256, fov_up=10.67, fov_down=-30.67)
    scan.set_points(pts, semantic=sem, rgb=rgb)
    scan.do_range_projection()
    lr = np.clip(np.log2(np.where(real < 0, 0, real) + 0.0001 + 1) / 6.5, 0, 1)
    save("fn_range_image", points=pts, semantic=sem, rgb=rgb, proj_range=real, proj_semantic=psem, proj_mask=pmask, proj_rgb=prgb,
         proj_xyz=pxyz, proj_idx=scan.proj_idx, log_range=lr, var2=rdl.real_to_var(lr, size=2))


def gen_unet():
    print("UNet fixture (reference RD/unet)")
    sys.path.insert(0, "/root/reference/NeRF_LiDAR/NeRF_Lidar_code/src")
    from unet.unet_model import UNet as RefUNet
    for tag, reg in (("logits", False), ("regression", True)):
        m = RefUNet(n_channels=6, n_classes=2, bilinear=True, regression=reg).eval()
        unet_fill(m, 7)
        x = rnd(7, 30, (2, 6, 32, 64))
        with torch.no_grad():
            out = m(x)
        if reg:
            save(f"unet_{tag}", x=x, logits=out[0], reg=out[1])
        else:
            save(f"unet_{tag}", x=x, logits=out)
    # The mask term of one training iteration (ray_drop_train.py:96-101 with vgg = False: the VGG term needs ImageNet weights that
    # cannot be fetched here): reference UNet in train mode (BatchNorm on batch statistics), `F.cross_entropy(prediction, gt_mask) * 1.`,
    # backward.  Stored: loss and the gradients of the first, a middle and the last convolution + a BatchNorm weight.
    m = RefUNet(n_channels=6, n_classes=2, bilinear=True, regression=False).train()
    unet_fill(m, 9)
    x = rnd(9, 30, (4, 6, 32, 64))
    gt = (rnd(9, 31, (4, 32, 64)) > 0.35).long()
    pred = m(x)
    loss = torch.nn.functional.cross_entropy(pred, gt) * 1.
    loss.backward()
    named = dict(m.named_parameters())
    keys = ["inc.double_conv.0.weight", "inc.double_conv.1.weight", "down2.maxpool_conv.1.double_conv.3.weight", "up3.conv.double_conv.0.weight",
            "outc.conv.weight", "outc.conv.bias"]
    out = {}
    for k in keys:  # (fixture size: the first 4 096 values and the norm of each gradient)
        g_ = named[k].grad.reshape(-1)
        out["grad_" + k] = g_[:4096].clone()
        out["gnorm_" + k] = g_.double().norm().float()
    save("unet_ce_step", x=x, gt_mask=gt, loss=loss.detach(), logits=pred.detach()[:1], **out)


def gen_composite_grad():
    """Row f-3: gradients of the reference's compute_alpha_weights + volumetric_rendering (autograd on the reference's own code)
    for a loss that touches every differentiable output and the weights themselves."""
    print("compositing gradient fixtures")
    for tag, opaque, (N, S, K) in (("opaque", True, (24, 64, 19)), ("transparent", False, (17, 128, 5))):
        seed = 40 + int(opaque)
        tdist = torch.sort(rnd(seed, 1, (N, S + 1), 0.05, 2.0), dim=-1)[0]
        tdist[3, 10] = tdist[3, 9]  # a zero-width interval
        dens = (rnd(seed, 2, (N, S), 0, 1) ** 4 * 40).requires_grad_(True)
        dirs = rnd(seed, 3, (N, 3), -1, 1)
        rgbs = rnd(seed, 4, (N, S, 3)).requires_grad_(True)
        sem = torch.softmax(rnd(seed, 5, (N, S, K), -2, 2), -1).requires_grad_(True)
        inten = rnd(seed, 6, (N, S)).requires_grad_(True)
        cot = {k: rnd(seed, 10 + i, sh, -1, 1) for i, (k, sh) in enumerate(dict(rgb=(N, 3), depth=(N,), semantic=(N, K), intensity=(N,),
                                                                               acc=(N,), weights=(N, S)).items())}
        w = rrender.compute_alpha_weights(dens, tdist, dirs, opaque_background=opaque)[0]
        r = rrender.volumetric_rendering(rgbs, w, tdist, 1.0, torch.full((N, 1), 2.5), True, semantic=sem, intensity=inten)
        loss = sum((r[k] * cot[k]).sum() for k in ("rgb", "depth", "semantic", "intensity", "acc")) + (w * cot["weights"]).sum()
        gd, gr, gs, gi = torch.autograd.grad(loss, [dens, rgbs, sem, inten])
        save(f"grad_composite_{tag}", opaque=np.array(int(opaque)), tdist=tdist, density=dens.detach(), dirs=dirs, rgbs=rgbs.detach(),
             sem=sem.detach(), intensity=inten.detach(), **{"cot_" + k: v for k, v in cot.items()}, g_density=gd, g_rgbs=gr, g_sem=gs,
             g_intensity=gi, weights=w.detach(), **{"out_" + k: r[k].detach() for k in ("rgb", "depth", "semantic", "intensity", "acc")})


def _build_obj_model(train=False):
    """The reference `Model` with Config.instance_obj=True (latent mode, shipped ObjMLP gin bindings) on a sweep with three synthetic
    tracks, filled with the synthetic state dict.  train: grid encoders with autograd (RefGridEncoderTrain), model left in train mode."""
    from internal import obj_utils as robj
    from nerflidar_hip import objects as nobj
    lg, width, seed = 12, 16, 2
    mc = nconfig.workload("REF", lg)
    beams = nlidar.LIDAR_ANGLES[::8][:4]
    batch_np = nlidar.synthetic_sweep(width=width, seed=seed, beams=beams)
    N = batch_np["origins"].shape[0]
    batch_np["timestamp"] = nobj.synthetic_timestamps(N, seed)
    names = ["vehicle.car", "vehicle.truck", "vehicle.car"]
    tracks = nobj.synthetic_tracks(batch_np, n_tracks=3, n_times=5, seed=seed)
    cids = [robj.query_class(c) for c in names]
    assert cids == [nobj.query_class(c) for c in names]
    obj_cfgs = {cid: nconfig.obj_mlp_config(cid, latent_size=128, log2_hashmap=lg) for cid in sorted(set(cids))}
    sd_np = nweights.synth_state_dict(mc, seed=seed, trained_like=True)
    sd_np.update(nweights.synth_object_state_dict(obj_cfgs, len(names), seed=seed))
    # gin bindings of nuscenes_single.gin:36-44 (+ the small hash map of this fixture) as class attributes
    for k, v in dict(disable_rgb=False, grid_disired_resolution=1024, density_init=True, disable_density_normals=True, obj_mode=False,
                     bottleneck_width=64, grid_level_dim=2, net_width_viewdirs=32, split_latent=True, grid_log2_hashmap_size=lg).items():
        setattr(rmodels.ObjMLP, k, v)
    for cls, cfg in ((rmodels.NerfMLP, mc.nerf_mlp), (rmodels.PropMLP, mc.prop_mlp)):
        for f in ("bottleneck_width", "net_depth_viewdirs", "net_width_viewdirs", "skip_layer_dir", "deg_view", "disable_density_normals",
                  "disable_rgb", "grid_level_dim", "grid_base_resolution", "grid_disired_resolution", "grid_log2_hashmap_size", "class_num"):
            setattr(cls, f, getattr(cfg, f))
    c = mc.config
    cfg_ns = types.SimpleNamespace(use_semantic=c.use_semantic, analytic_gradient=c.analytic_gradient, use_intensity=c.use_intensity,
                                   no_sem_layer=c.no_sem_layer, zero_glo=True, instance_obj=True, sem_detach=True,
                                   vis_num_rays=c.vis_num_rays, hash_decay_mults=0, symmetrize=False, latent_size=128, fuse_render=False)
    bboxes = ({i: tracks[i] for i in range(len(names))}, {i: names[i] for i in range(len(names))})
    latents = {f"obj_latent_{i}": nn.Parameter(torch.from_numpy(sd_np[f"latent_vector_dict.obj_latent_{i}"])) for i in range(len(names))}
    if train:
        rmodels.GridEncoder = RefGridEncoderTrain
    try:
        model = rmodels.Model(config=cfg_ns, raydist_fn=mc.raydist_fn, opaque_background=mc.opaque_background,
                              num_prop_samples=tuple(mc.num_prop_samples), num_nerf_samples=mc.num_nerf_samples, num_levels=mc.num_levels,
                              prop_desired_grid_size=list(mc.prop_desired_grid_size), bboxes=bboxes, latent_vector_dict=latents)
    finally:
        rmodels.GridEncoder = RefGridEncoder
    ref_sd = model.state_dict()
    new_sd = {}
    for k, v in ref_sd.items():
        if k.endswith(".idx"):
            continue
        assert k in sd_np, f"reference key {k} missing from the synthetic state_dict"
        t = torch.from_numpy(np.ascontiguousarray(sd_np[k]))
        assert tuple(t.shape) == tuple(v.shape), (k, t.shape, v.shape)
        new_sd[k] = t.to(v.dtype)
    assert not (set(sd_np) - set(ref_sd)), set(sd_np) - set(ref_sd)
    model.load_state_dict(new_sd, strict=False)
    model.train() if train else model.eval()
    batch = {k: torch.from_numpy(v) for k, v in batch_np.items()}
    return model, batch, batch_np, tracks, cids, lg, seed, width, beams


def probe_obj_rendering():
    """Row f: does the reference's `Model.obj_rendering` (models.py:579-794, reached through
    `render_image(render_instance=True)`) run under the shipped gin (latent mode)?  Prints / returns the outcome."""
    model, batch, *_ = _build_obj_model()
    import bdb
    import traceback
    stdin = sys.stdin
    sys.stdin = open(os.devnull)  # the path contains a `pdb.set_trace()`: with no terminal the debugger quits at once (BdbQuit)
    try:
        with torch.no_grad():
            rend, hist = model.obj_rendering(False, batch, train_frac=1.0, compute_extras=True, zero_glo=True, track_id=0)
        msg = "ran: keys " + ",".join(sorted(rend[-1]))
    except (Exception, bdb.BdbQuit) as e:  # noqa: BLE001
        where = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_tb(e.__traceback__) if "internal" in f.filename]
        msg = f"raises {type(e).__name__} {str(e).splitlines()[0][:200] if str(e) else ''} at {' <- '.join(reversed(where[-3:]))}"
    finally:
        sys.stdin = stdin
    print("obj_rendering (latent mode, shipped ObjMLP bindings):", msg)
    return msg


def gen_objects():
    """Row f-1: the reference `Model` with Config.instance_obj=True (latent mode, shipped ObjMLP gin bindings) on a sweep with
    three synthetic tracks; per-function fixtures for get_pose / box_pts and the whole forward."""
    print("dynamic-object fixtures")
    from internal import obj_utils as robj
    model, batch, batch_np, tracks, cids, lg, seed, width, beams = _build_obj_model()
    with torch.no_grad():
        pose = robj.get_pose(batch["timestamp"], model.tracks)
        rend, hist = model(False, batch, train_frac=1.0, compute_extras=True, zero_glo=True)
        tdist = hist[-1]["tdist"]
        t_mid = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
        pts_w = t_mid[..., None] * batch["directions"][:, None, :] + batch["origins"][:, None, :]
        pts_o, dirs_o, imap = robj.box_pts(pts=pts_w, viewdirs=batch["viewdirs"], obj_pose=pose, sym=False)
    out = {"out_" + k: v for k, v in rend[-1].items() if not k.startswith("ray_")}
    for lvl, h in enumerate(hist):
        for k in ("sdist", "tdist", "weights", "density", "obj_mask"):
            out[f"hist{lvl}_{k}"] = h[k]
        out[f"lvl{lvl}_depth"] = rend[lvl]["depth"]
    out["hist2_rgb"], out["hist2_semantic"] = hist[-1]["rgb"], hist[-1]["semantic"]
    assert int(imap.sum()) > 50 and all(int(hist[l]["obj_mask"].sum()) > 0 for l in range(3)), "tracks must intersect samples"
    save("obj_REF_small", log2_hashmap=np.array(lg), seed=np.array(seed), width=np.array(width), beams=np.array(beams),
         tracks=tracks, class_ids=np.array(cids), timestamp=batch_np["timestamp"], pose=pose, pts_w=pts_w, pts_o=pts_o, dirs_o=dirs_o,
         imap=imap, **out)


def gen_raydrop_apply():
    """f-4: the reference's ray-drop application itself, NeRF_Lidar_code/src/drop_simulation_rays.py:drop_simulation, run end to
    end (Generate_feature.generate_simulation_data -> LaserScan projection -> runner.test -> mask rule -> sky / road-outlier
    drop) on a synthetic simulated sweep, with the UNet replaced by a runner that returns seeded logits.  Both branches of the
    mask rule (plain threshold; place_car = cars thresholded at their median) in the `save_near` form our apply_ray_drop covers."""
    print("ray-drop application fixture (reference drop_simulation_rays.drop_simulation)")
    import tempfile
    src = "/root/reference/NeRF_LiDAR/NeRF_Lidar_code/src"
    sys.path.insert(0, src)
    for n in ("tkinter", "imageio", "open3d", "matplotlib", "matplotlib.pyplot", "mayavi", "mayavi.mlab"):
        if n not in sys.modules:
            _stub(n)
    _stub("tkinter.tix", Tree=object)
    _stub("model")
    _stub("model.ray_drop_train", ray_drop_learning=object)
    for _ in range(8):
        try:
            import drop_simulation_rays as dsr
            break
        except ModuleNotFoundError as e:  # plotting / IO packages the script imports and this path never calls
            _stub(e.name)
    H, W = 32, 1024
    d = nlidar.get_directions(nlidar.LIDAR_ANGLES, np.linspace(270, -90, 256) / 180 * np.pi).astype(np.float64)
    rr = synth.uniform(5, 60, (d.shape[0],), 1.0, 60.0).astype(np.float64)
    pts = d * rr[:, None] + synth.uniform(5, 61, d.shape, -0.01, 0.01)
    pts = np.concatenate([pts, pts[:700] * 1.3], 0)
    sem = np.floor(synth.uniform(5, 62, (pts.shape[0],), 0, 19)).astype(np.float32)
    sem[::7] = 13.0   # cars
    sem[3::11] = 10.0  # sky
    sem[5::13] = 0.0   # road; some of them far below the sensor
    pts[5::13, 2] -= 4.0
    logits = synth.uniform(5, 63, (2, H, W), -2.0, 2.0).astype(np.float32)

    class Runner:
        def test(self, feats):
            return logits.copy(), np.zeros((1, 1, H, W), np.float32)

    out = dict(points=pts, semantic=sem)  # (the logits are synth.uniform(5, 63, (2, 32, 1024), -2, 2): regenerated by the tests)
    with tempfile.TemporaryDirectory() as tmp:
        sim, data = os.path.join(tmp, "sim"), os.path.join(tmp, "data")
        os.makedirs(sim); os.makedirs(data)
        np.save(os.path.join(sim, "points_0000.npy"), pts)
        np.save(os.path.join(sim, "points_semantic_0000.npy"), sem)
        np.save(os.path.join(data, "c2w.npy"), np.eye(4))
        np.save(os.path.join(data, "c2w_recenter_transform.npy"), np.eye(4))
        for tag, place_car in (("plain", False), ("car", True)):
            dsr.args = types.SimpleNamespace(verbose=False, depth_filter=0, semantic_align=False, filter_thre=0.0, dist_thre=0.0,
                                             place_car=place_car, mask_thre=0.5, pre_mask=False, var=True, normalize=False,
                                             onehot_encoding=False)
            rp, rl = dsr.drop_simulation(sim, tmp, [np.eye(4)], Runner(), {}, {}, save_near=True, datadir=data, mask_thre=0.5)
            out[f"{tag}_points"], out[f"{tag}_labels"] = rp[0], rl[0]
            print(f"  {tag}: {len(rl[0])} of {len(sem)} points remain")
    save("fn_raydrop_apply", **out)


def gen_checkpoint():
    """f-4: a checkpoint written by the reference's own internal/checkpoints.py:save_checkpoint from the reference Model's
    state_dict (the file train.py:559-566 leaves on disk), for the importer to read.  Small architecture (4 x 128, 2^9-entry tables)."""
    print("checkpoint fixture (reference internal/checkpoints.save_checkpoint)")
    import tempfile
    import shutil
    mc = nconfig.workload("C1", 9)
    sd_np = nweights.synth_state_dict(mc, seed=21, trained_like=True)
    model = build_ref_model(mc, sd_np)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    with tempfile.TemporaryDirectory() as tmp:
        path = rcheckpoints.save_checkpoint(tmp, {"state_dict": model.state_dict(), "optimizer": opt.state_dict()}, 1234)
        dst = os.path.join(HERE, "ckpt_ref")
        os.makedirs(dst, exist_ok=True)
        shutil.copy(path, os.path.join(dst, os.path.basename(path)))
        print(f"  wrote ckpt_ref/{os.path.basename(path)}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def gen_losses():
    """Row f-3: the loss terms train.py:326-446 sums, evaluated by the reference's own train_utils / stepfun / math on seeded ray
    histories, with their gradients (autograd on the reference's code) w.r.t. what the optimiser can reach: the proposal weights
    (interlevel terms), the final weights (distortion) and the rendered colours (data term)."""
    print("loss fixtures")
    for name in ("rawpy", "mediapy", "imageio", "tensorboardX", "plyfile", "trimesh", "nuscenes"):  # IO packages of datasets.py
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    _stub("pycolmap", SceneManager=object)
    for _ in range(20):
        try:
            from internal import train_utils as rtu, configs as rcfg
            break
        except ModuleNotFoundError as e:
            _stub(e.name)
    cfg = rcfg.Config()
    cfg.interlevel_loss_mult = 1.0      # (0 in the shipped gin: exercised here all the same)
    N, samples = 48, (64, 64, 32)
    hist = []
    for li, S in enumerate(samples):
        inner = torch.sort(rnd(70, li, (N, S - 1), 0.0, 1.0), dim=-1)[0]
        if li == 2:  # the final level concentrates where the weights are
            inner = torch.sort(0.3 + 0.4 * rnd(70, 10 + li, (N, S - 1), 0.0, 1.0) ** 2, dim=-1)[0]
        sd = torch.cat([torch.zeros(N, 1), inner, torch.ones(N, 1)], dim=-1)
        sd[5, 7] = sd[5, 6]  # a zero-width interval
        w = torch.softmax(rnd(70, 20 + li, (N, S), -4, 4), dim=-1) * rnd(70, 30 + li, (N, 1), 0.5, 1.0)
        hist.append(dict(sdist=sd, weights=w.clone().requires_grad_(True)))
    out = dict(pulse_width=np.asarray(cfg.pulse_width, np.float32), charb_padding=np.float32(cfg.charb_padding),
               anti_mult=np.float32(cfg.anti_interlevel_loss_mult), dist_mult=np.float32(cfg.distortion_loss_mult),
               inter_mult=np.float32(cfg.interlevel_loss_mult))
    for li, h in enumerate(hist):
        out[f"sdist{li}"], out[f"weights{li}"] = h["sdist"], h["weights"].detach()
    for tag, fn in (("anti", rtu.anti_interlevel_loss), ("inter", rtu.interlevel_loss), ("dist", rtu.distortion_loss)):
        loss = fn(hist, cfg)
        grads = torch.autograd.grad(loss, [h["weights"] for h in hist], allow_unused=True)
        out[f"{tag}_loss"] = loss.detach()
        for li, g in enumerate(grads):
            out[f"{tag}_g{li}"] = torch.zeros_like(hist[li]["weights"]) if g is None else g
    # with the dynamic-object mask (train_utils.py:153-154)
    mask = rnd(70, 50, (N, samples[0])) > 0.8
    hist_m = [dict(h) for h in hist]
    hist_m[0]["obj_mask"] = mask
    hist_m[1]["obj_mask"] = rnd(70, 51, (N, samples[1])) > 0.9
    loss = rtu.anti_interlevel_loss(hist_m, cfg)
    g = torch.autograd.grad(loss, [hist[0]["weights"], hist[1]["weights"]])
    out.update(obj_mask0=hist_m[0]["obj_mask"], obj_mask1=hist_m[1]["obj_mask"], anti_masked_loss=loss.detach(), anti_masked_g0=g[0], anti_masked_g1=g[1])
    # data term (train_utils.py:55-117), charbonnier and mse, with a ray mask
    rgbs = [rnd(71, li, (N, 3)).requires_grad_(True) for li in range(3)]
    batch = dict(rgb=rnd(71, 10, (N, 3)), mask_rgb=(rnd(71, 11, (N,)) > 0.3))
    for kind in ("charb", "mse"):
        cfg.data_loss_type = kind
        cfg.data_coarse_loss_mult = 0.5
        loss, _ = rtu.compute_data_loss(batch, [dict(rgb=r) for r in rgbs], cfg)
        g = torch.autograd.grad(loss, rgbs)
        out[f"data_{kind}_loss"] = loss.detach()
        for li in range(3):
            out[f"data_{kind}_g{li}"] = g[li]
    out.update(data_rgb=batch["rgb"], data_mask=batch["mask_rgb"], data_coarse_mult=np.float32(0.5), **{f"data_pred{li}": r.detach() for li, r in enumerate(rgbs)})
    save("fn_losses", **out)


def gen_lr_schedule():
    """The reference's learning-rate function (internal/math.py:54-85, the defaults of configs.py:85-88 and a short schedule) at a few
    steps, for `training.learning_rate_decay`."""
    print("learning-rate schedule fixture")
    steps = np.array([0, 1, 10, 100, 1000, 2500, 5000, 12500, 24999, 25000, 30000])
    out = dict(steps=steps)
    for tag, kw in (("default", dict(lr_init=0.01, lr_final=0.001, max_steps=25000, lr_delay_steps=5000, lr_delay_mult=1e-8)),
                    ("short", dict(lr_init=0.02, lr_final=0.0005, max_steps=3000, lr_delay_steps=600, lr_delay_mult=0.01)),
                    ("nodelay", dict(lr_init=0.01, lr_final=0.001, max_steps=25000, lr_delay_steps=0, lr_delay_mult=1.0))):
        out[tag] = np.array([rmath.learning_rate_decay(int(s_), **kw) for s_ in steps], np.float64)
        out[tag + "_kw"] = np.array([kw["lr_init"], kw["lr_final"], kw["max_steps"], kw["lr_delay_steps"], kw["lr_delay_mult"]], np.float64)
    save("fn_lr_schedule", **out)


def gen_trained():
    """a TRAINED, well-conditioned scene through the reference.  `tests/golden/ckpt_trained/checkpoint_<step>.ckpt`
    was written on the GPU box by `python -m nerflidar_hip.train_scene` (`training.training_step` on the analytic scene of
    nerflidar_hip/scene.py, saved by `checkpoints.save_checkpoint` in the reference's format; `train_summary.json` beside it is that
    run's log).  Here the REFERENCE restores it - `internal/checkpoints.restore_checkpoint` into the reference `Model`
    (render_lidar.py:74) - and renders rays of a held-out sweep (a sensor position no training ray started from) with
    `Model.forward`; stored: inputs, the reference's outputs for every ray, per-sample history of the first rays."""
    print("trained-scene fixture (reference restore_checkpoint + Model.forward)")
    import json
    for sub in ("ckpt_trained", "ckpt_trained_c2"):   # the shipped architecture + intensity head; the benchmark architecture (C2)
        _gen_trained_one(os.path.join(HERE, sub))


def _gen_trained_one(ck_dir):
    import json
    summ = json.load(open(os.path.join(ck_dir, "train_summary.json")))["summary"]
    wl, lg = summ["workload"], summ["log2_hashmap"]
    mc = nconfig.workload(wl, lg)
    model = build_ref_model(mc, nweights.synth_state_dict(mc, seed=0, trained_like=False))   # placeholder weights, replaced next
    step = rcheckpoints.restore_checkpoint(ck_dir, model)
    assert step == summ["steps"], (step, summ["steps"])
    model.eval()
    sweep_idx, width, every = 100, 1024, 40
    full = nlidar.synthetic_sweep(width=width, seed=0, sweep_idx=sweep_idx)   # viewdirs carry the whole sweep's Frobenius norm
    idx = np.arange(0, full["origins"].shape[0], every)
    batch_np = {k: np.ascontiguousarray(v[idx]) for k, v in full.items()}
    batch = {k: torch.from_numpy(v) for k, v in batch_np.items()}
    with torch.no_grad():
        rend, hist = model(False, batch, train_frac=1.0, compute_extras=True, zero_glo=True)
    out = {"out_" + k: v for k, v in rend[-1].items() if not k.startswith("ray_")}
    K = 24
    for lvl, h in enumerate(hist):
        for k in ("sdist", "weights", "tdist", "density", "rgb", "semantic", "intensity"):
            if h.get(k) is not None and not (k == "rgb" and lvl < len(hist) - 1):
                out[f"hist{lvl}_{k}"] = h[k][:K]
        out[f"lvl{lvl}_depth"] = rend[lvl]["depth"]
    # how concentrated the trained field is (the white-noise fixtures spread their weight over many samples)
    w = hist[-1]["weights"]
    print(f"   {len(idx)} rays; max weight per ray: median {float(w.max(-1).values.median()):.3f}; depth {float(rend[-1]['depth'].min()):.4f}.."
          f"{float(rend[-1]['depth'].max()):.4f}; labels {sorted(set(rend[-1]['semantic'].argmax(-1).tolist()))}")
    save(f"fwd_TRAINED_{wl}", workload=np.array(wl), log2_hashmap=np.array(lg), ckpt_step=np.array(step), sweep_idx=np.array(sweep_idx),
         width=np.array(width), ray_index=idx, **{"in_" + k: v for k, v in batch_np.items()
                                                 if k in ("origins", "directions", "viewdirs", "radii", "near", "far")}, **out)



class _RefGridFn(torch.autograd.Function):
    """Autograd through the CPU restatement of the grid kernels (forward cu:87-199, backward cu:248-340) for the reference import in
    training mode: gradients reach the embeddings (positions carry none: Model.stop_level_grad)."""

    @staticmethod
    def forward(ctx, flat, emb, enc):
        out, _ = orc.grid_encode_c(flat.numpy(), emb.detach().numpy(), enc.offsets.numpy(), float(np.log2(enc.per_level_scale)),
                                   enc.base_resolution, enc.gridtype_id, enc.align_corners, enc.interp_id)
        ctx.enc, ctx.flat, ctx.n = enc, flat, emb.shape[0]
        return torch.from_numpy(out).permute(1, 0, 2).reshape(flat.shape[0], enc.output_dim)

    @staticmethod
    def backward(ctx, g):
        enc = ctx.enc
        gl = g.reshape(g.shape[0], enc.num_levels, enc.level_dim).permute(1, 0, 2).contiguous().numpy()
        gt, _ = orc.grid_backward_c(gl, ctx.flat.numpy(), enc.offsets.numpy(), ctx.n, enc.level_dim, float(np.log2(enc.per_level_scale)),
                                    enc.base_resolution, None, enc.gridtype_id, enc.align_corners, enc.interp_id)
        return None, torch.from_numpy(gt), None


class RefGridEncoderTrain(RefGridEncoder):
    def forward(self, inputs, bound=1):
        x01 = (inputs + bound) / (2 * bound)
        prefix = list(x01.shape[:-1])
        flat = x01.reshape(-1, self.input_dim).detach().contiguous()
        return _RefGridFn.apply(flat, self.embeddings, self).view(prefix + [self.output_dim])


def gen_train_step():
    """Row f-3, the WHOLE training step (missing 3): the reference's `model(...)` in training mode on one batch, then
    the reference's OWN loss assembly - lines 283-453 of train.py, read from the file and executed here, not transcribed: masks,
    data term, depth / semantic / intensity terms, anti-aliased interlevel and distortion terms, their weights - and `.backward()`.
    Stored: every loss term, the total, and the gradients of named parameters of all three MLPs.  Deterministic sample positions
    (rand=False): the jitter only moves the samples, the composition of the step is what this pins."""
    print("training-step fixture")
    import tempfile
    import textwrap
    for name in ("rawpy", "mediapy", "imageio", "tensorboardX", "plyfile", "trimesh", "nuscenes"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    _stub("pycolmap", SceneManager=object)
    for _ in range(20):
        try:
            from internal import train_utils as rtu, configs as rcfg
            break
        except ModuleNotFoundError as e:
            _stub(e.name)
    mc = nconfig.workload("REF", 12)
    mc.config.use_intensity = True          # all three heads, so that every term of the assembly is live
    mc.__post_init__()
    seed, width = 3, 12
    sd_np = nweights.synth_state_dict(mc, seed=seed, trained_like=True)
    rmodels.GridEncoder = RefGridEncoderTrain
    try:
        model = build_ref_model(mc, sd_np)
    finally:
        rmodels.GridEncoder = RefGridEncoder
    model.train()
    beams = nlidar.LIDAR_ANGLES[::4]
    batch_np = nlidar.synthetic_sweep(width=width, seed=seed, beams=beams)
    N = batch_np["origins"].shape[0]
    batch = {k: torch.from_numpy(v) for k, v in batch_np.items()}
    # supervision of the batch (datasets.py nuScenes loader keys read by train.py:283-424)
    sup = dict(rgb=rnd(80, 1, (N, 3)), depth=rnd(80, 2, (N,), 0.2, 1.8), intensity=rnd(80, 3, (N,), 0.0, 1.0),
               semantic=(rnd(80, 4, (N,)) * 19).floor().clamp(0, 18), mask=(rnd(80, 5, (N,)) > 0.7).float(),
               patch_mask=torch.zeros(N), lidar_mask=(rnd(80, 6, (N,)) > 0.5).float())
    sup["semantic"][::7] = 255              # unlabelled rays
    sup["depth"][1::9] = 0.0                # rays without a depth target
    batch.update({k: v.clone() for k, v in sup.items()})
    config = rcfg.Config()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "depth"))
    open(os.path.join(tmp, "depth", "x"), "w").write("1")
    config.data_dir, config.dataset_loader, config.patch_size = tmp, "nusc", 1
    config.lidar_supervision, config.only_lidar_supervison, config.pose_refine = True, False, False
    config.use_semantic, config.use_intensity, config.instance_obj, config.latent_size = True, True, False, 0
    config.hash_decay_mults, config.symmetrize = 0.0, False
    train_frac, step = 0.37, 5
    renderings, ray_history = model(False, batch, train_frac=train_frac, compute_extras=True, sample_n=7, sample_m=3, zero_glo=False,
                                    step=step, max_step=25000, curr_track=None)
    src = open(os.path.join(REF, "train.py")).read().split("\n")
    first = next(i for i, l in enumerate(src) if l.strip() == "losses = {}")
    last = next(i for i, l in enumerate(src) if l.strip() == "loss = sum(losses.values())")
    block = textwrap.dedent("\n".join(src[first:last + 1]))
    ns = dict(torch=torch, nn=nn, os=os, train_utils=rtu, config=config, batch=batch, renderings=renderings, ray_history=ray_history,
              model=model, module=model, step=step, start_step=config.start_step, end_step=config.end_step, latent_vector_dict={})
    exec(compile(block, "train.py[%d:%d]" % (first + 1, last + 1), "exec"), ns)
    losses, loss = ns["losses"], ns["loss"]
    loss.backward()
    for p_ in model.parameters():            # train_utils.clip_gradients' unconditional nan_to_num_
        if p_.grad is not None:
            p_.grad.nan_to_num_()
    out = dict(workload=np.array("REF"), log2_hashmap=np.array(12), seed=np.array(seed), width=np.array(width), beams=np.array(beams),
               train_frac=np.float32(train_frac), loss=loss.detach(), train_py_lines=np.array([first + 1, last + 1]),
               anti_mult=np.float32(config.anti_interlevel_loss_mult), inter_mult=np.float32(config.interlevel_loss_mult),
               dist_mult=np.float32(config.distortion_loss_mult), pulse_width=np.asarray(config.pulse_width, np.float32),
               data_loss_type=np.array(config.data_loss_type), charb_padding=np.float32(config.charb_padding),
               data_coarse_mult=np.float32(config.data_coarse_loss_mult), data_mult=np.float32(config.data_loss_mult),
               mask_rgb=batch["mask_rgb"])
    for k, v in losses.items():
        out["loss_" + k] = v.detach()
    for k, v in sup.items():
        out["sup_" + k] = v
    out["out_depth"], out["out_rgb"] = renderings[-1]["depth"].detach(), renderings[-1]["rgb"].detach()
    named = dict(model.named_parameters())
    for k in ("nerf_mlp.density_layer.0.weight", "nerf_mlp.density_layer.2.bias", "nerf_mlp.lin_second_stage_0.weight",
              "nerf_mlp.lin_second_stage_1.weight", "nerf_mlp.rgb_layer.weight", "nerf_mlp.sem_layer.2.weight",
              "nerf_mlp.intensity_layer.0.weight", "nerf_mlp.encoder.embeddings", "prop_mlp_0.density_layer.0.weight",
              "prop_mlp_0.encoder.embeddings", "prop_mlp_1.density_layer.2.weight", "prop_mlp_1.encoder.embeddings"):
        g = named[k].grad
        assert g is not None and float(g.abs().sum()) > 0, k
        out["grad_" + k] = g
    print("   loss terms:", {k: float(v.detach()) for k, v in losses.items()}, "total", float(loss.detach()))
    save("train_step_REF", **out)
    # The mask logic alone (train.py: from `if config.dataset_loader == 'nusc':` to `batch['mask_rgb'] = rgb_mask`), executed from the
    # reference's file for every combination of the four switches it reads.
    m0 = next(i for i, l in enumerate(src) if l.strip().startswith("if config.dataset_loader == 'nusc':"))
    m1 = next(i for i, l in enumerate(src) if l.strip() == "batch['mask_rgb'] = rgb_mask")
    mblock = textwrap.dedent("\n".join(src[m0:m1 + 1]))
    M = 257
    raw = dict(mask=(rnd(81, 1, (M,)) > 0.6).float(), patch_mask=torch.zeros(M), depth=rnd(81, 2, (M,), -0.2, 1.0), aug_mask=(rnd(81, 5, (M,)) > 0.8).float(),
               semantic=torch.where(rnd(81, 3, (M,)) > 0.8, torch.full((M,), 255.0), (rnd(81, 4, (M,)) * 19).floor()), lidar_mask=(rnd(81, 6, (M,)) > 0.5).float())
    mout = {"in_" + k: v for k, v in raw.items()}
    for code in range(16):
        lid, only, inst, aug = bool(code & 1), bool(code & 2), bool(code & 4), bool(code & 8)
        config.lidar_supervision, config.only_lidar_supervison, config.instance_obj, config.aug_road = lid, only, inst, aug
        bb = {k: v.clone() for k, v in raw.items()}
        ns2 = dict(torch=torch, config=config, batch=bb)
        exec(compile(mblock, "train.py[%d:%d]" % (m0 + 1, m1 + 1), "exec"), ns2)
        mout[f"c{code}_mask_rgb"], mout[f"c{code}_depth_mask"], mout[f"c{code}_sem_mask"] = ns2["rgb_mask"], ns2["depth_mask"], ns2["sem_mask"]
    save("fn_nusc_masks", train_py_lines=np.array([m0 + 1, m1 + 1]), **mout)


def gen_train_step_obj():
    """Row f-1 x f-3: the training step of the SHIPPED configuration, `Config.instance_obj = True`
    (nuscenes_single.gin:13): the reference's `model(...)` in training mode with the dynamic-object branch (models.py:401-477: ObjMLPs on
    the samples inside the boxes, detached on the proposal levels, latent codes per track), then lines 283-453 of train.py executed from the
    reference's file (mask logic with instance_obj, latent regulariser, obj_mask in the interlevel term), `.backward()`.  Stored: loss
    terms, the gradients of ObjMLP parameters, of the latent codes and of static parameters."""
    print("training-step fixture with dynamic objects")
    import tempfile
    import textwrap
    for name in ("rawpy", "mediapy", "imageio", "tensorboardX", "plyfile", "trimesh", "nuscenes"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    _stub("pycolmap", SceneManager=object)
    for _ in range(20):
        try:
            from internal import train_utils as rtu, configs as rcfg
            break
        except ModuleNotFoundError as e:
            _stub(e.name)
    model, batch, batch_np, tracks, cids, lg, seed, width, beams = _build_obj_model(train=True)
    N = batch_np["origins"].shape[0]
    sup = dict(rgb=rnd(90, 1, (N, 3)), depth=rnd(90, 2, (N,), 0.05, 0.4), semantic=(rnd(90, 4, (N,)) * 19).floor().clamp(0, 18),
               mask=(rnd(90, 5, (N,)) > 0.7).float(), patch_mask=torch.zeros(N), lidar_mask=(rnd(90, 6, (N,)) > 0.5).float())
    sup["semantic"][::7] = 255
    sup["depth"][1::9] = 0.0
    batch.update({k: v.clone() for k, v in sup.items()})
    config = rcfg.Config()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "depth"))
    open(os.path.join(tmp, "depth", "x"), "w").write("1")
    config.data_dir, config.dataset_loader, config.patch_size = tmp, "nusc", 1
    config.lidar_supervision, config.only_lidar_supervison, config.pose_refine = True, False, False
    config.use_semantic, config.use_intensity, config.instance_obj, config.latent_size = True, False, True, 128
    config.hash_decay_mults, config.symmetrize = 0.0, False
    train_frac, step = 0.61, 9
    renderings, ray_history = model(False, batch, train_frac=train_frac, compute_extras=True, sample_n=7, sample_m=3, zero_glo=False,
                                    step=step, max_step=25000, curr_track=None)
    assert all(int(h["obj_mask"].sum()) > 0 for h in ray_history)
    src = open(os.path.join(REF, "train.py")).read().split("\n")
    first = next(i for i, l in enumerate(src) if l.strip() == "losses = {}")
    last = next(i for i, l in enumerate(src) if l.strip() == "loss = sum(losses.values())")
    block = textwrap.dedent("\n".join(src[first:last + 1]))
    ns = dict(torch=torch, nn=nn, os=os, train_utils=rtu, config=config, batch=batch, renderings=renderings, ray_history=ray_history,
              model=model, module=model, step=step, start_step=config.start_step, end_step=config.end_step,
              latent_vector_dict=model.latent_vector_dict)
    exec(compile(block, "train.py[%d:%d]" % (first + 1, last + 1), "exec"), ns)
    losses, loss = ns["losses"], ns["loss"]
    loss.backward()
    for p_ in model.parameters():
        if p_.grad is not None:
            p_.grad.nan_to_num_()
    out = dict(log2_hashmap=np.array(lg), seed=np.array(seed), width=np.array(width), beams=np.array(beams), tracks=tracks,
               class_ids=np.array(cids), timestamp=batch_np["timestamp"], train_frac=np.float32(train_frac), loss=loss.detach(),
               train_py_lines=np.array([first + 1, last + 1]), latent_reg=np.float32(config.latent_reg), mask_rgb=batch["mask_rgb"])
    for k, v in losses.items():
        out["loss_" + k] = v.detach()
    for k, v in sup.items():
        out["sup_" + k] = v
    for lvl, h in enumerate(ray_history):
        out[f"hist{lvl}_obj_mask"] = h["obj_mask"]
    out["out_depth"], out["out_rgb"] = renderings[-1]["depth"].detach(), renderings[-1]["rgb"].detach()
    named = dict(model.named_parameters())
    cid = cids[0]
    keys = [f"obj_mlp_{cid}.density_layer.0.weight", f"obj_mlp_{cid}.rgb_layer.weight", f"obj_mlp_{cid}.lin_second_stage_0.weight",
            f"obj_mlp_{cid}.encoder.embeddings", "nerf_mlp.density_layer.0.weight", "nerf_mlp.rgb_layer.weight", "nerf_mlp.encoder.embeddings",
            "prop_mlp_0.density_layer.0.weight", "prop_mlp_1.encoder.embeddings"]
    for k in keys:
        g = named[k].grad
        assert g is not None and float(g.abs().sum()) > 0, k
        out["grad_" + k] = g
    lat = []
    for t in range(len(cids)):
        g = model.latent_vector_dict[f"obj_latent_{t}"].grad
        lat.append(torch.zeros(128) if g is None else g)
    out["grad_latents"] = torch.stack(lat)
    assert float(out["grad_latents"].abs().sum()) > 0
    print("   loss terms:", {k: float(v.detach()) for k, v in losses.items()}, "total", float(loss.detach()),
          "owned samples per level:", [int(h["obj_mask"].sum()) for h in ray_history])
    save("train_step_OBJ", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if os.environ.get("NLR_GOLDEN_ONLY") == "losses":
        gen_losses()
        raise SystemExit(0)
    if os.environ.get("NLR_GOLDEN_ONLY") == "unet":
        gen_unet()
        raise SystemExit(0)
    if os.environ.get("NLR_GOLDEN_ONLY") == "obj_probe":
        probe_obj_rendering()
        raise SystemExit(0)
    if os.environ.get("NLR_GOLDEN_ONLY") == "train_step_obj":
        gen_train_step_obj()
        raise SystemExit(0)
    if os.environ.get("NLR_GOLDEN_ONLY") == "train_step":
        gen_train_step()
        raise SystemExit(0)
    if os.environ.get("NLR_GOLDEN_ONLY") == "lr":
        gen_lr_schedule()
        raise SystemExit(0)
    if os.environ.get("NLR_GOLDEN_ONLY") == "trained":
        gen_trained()
        raise SystemExit(0)
    if os.environ.get("NLR_GOLDEN_ONLY") == "f4":
        gen_raydrop_apply()
        gen_checkpoint()
        raise SystemExit(0)
    if os.environ.get("NLR_GOLDEN_ONLY"):  # only some whole-forward fixtures
        gen_mlp_and_forward()
        raise SystemExit(0)
    gen_stepfun()
    gen_coord_render()
    gen_lidar()
    gen_mlp_and_forward()
    gen_unet()
    gen_camera()
    gen_range_image()
    gen_objects()
    gen_composite_grad()
    gen_raydrop_apply()
    gen_checkpoint()
    gen_losses()
    gen_train_step()
    gen_train_step_obj()
    gen_lr_schedule()
    gen_trained()
    print("done")
