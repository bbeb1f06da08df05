"""Pins the CPU oracle (oracle/nlr_oracle.py) against fixtures produced by the REFERENCE's own
Python (tests/golden/make_golden.py).  CPU-only; runs in the build container and on the GPU box.

Tolerances: the oracle uses the same torch fp32 ops as the reference, so differences come only
from algebraically equal reformulations (index form of sorted_interp) -> 1e-6 absolute unless
stated; whole-forward outputs use 2e-5 (three resampling levels amplify 1-ulp differences).
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden
from oracle import nlr_oracle as orc
from nerflidar_hip import config as nconfig, lidar as nlidar, weights as nweights

T = torch.from_numpy


def _names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")) if "TRAINED" not in p)


def close(a, b, atol=1e-6, rtol=1e-6):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.mark.parametrize("name", _names("fn_max_dilate"))
def test_max_dilate_weights(name):
    g = golden(name)
    td, wd = orc.max_dilate_weights(T(g["t"]), T(g["w"]), float(g["dilation"]), (0., 1.), True)
    np.testing.assert_array_equal(td.numpy(), g["t_dilate"])  # sort + clip: exact
    close(wd, g["w_dilate"], atol=1e-7)


@pytest.mark.parametrize("name", _names("fn_sample_intervals"))
def test_sample_intervals(name):
    g = golden(name)
    n = g["sdist"].shape[-1] - 1
    sd = orc.sample_intervals(T(g["t"]), T(g["logits"]), n, (0., 1.))
    close(sd, g["sdist"], atol=1e-7)
    assert (np.diff(sd.numpy(), axis=-1) >= 0).all()


@pytest.mark.parametrize("name", _names("fn_sorted_interp"))
def test_sorted_interp(name):
    g = golden(name)
    np.testing.assert_array_equal(orc.sorted_interp(T(g["x"]), T(g["xp"]), T(g["fp"])).numpy(), g["out"])


@pytest.mark.parametrize("name", _names("fn_weighted_percentile"))
def test_weighted_percentile(name):
    g = golden(name)
    close(orc.weighted_percentile(T(g["t"]), T(g["w"]), [5, 50, 95]), g["out"], atol=1e-7)


@pytest.mark.parametrize("name", _names("fn_ray_warp"))
def test_ray_warp(name):
    g = golden(name)
    _, s_to_t = orc.construct_ray_warps(T(g["near"]), T(g["far"]), -1.5)
    np.testing.assert_array_equal(s_to_t(T(g["s"])).numpy(), g["t"])


@pytest.mark.parametrize("name", _names("fn_cast_rays"))
def test_cast_rays(name):
    g = golden(name)
    m, s = orc.cast_rays(T(g["tdist"]), T(g["origins"]), T(g["directions"]), T(g["radii"]), T(g["base_x"]),
                         T(g["base_y"]))
    close(m, g["means"], atol=1e-7)
    close(s, g["stds"], atol=0, rtol=1e-6)


@pytest.mark.parametrize("name", _names("fn_contract"))
def test_contract(name):
    g = golden(name)
    z, zs = orc.contract_mean_std(T(g["x"]), T(g["std"]))
    np.testing.assert_array_equal(z.numpy(), g["z"])
    np.testing.assert_array_equal(zs.numpy(), g["zstd"])


@pytest.mark.parametrize("name", _names("fn_pos_enc"))
def test_pos_enc(name):
    g = golden(name)
    np.testing.assert_array_equal(orc.pos_enc(T(g["x"]), 0, 4).numpy(), g["out"])


@pytest.mark.parametrize("name", _names("fn_composite"))
def test_composite(name):
    g = golden(name)
    w = orc.compute_alpha_weights(T(g["density"]), T(g["tdist"]), T(g["dirs"]), "opaque" in name)
    np.testing.assert_array_equal(w.numpy(), g["weights"])
    r = orc.volumetric_rendering(T(g["rgbs"]), w, T(g["tdist"]), 1.0, T(g["far"]), True, semantic=T(g["sem"]),
                                 intensity=T(g["intensity"]))
    for k, v in r.items():
        close(v, g["out_" + k], atol=1e-7)
    assert set("out_" + k for k in r) == {k for k in g if k.startswith("out_")}


def test_lidar_batch():
    """a-0: nerflidar_hip.lidar (product host code) against the reference's numpy."""
    g = golden("fn_lidar_batch")
    d = nlidar.get_directions(g["beams"], g["az"])
    o = np.broadcast_to(g["origin"][None, :], d.shape)
    b = nlidar.cast_lidar_ray_batch(np.array(o, np.float64), d, 0.008, 2.0)
    for k in ("origins", "directions", "viewdirs", "radii", "near", "far", "base_x", "base_y", "cam_idx"):
        np.testing.assert_allclose(b[k], g["out_" + k], rtol=1e-6, atol=1e-9, err_msg=k)
    # the Frobenius quirk: every viewdir has norm 1/sqrt(N)
    np.testing.assert_allclose(np.linalg.norm(b["viewdirs"], axis=-1), 1 / np.sqrt(d.shape[0]), rtol=1e-5)


def test_camera_rays():
    """Config C3 input producer: nerflidar_hip.camera (product host code) against the reference's pixels_to_rays."""
    from nerflidar_hip import camera as ncamera
    g = golden("fn_camera_rays")
    out = ncamera.pixels_to_rays(g["pix_x"], g["pix_y"], np.linalg.inv(g["K"]), g["c2w"])
    for got, k in zip(out, ("origins", "directions", "viewdirs", "radii", "imageplane", "base_x", "base_y")):
        np.testing.assert_allclose(got, g[k], rtol=1e-12, atol=1e-14, err_msg=k)


def test_model_forward_camera_c3():
    """Whole forward on a camera batch (orthonormal base_x/base_y, per-ray radii), hierarchical (64 + 128) levels."""
    from nerflidar_hip import camera as ncamera
    g = golden("camfwd_C3")
    mc = nconfig.workload("C3", int(g["log2_hashmap"]))
    sd = nweights.synth_state_dict(mc, seed=int(g["seed"]), trained_like=True)
    W, H, f = g["cam"]
    b = ncamera.synthetic_camera_batch(width=int(W), height=int(H), focal=float(f), seed=0, rows=g["rows"])
    rend, hist = orc.model_forward(sd, mc, {k: T(v) for k, v in b.items()})
    for k in [k for k in g if k.startswith("out_")]:
        close(rend[-1][k[4:]], g[k], atol=2e-5, rtol=1e-4)
    for lvl in range(mc.num_levels):
        close(hist[lvl]["sdist"][:24], g[f"hist{lvl}_sdist"], atol=2e-5, rtol=1e-4)


def test_grid_numpy_twin_matches_c():
    """The C restatement of kernel_grid against an independently written numpy twin."""
    mc = nconfig.workload("REF", 12)
    sd = nweights.synth_state_dict(mc, seed=3, table_std=0.5)
    rng = np.random.default_rng(0)
    x = rng.random((4096, 3)).astype(np.float32)
    x[:16] = np.array([0.0, 1.0, 0.5], np.float32)  # exact cell boundaries
    x[16:24, 0] = 1.0001  # out of range -> zeros
    x[24:32, 2] = -1e-6
    for prefix, cfg in nweights.mlp_names(mc):
        enc = orc.make_encoders(sd, mc)[prefix]
        c_out, _ = orc.grid_encode_c(x, enc.table, enc.offsets, enc.S, enc.H)
        n_out = orc.grid_encode_numpy(x, enc.table, enc.offsets, enc.S, enc.H)
        np.testing.assert_allclose(c_out, n_out, atol=2e-7, rtol=1e-6)
        assert (c_out[:, 16:32] == 0).all()
        # layout facts of grid.py:122-142 for the shipped sizes
        assert enc.offsets[-1] == sd[f"{prefix}.encoder.embeddings"].shape[0]


def test_grid_dense_levels_match_torch_grid_sample():
    """Independent pin for row a-7 (missing 6): on a DENSE level the hash grid is plain trilinear interpolation of a
    (res + 1)^3 lattice, and `torch.nn.functional.grid_sample(mode='bilinear', align_corners=True)` is a third-party implementation
    of exactly that.  Mapping (gridencoder.cu:138-153, grid.py:162): lattice coordinate pos = x01 * scale + 0.5 with scale = res - 1,
    row index = x + y * (res + 1) + z * (res + 1)^2; grid_sample's normalised coordinate u = 2 * pos / res - 1 lands on the same lattice
    coordinate.  Checked for the dense levels of all three shipped grids (17^3, 33^3, 65^3 rows), C = 1 and C = 4."""
    import torch.nn.functional as F
    mc = nconfig.workload("REF", 21)
    sd = nweights.synth_state_dict(mc, seed=5, table_std=0.5)
    rng = np.random.default_rng(1)
    x = rng.random((20000, 3)).astype(np.float32)
    x[:8] = np.array([[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0, 1, 0.25], [1, 0, 0], [0.999999, 0.5, 0.5], [1e-7, 1e-7, 1e-7],
                      [0.25, 0.75, 1.0]], np.float32)
    checked = 0
    for prefix, cfg in nweights.mlp_names(mc):
        enc = orc.make_encoders(sd, mc)[prefix]
        c_out, _ = orc.grid_encode_c(x, enc.table, enc.offsets, enc.S, enc.H)         # [L, B, C]
        scale, res = orc.level_scale(enc.num_levels, enc.S, enc.H)
        C = enc.level_dim
        for l in range(enc.num_levels):
            step = int(res[l]) + 1
            if step ** 3 > int(enc.offsets[l + 1] - enc.offsets[l]):
                continue                                                              # hashed level: no lattice to sample
            lat = torch.from_numpy(enc.table[int(enc.offsets[l]):int(enc.offsets[l]) + step ** 3].astype(np.float64))
            vol = lat.reshape(step, step, step, C).permute(3, 0, 1, 2)[None]           # [1, C, D = z, H = y, W = x]
            pos = x.astype(np.float64) * float(scale[l]) + 0.5
            u = torch.from_numpy(2.0 * pos / float(res[l]) - 1.0)[None, None, None]    # [1, 1, 1, B, (x, y, z)]
            ref = F.grid_sample(vol, u, mode="bilinear", padding_mode="border", align_corners=True)[0, :, 0, 0].T.numpy()
            # f32 positions / weights in the C checker against float64 here: 1-2 ulp of the lattice coordinate (x scale up to 64)
            np.testing.assert_allclose(c_out[l], ref, atol=2e-5 * np.abs(lat.numpy()).max(), rtol=0)
            checked += 1
    assert checked == 9  # levels 0-2 of PropMLP_0, PropMLP_1 and the NerfMLP grid


def test_grid_tv_oracle_matches_numpy_restatement():
    """kernel_grad_tv (gridencoder.cu:506-601) as the C checker states it against an independently written float64 numpy
    version: dense and hashed levels, cells on the level boundary (a missing left / right neighbour), out-of-range points."""
    from nerflidar_hip import synth
    cfg = nconfig.workload("REF", 12).prop_cfg(1)
    offsets, sizes, pls = nweights.grid_layout(cfg)
    S, H, Cc = float(np.log2(pls)), cfg.grid_base_resolution, cfg.grid_level_dim
    table = synth.table_init(7, "tv", int(offsets[-1]), Cc, 1.0)
    rng = np.random.default_rng(1)
    x = rng.random((3000, 3)).astype(np.float32)
    x[:4] = np.array([[0, 0, 0], [1, 1, 1], [0, 1, 0.5], [1.0001, 0.5, 0.5]], np.float32)
    g0 = rng.standard_normal(table.shape).astype(np.float32) * 1e-3
    got = orc.grid_tv_c(x, table, g0, offsets, 1e-2, S, H)
    scale, res = orc.level_scale(len(offsets) - 1, S, H)
    want = g0.astype(np.float64)
    primes = np.array([1, 2654435761, 805459861], np.uint64)
    for lvl in range(len(offsets) - 1):
        hs, r, step = int(offsets[lvl + 1] - offsets[lvl]), int(res[lvl]), int(res[lvl]) + 1
        dense = step ** 3 <= hs
        def index(p):
            if dense:
                return int(p[0] + p[1] * step + p[2] * step * step) % hs
            h = np.uint64(0)
            for d in range(3):
                h ^= (np.uint64(p[d]) * primes[d]) & np.uint64(0xFFFFFFFF)
            return int(h) % hs
        # (gridencoder.cu:66-84: a level whose stride walk exceeds the table hashes the full coordinate, else x + y s + z s^2)
        tb = table[offsets[lvl]:offsets[lvl + 1]].astype(np.float64)
        for b in range(len(x)):
            if ((x[b] < 0) | (x[b] > 1)).any():
                continue
            # x * scale + 0.5 as one float32 FMA (nvcc's contraction): exact in float64, rounded once
            pg = np.floor((x[b].astype(np.float64) * float(scale[lvl]) + 0.5).astype(np.float32)).astype(np.int64)
            i0 = index(pg)
            resu, idel = np.zeros(Cc), np.zeros(Cc)
            for d in range(3):
                for sgn, ok in ((1, pg[d] < r), (-1, pg[d] > 0)):
                    if ok:
                        q = pg.copy()
                        q[d] += sgn
                        gval = tb[i0] - tb[index(q)]
                        resu += gval
                        idel += gval * gval
            want[offsets[lvl] + i0] += (1e-2 / 6) * resu / np.sqrt(idel + 1e-9)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6)
    assert np.abs(got - g0).max() > 1e-3  # the op did something


def test_grid_layout_full_size():
    """Table footprints quoted in SURVEY a-7 (229 MiB NerfMLP etc.)."""
    mc = nconfig.workload("REF")
    off, sizes, pls = nweights.grid_layout(mc.nerf_mlp)
    assert pls == 2.0 and list(sizes[:3]) == [17, 33, 65] and len(sizes) == 10
    assert int(off[-1]) == 14995560
    assert nweights.grid_layout(mc.prop_cfg(0))[1].shape[0] == 6
    assert nweights.grid_layout(mc.prop_cfg(1))[1].shape[0] == 8


@pytest.mark.parametrize("name", _names("mlp_"))
def test_mlp_forward(name):
    g = golden(name)
    mc = nconfig.workload(str(g["workload"]), int(g["log2_hashmap"]))
    sd = nweights.synth_state_dict(mc, seed=int(g["seed"]), trained_like=bool(g["trained_like"]))
    enc = orc.make_encoders(sd, mc)
    sdt = orc.to_torch_sd(sd)
    res = orc.mlp_forward(sdt, "nerf_mlp", mc.nerf_mlp, enc["nerf_mlp"], T(g["means"]), T(g["stds"]), T(g["viewdirs"]))
    close(res["density"], g["density"], atol=1e-6, rtol=1e-5)
    close(res["rgb"], g["rgb"], atol=1e-6)
    close(res["semantic"], g["semantic"], atol=1e-7)
    if "intensity" in g:
        close(res["intensity"], g["intensity"], atol=1e-6)
    pres = orc.mlp_forward(sdt, "prop_mlp_0", mc.prop_cfg(0), enc["prop_mlp_0"], T(g["means"]), T(g["stds"]), None)
    close(pres["density"], g["prop_density"], atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("name", _names("fwd_"))
def test_model_forward(name):
    g = golden(name)
    lg = int(g["log2_hashmap"])
    mc = nconfig.workload(str(g["workload"]), None if lg < 0 else lg)
    sd = nweights.synth_state_dict(mc, seed=int(g["seed"]), trained_like=bool(g["trained_like"]))
    batch_np = nlidar.synthetic_sweep(width=int(g["width"]), seed=int(g["seed"]), beams=list(g["beams"]))
    for k in ("origins", "directions", "viewdirs", "radii", "near", "far"):
        np.testing.assert_array_equal(batch_np[k], g["in_" + k])  # sweep generator is deterministic
    batch = {k: T(v) for k, v in batch_np.items()}
    rend, hist = orc.model_forward(sd, mc, batch)
    for k in [k for k in g if k.startswith("out_")]:
        close(rend[-1][k[4:]], g[k], atol=2e-5, rtol=1e-4)
    K = g["hist0_sdist"].shape[0]
    for lvl in range(mc.num_levels):
        for k in ("sdist", "weights", "tdist", "density", "rgb", "semantic", "intensity"):
            key = f"hist{lvl}_{k}"
            if key in g:
                close(hist[lvl][k][:K], g[key], atol=2e-5, rtol=1e-4)
        close(rend[lvl]["depth"], g[f"lvl{lvl}_depth"], atol=2e-5, rtol=1e-4)
    # a-16: labels are bit-exact against the reference's argmax
    if "out_semantic" in g:
        _, labels = orc.lidar_post(batch, rend[-1], 1 / 250)
        np.testing.assert_array_equal(labels.numpy(), g["out_semantic"].argmax(-1))
