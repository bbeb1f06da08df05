"""Scope rows f-4 and 8b "who calls it": checkpoint importer, ray-drop application, render_lidar driver."""
import os

import numpy as np
import pytest
import torch

from nerflidar_hip import checkpoints as ck
from nerflidar_hip import config as nconfig
from nerflidar_hip import raydrop
from nerflidar_hip import weights as nweights


def _sd(name, log2=12, seed=3):
    mc = nconfig.workload(name, log2)
    return mc, nweights.synth_state_dict(mc, seed=seed)


def test_checkpoint_roundtrip_and_latest(tmp_path):
    mc, sd = _sd("REF")
    ck.save_checkpoint(tmp_path, sd, 5)
    ck.save_checkpoint(tmp_path, {k: v * 2 for k, v in sd.items()}, 25000)
    ck.save_checkpoint(tmp_path, sd, 900)
    # numeric, not lexicographic, ordering (ZI/checkpoints.py:12-23)
    assert os.path.basename(ck.latest_checkpoint(tmp_path)) == "checkpoint_25000.ckpt"
    got, step = ck.load_checkpoint(tmp_path)
    assert step == 25000 and set(got) == set(sd)
    for k in sd:
        np.testing.assert_array_equal(got[k], sd[k] * 2)
    got, step = ck.load_checkpoint(tmp_path, step=5)
    assert step == 5
    np.testing.assert_array_equal(got["nerf_mlp.rgb_layer.weight"], sd["nerf_mlp.rgb_layer.weight"])
    got, step = ck.load_checkpoint(os.path.join(tmp_path, "checkpoint_900.ckpt"))
    assert step == 900
    with pytest.raises(ValueError):
        ck.load_checkpoint(tmp_path, step=6)
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint(os.path.join(tmp_path, "nope"))


def test_reference_format_on_disk(tmp_path):
    """What `restore_checkpoint` (ZI/checkpoints.py:49-54) reads: a dict with 'state_dict' (tensors) and 'step'."""
    mc, sd = _sd("C1")
    path = ck.save_checkpoint(tmp_path, sd, 7, optimizer_state={"state": {}, "param_groups": []})
    blob = torch.load(path, map_location="cpu", weights_only=True)
    assert set(blob) == {"step", "state_dict", "optimizer"} and blob["step"] == 7
    assert all(isinstance(v, torch.Tensor) for v in blob["state_dict"].values())


@pytest.mark.parametrize("name", ["REF", "C1", "C2"])
def test_infer_model_config_from_shapes(name):
    mc, sd = _sd(name, log2=13)
    # a real checkpoint also carries buffers and (shipped gin) dynamic-object parameters: both must be set aside
    sd = dict(sd)
    sd["nerf_mlp.encoder.offsets"] = np.zeros(11, np.int32)
    sd["nerf_mlp.encoder.idx"] = np.zeros(8, np.int32)
    sd["nerf_mlp.encoder.grid_sizes"] = np.zeros(10, np.int32)
    sd["obj_mlp_car.density_layer.0.weight"] = np.zeros((64, 14), np.float32)
    sd["module.latent_vector_dict.3"] = np.zeros(128, np.float32)
    keep, ignored = ck.split_state_dict(sd)
    assert sorted(ignored) == ["latent_vector_dict.3", "obj_mlp_car.density_layer.0.weight"]
    assert not any("encoder.idx" in k or "encoder.offsets" in k or "grid_sizes" in k for k in keep)
    base = nconfig.ModelConfig(num_prop_samples=mc.num_prop_samples, num_nerf_samples=mc.num_nerf_samples)
    got = ck.infer_model_config(keep, base)
    assert got.num_levels == mc.num_levels and got.level_samples() == mc.level_samples()
    for a, b in zip(nweights.mlp_names(got), nweights.mlp_names(mc)):
        assert a[0] == b[0]
        for f in ("grid_level_dim", "grid_num_levels", "grid_log2_hashmap_size", "grid_disired_resolution", "disable_rgb"):
            assert getattr(a[1], f) == getattr(b[1], f), (a[0], f)
        assert nweights.mlp_param_shapes(a[1]) == nweights.mlp_param_shapes(b[1])
    assert got.nerf_mlp.use_intensity == mc.nerf_mlp.use_intensity
    assert got.nerf_mlp.no_sem_layer == mc.nerf_mlp.no_sem_layer and got.nerf_mlp.class_num == 19
    assert got.nerf_mlp.skip_layer_dir == 0 and got.nerf_mlp.deg_view == 4


def test_infer_model_config_rejects_incomplete():
    mc, sd = _sd("REF")
    bad = {k: v for k, v in sd.items() if k != "nerf_mlp.rgb_layer.bias"}
    with pytest.raises(KeyError):
        ck.infer_model_config(bad)
    bad = dict(sd)
    bad["prop_mlp_0.encoder.embeddings"] = bad["prop_mlp_0.encoder.embeddings"][:-8]
    with pytest.raises(ValueError):
        ck.infer_model_config(bad)


def _ray_drop_np(proj, logits, mask_thre, place_car):
    """numpy restatement of drop_simulation_rays.py:88-166 (save_near, no depth filter) used as the checker."""
    e = np.exp(logits - logits.max(0, keepdims=True))
    p = (e / e.sum(0, keepdims=True))[1]
    if place_car:
        car = proj["proj_semantic"] == 13
        if car.sum() > 0:
            thre = np.percentile(p[car], 50)
            p[car] = p[car] > thre
    mask = (p > mask_thre) & (proj["proj_mask"] == 1)
    pts, lab = proj["proj_xyz"][mask], proj["proj_semantic"][mask]
    sky = lab == 10
    pts, lab = pts[~sky], lab[~sky]
    out = (lab == 0) & (pts[:, 2] < -3)
    return pts[~out], lab[~out]


@pytest.mark.parametrize("place_car", [False, True])
def test_apply_ray_drop_matches_restatement(place_car):
    rng = np.random.default_rng(5)
    H, W = 32, 128
    proj = dict(proj_range=rng.uniform(1, 60, (H, W)).astype(np.float32),
                proj_xyz=rng.uniform(-20, 20, (H, W, 3)).astype(np.float32),
                proj_semantic=rng.integers(0, 19, (H, W)).astype(np.float32),
                proj_mask=(rng.uniform(size=(H, W)) > 0.2).astype(np.float32))
    proj["proj_xyz"][..., 2] = rng.uniform(-6, 3, (H, W))
    logits = rng.normal(size=(2, H, W)).astype(np.float32)
    want_p, want_l = _ray_drop_np({k: v.copy() for k, v in proj.items()}, logits.astype(np.float64), 0.5, place_car)
    got_p, got_l = raydrop.apply_ray_drop({k: torch.from_numpy(v) for k, v in proj.items()}, torch.from_numpy(logits), 0.5, place_car)
    assert got_p.shape[0] > 500 and (want_l == 10).sum() == 0
    np.testing.assert_array_equal(got_l.numpy(), want_l.astype(np.int64))
    np.testing.assert_array_equal(got_p.numpy(), want_p)


def test_write_points_and_labels_kitti_layout(tmp_path):
    pts = np.arange(15, dtype=np.float64).reshape(5, 3)
    lab = np.array([1, 0, 13, 18, 4])
    raydrop.write_points_and_labels(3, tmp_path, torch.from_numpy(pts), torch.from_numpy(lab))
    b = np.fromfile(os.path.join(tmp_path, "velodyne", "000003.bin"), dtype=np.float32)
    l = np.fromfile(os.path.join(tmp_path, "labels", "000003.label"), dtype=np.uint32)
    np.testing.assert_array_equal(b.reshape(-1, 3), pts.astype(np.float32))
    np.testing.assert_array_equal(l, lab.astype(np.uint32))


def test_reads_a_checkpoint_written_by_the_reference():
    """tests/golden/ckpt_ref/checkpoint_1234.ckpt was written by the REFERENCE's internal/checkpoints.py:save_checkpoint from the
    reference Model's state_dict (tests/golden/make_golden.py:gen_checkpoint): the importer must read that file, recover the
    architecture from the tensor shapes and hand back exactly the parameters the reference model held."""
    from conftest import GOLDEN
    d = os.path.join(GOLDEN, "ckpt_ref")
    assert os.path.basename(ck.latest_checkpoint(d)) == "checkpoint_1234.ckpt"
    sd, step = ck.load_checkpoint(d)
    assert step == 1234
    mc = nconfig.workload("C1", 9)
    want = nweights.synth_state_dict(mc, seed=21, trained_like=True)
    keep, ignored = ck.split_state_dict(sd)
    assert ignored == []
    for k, v in want.items():  # (split_state_dict sets the encoder buffers aside: they are re-derived from the configuration)
        np.testing.assert_array_equal(sd[k], v, err_msg=k)
        assert k in keep or k.rsplit(".", 1)[-1] in ("offsets", "grid_sizes", "idx"), k
    assert "nerf_mlp.encoder.idx" in sd  # the buffer the reference registers and we re-derive
    got = ck.infer_model_config(keep, nconfig.ModelConfig(num_prop_samples=(), num_nerf_samples=64, num_levels=1))
    assert got.nerf_mlp.net_depth_viewdirs == 4 and got.nerf_mlp.net_width_viewdirs == 128
    assert got.nerf_mlp.grid_log2_hashmap_size == 9 and got.config.use_semantic and not got.config.use_intensity


# ---- GPU ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_reference_written_checkpoint_renders_like_its_parameters():
    from conftest import GOLDEN
    from nerflidar_hip import lidar as nlidar
    from nerflidar_hip.models import Model
    mc = nconfig.workload("C1", 9)
    base = nconfig.ModelConfig(num_prop_samples=(), num_nerf_samples=64, num_levels=1)
    m2, step, ignored = ck.model_from_checkpoint(os.path.join(GOLDEN, "ckpt_ref"), base=base)
    m1 = Model(mc, nweights.synth_state_dict(mc, seed=21, trained_like=True))
    b = nlidar.synthetic_sweep(width=16, seed=2)
    batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    r1, r2 = m1.render_rays(batch)[0], m2.render_rays(batch)[0]
    for k in ("depth", "rgb", "semantic"):
        assert torch.equal(r1[k], r2[k]), k

@pytest.mark.gpu
def test_model_from_checkpoint_renders_identically(tmp_path):
    from nerflidar_hip import lidar as nlidar
    from nerflidar_hip.models import Model
    mc = nconfig.workload("C2", 12)
    sd = nweights.synth_state_dict(mc, seed=1, trained_like=True)
    ck.save_checkpoint(tmp_path, sd, 1234)
    base = nconfig.ModelConfig(num_prop_samples=(64, 64), num_nerf_samples=128)
    m2, step, ignored = ck.model_from_checkpoint(tmp_path, base=base, precision=2)
    assert step == 1234 and ignored == []
    m1 = Model(mc, sd, precision=2)
    b = nlidar.synthetic_sweep(width=16, seed=2)
    batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    r1, r2 = m1.render_rays(batch)[0], m2.render_rays(batch)[0]
    for k in ("depth", "rgb", "semantic", "intensity"):
        assert torch.equal(r1[k], r2[k]), k


@pytest.mark.gpu
def test_render_lidar_driver_writes_reference_files(tmp_path):
    from nerflidar_hip import render_lidar
    unet = raydrop.UNet(6, 2, bilinear=True)
    torch.manual_seed(0)
    pth = os.path.join(tmp_path, "unet.pth")
    torch.save(unet.state_dict(), pth)
    rc = render_lidar.main(["--workload", "C1", "--log2-hashmap", "12", "--width", "64", "--sweeps", "2", "--render-dir", str(tmp_path),
                            "--raydrop-unet", pth, "--mask-thre", "0.3"])
    assert rc == 0
    d = os.path.join(tmp_path, "lidar_replay")
    for i in range(2):
        p = np.load(os.path.join(d, f"points_{i:04d}.npy"))
        s = np.load(os.path.join(d, f"points_semantic_{i:04d}.npy"))
        c = np.load(os.path.join(d, f"points_rgb_{i:04d}.npy"))
        assert p.shape == (32 * 64, 3) and s.shape == (32 * 64,) and c.shape == (32 * 64, 3)
        assert np.isfinite(p).all() and s.min() >= 0 and s.max() < 19
        b = np.fromfile(os.path.join(tmp_path, "raydrop", "velodyne", f"{i:06d}.bin"), dtype=np.float32)
        l = np.fromfile(os.path.join(tmp_path, "raydrop", "labels", f"{i:06d}.label"), dtype=np.uint32)
        assert b.size == 3 * l.size and l.size <= 32 * 64 and not (l == 10).any()


@pytest.mark.gpu
def test_render_sweep_points_follow_reference_formula():
    """points = (o + depth * d) / scale_factor, labels = argmax (render_lidar.py:142-156), against the oracle's post-step."""
    from nerflidar_hip import lidar as nlidar, render_lidar
    from nerflidar_hip.models import Model
    from oracle import nlr_oracle as orc
    mc = nconfig.workload("REF", 12)
    sd = nweights.synth_state_dict(mc, seed=4, trained_like=True)
    m = Model(mc, sd, precision=1)
    b = nlidar.synthetic_sweep(width=8, seed=4)
    res = render_lidar.render_sweep(m, b, 1.0 / 250.0)
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    pts, lab = orc.lidar_post(tb, {"depth": res["depth"].cpu(), "semantic": res["semantic"].cpu()}, 1.0 / 250.0)
    np.testing.assert_allclose(res["points"].cpu().numpy(), pts.numpy(), rtol=1e-6, atol=1e-5)
    np.testing.assert_array_equal(res["labels"].cpu().numpy(), lab.numpy())


@pytest.mark.gpu
def test_render_lidar_driver_dynamic_objects(tmp_path):
    from nerflidar_hip import render_lidar
    rc = render_lidar.main(["--workload", "REF", "--log2-hashmap", "12", "--width", "64", "--sweeps", "2", "--render-dir", str(tmp_path),
                            "--synthetic-tracks", "4"])
    assert rc == 0
    s0 = np.load(os.path.join(tmp_path, "lidar_replay", "points_semantic_0000.npy"))
    assert s0.shape == (32 * 64,) and ({13, 14, 15} & set(s0.tolist()))  # some ray ends on an object
