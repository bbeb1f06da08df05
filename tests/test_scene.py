"""CPU tests of the pieces around the trained-scene chain that need no GPU: the analytic scene's ray casting (`nerflidar_hip.scene`), the
training batch contract, the learning-rate schedule against the reference's own function, the checkpoint writer / reader round trip."""
import numpy as np
import torch

from conftest import golden
from nerflidar_hip import checkpoints as nckpt, lidar as nlidar, scene as nscene, training as ntrain


def test_scene_is_closed_and_every_ray_returns():
    b = nlidar.synthetic_sweep(width=128, seed=0, sweep_idx=5)
    gt = nscene.cast(torch.from_numpy(b["origins"]), torch.from_numpy(b["directions"]), nlidar.seeded_rotation(0), 1 / 250)
    d_m = gt["depth"].numpy() * 250
    assert np.isfinite(d_m).all() and d_m.min() > 2.0 and d_m.max() < 130.0            # between near (2 m) and the far walls
    assert set(np.unique(gt["semantic"].numpy())) <= {0, 1, 2, 3, 5, 8, 9, 13, 14} and len(np.unique(gt["semantic"].numpy())) >= 7
    assert 0.0 < float(gt["intensity"].min()) and float(gt["intensity"].max()) <= 1.0
    n = gt["normal"].numpy()
    np.testing.assert_allclose(np.abs(n).sum(-1), 1.0)                                    # axis-aligned faces
    # the hit point lies on the surface it names: ground hits at GROUND_Z in the sensor frame
    R = nlidar.seeded_rotation(0)
    p = (b["origins"].astype(np.float64) + gt["depth"].numpy()[:, None].astype(np.float64) * b["directions"]) * 250 @ R
    ground = n[:, 2] == 1
    np.testing.assert_allclose(p[ground & (np.abs(p[:, 2] - nscene.GROUND_Z) < 1.0), 2], nscene.GROUND_Z, atol=2e-3)
    assert (np.abs(p[:, 0]) <= nscene.WALL_X + 1e-2).all() and (np.abs(p[:, 1]) <= nscene.WALL_Y + 1e-2).all()


def test_scene_depth_is_the_ray_parameter_of_unnormalised_directions():
    """`depth` runs along `directions` as given (the renderer's t, ZI/render.py:217-252), whatever their length."""
    b = nlidar.synthetic_sweep(width=16, seed=0, sweep_idx=2)
    o, d = torch.from_numpy(b["origins"]), torch.from_numpy(b["directions"])
    a = nscene.cast(o, d, nlidar.seeded_rotation(0), 1 / 250)["depth"]
    c = nscene.cast(o, d * 2.0, nlidar.seeded_rotation(0), 1 / 250)["depth"]
    np.testing.assert_allclose(c.numpy() * 2.0, a.numpy(), rtol=1e-5)


def test_training_batch_follows_the_lidar_batch_contract():
    """`scene.random_lidar_rays` (built with torch on the training device) = `lidar.cast_lidar_ray_batch` (ZI/lidar_utils.py:8-33) on the
    same rays, key for key - Frobenius `viewdirs`, `base_x = base_y = directions` included - and is deterministic in (seed, step)."""
    r = nscene.random_lidar_rays(1000, 3, 7, "cpu", rot_seed=0)
    ref = nlidar.cast_lidar_ray_batch(r["origins"].double().numpy(), r["directions"].double().numpy(), 0.008, 2.0)
    assert set(r) == set(ref)
    for k in ref:
        np.testing.assert_allclose(r[k].numpy().reshape(ref[k].shape), ref[k], atol=1e-8, err_msg=k)
    r2 = nscene.random_lidar_rays(1000, 3, 7, "cpu", rot_seed=0)
    assert all(torch.equal(r[k], r2[k]) for k in r)
    assert not torch.equal(r["directions"], nscene.random_lidar_rays(1000, 3, 8, "cpu", rot_seed=0)["directions"])
    pos = np.stack([nlidar.synthetic_sweep(width=2, seed=0, sweep_idx=i)["origins"][0] for i in range(64)])
    assert all(np.abs(pos - o_).max(-1).min() < 1e-7 for o_ in r["origins"].numpy()[:50])   # origins are the first 64 sweep positions
    s = nscene.supervise(r)
    assert set(("rgb", "depth", "semantic", "intensity", "mask_rgb", "depth_mask", "sem_mask", "lidar_mask")) <= set(s)


def test_learning_rate_schedule_matches_the_reference():
    g = golden("fn_lr_schedule")
    for tag in ("default", "short", "nodelay"):
        lr_init, lr_final, max_steps, delay, mult = g[tag + "_kw"]
        got = [ntrain.learning_rate_decay(int(s_), lr_init, lr_final, int(max_steps), int(delay), mult) for s_ in g["steps"]]
        # (beyond max_steps the reference extrapolates the log-linear decay; the schedule here holds lr_final)
        inside = g["steps"] <= max_steps
        np.testing.assert_allclose(np.array(got)[inside], g[tag][inside], rtol=1e-12, atol=0, err_msg=tag)


def test_checkpoint_round_trip_keeps_every_tensor(tmp_path):
    """`save_checkpoint` writes what `load_checkpoint` / the reference's `restore_checkpoint` read: {'step', 'state_dict'} in
    `checkpoint_<step>.ckpt` (ZI/checkpoints.py:58-82), newest step wins."""
    sd = {"nerf_mlp.density_layer.0.weight": np.arange(12, dtype=np.float32).reshape(3, 4), "prop_mlp_0.encoder.embeddings": np.ones((8, 1), np.float32)}
    nckpt.save_checkpoint(tmp_path, sd, 10)
    p = nckpt.save_checkpoint(tmp_path, {k: v * 2 for k, v in sd.items()}, 200)
    assert p.endswith("checkpoint_200.ckpt") and nckpt.latest_checkpoint(tmp_path) == p
    got, step = nckpt.load_checkpoint(tmp_path)
    assert step == 200 and all(np.array_equal(got[k], sd[k] * 2) for k in sd)
    got10, step10 = nckpt.load_checkpoint(tmp_path, step=10)
    assert step10 == 10 and np.array_equal(got10["prop_mlp_0.encoder.embeddings"], sd["prop_mlp_0.encoder.embeddings"])
