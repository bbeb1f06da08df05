"""A TRAINED, well-conditioned scene through the whole chain: train on the GPU with `training.training_step`
(`python -m nerflidar_hip.train_scene`) -> `checkpoint_<step>.ckpt` in the reference's format (tests/golden/ckpt_trained/) -> the
REFERENCE restores it with `internal/checkpoints.restore_checkpoint` and renders rays of a held-out sweep
(tests/golden/make_golden.py:gen_trained -> fwd_TRAINED_REFI.npz) -> here the same file goes through `checkpoints.model_from_checkpoint`
into the fused HIP path, and the gates are MAXIMA: depth max <= 1e-3, intensity max <= 1e-3, labels exact, three precisions, both paths.

This is the scene on which the explanation of the white-noise fixtures' parity tail (conditioning of white-noise tables
under a x1500 density gain, not a kernel defect) predicts a clean result - and is checked."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden
from nerflidar_hip import checkpoints as nckpt, config as nconfig, scene as nscene, lidar as nlidar
from oracle import nlr_oracle as orc

# two trained checkpoints: the shipped architecture + intensity head (2 x 256 view MLP, 32 final samples) and the benchmark
# architecture C2 (8 x 256, 128 final samples; also bench.py's `trained_scene` workload)
CASES = {"REFI": ("ckpt_trained", "fwd_TRAINED_REFI"), "C2": ("ckpt_trained_c2", "fwd_TRAINED_C2")}
T = torch.from_numpy


def _ckpt(case):
    return os.path.join(GOLDEN, CASES[case][0])


def _setup(case="REFI"):
    CKPT = _ckpt(case)
    g = golden(CASES[case][1])
    summ = json.load(open(os.path.join(CKPT, "train_summary.json")))["summary"]
    mc = nconfig.workload(summ["workload"], summ["log2_hashmap"])
    sd, step = nckpt.load_checkpoint(CKPT)
    assert step == int(g["ckpt_step"]) == summ["steps"]
    keep, ignored = nckpt.split_state_dict(sd)
    assert not ignored
    from nerflidar_hip import weights as nweights
    for prefix, cfg in nweights.mlp_names(mc):   # buffers a checkpoint import re-derives from the config (grid.py:137-142); the oracle reads them
        keep[f"{prefix}.encoder.offsets"], keep[f"{prefix}.encoder.grid_sizes"], _ = nweights.grid_layout(cfg)
    batch = {k[3:]: g[k] for k in g if k.startswith("in_")}
    batch["base_x"] = batch["base_y"] = batch["directions"]          # ZI/lidar_utils.py:17-18
    return g, mc, keep, batch


@pytest.mark.parametrize("case", list(CASES))
def test_fixture_rays_are_the_held_out_sweep(case):
    """The fixture's rays are rows `ray_index` of `lidar.synthetic_sweep(sweep_idx=100)` (viewdirs with the whole sweep's Frobenius
    norm), a sensor position outside the 64 the training batches draw from."""
    g, mc, sd, batch = _setup(case)
    full = nlidar.synthetic_sweep(width=int(g["width"]), seed=0, sweep_idx=int(g["sweep_idx"]))
    for k in ("origins", "directions", "viewdirs", "radii", "near", "far"):
        np.testing.assert_array_equal(full[k][g["ray_index"]], batch[k])
    train_pos = np.stack([nlidar.synthetic_sweep(width=2, seed=0, sweep_idx=i)["origins"][0] for i in range(64)])
    assert np.abs(train_pos - batch["origins"][0]).max(-1).min() > 1e-4


@pytest.mark.parametrize("case", list(CASES))
def test_checkpoint_infers_the_trained_architecture(case):
    g, mc, sd, batch = _setup(case)
    inferred = nckpt.infer_model_config(sd, base=mc)
    assert inferred.nerf_mlp.grid_log2_hashmap_size == int(g["log2_hashmap"]) and inferred.config.use_intensity
    assert inferred.level_samples() == mc.level_samples()


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_matches_reference_on_the_trained_scene(case):
    """The CPU oracle on the restored weights against the reference's own run: on a smooth field the two fp32 evaluations agree to
    1e-5 - two orders below the white-noise fixtures - which pins the oracle on a realistic scene."""
    g, mc, sd, batch = _setup(case)
    rend, hist = orc.model_forward(sd, mc, {k: T(np.ascontiguousarray(v)) for k, v in batch.items()})
    r = rend[-1]
    d = np.abs(r["depth"].numpy() - g["out_depth"])
    assert d.max() <= 2e-5, f"depth max {d.max():.2e}"
    assert np.abs(r["intensity"].numpy() - g["out_intensity"]).max() <= 2e-5
    np.testing.assert_array_equal(r["semantic"].numpy().argmax(-1), g["out_semantic"].argmax(-1))
    K = g["hist2_weights"].shape[0]
    assert np.abs(hist[-1]["weights"][:K].numpy() - g["hist2_weights"]).max() <= 2e-4


def test_inflated_hash_maps_evaluate_the_same_field():
    """`weights.inflate_hashmaps` (bench.py's trained-scene workload: the committed small-map checkpoint on full-size maps): the oracle
    renders the inflated parameters to the SAME numbers - tiling a hashed level and re-indexing a hashed level that becomes dense
    reproduce every look-up exactly (gridencoder.cu:66-84)."""
    from nerflidar_hip import weights as nweights
    g, mc, sd, batch = _setup("REFI")
    sub = {k: T(np.ascontiguousarray(v[:96])) for k, v in batch.items()}
    want, _ = orc.model_forward(sd, mc, sub)
    for lg in (15, 18):    # 15: all hashed levels stay hashed; 18: level 1 and 2 (33^3, 65^3 rows) become dense
        sdb, mcb = nweights.inflate_hashmaps(sd, mc, lg)
        assert mcb.nerf_mlp.grid_log2_hashmap_size == lg and sdb["nerf_mlp.encoder.embeddings"].shape[0] > sd["nerf_mlp.encoder.embeddings"].shape[0]
        got, _ = orc.model_forward(sdb, mcb, sub)
        for k in ("depth", "intensity", "semantic", "rgb"):
            np.testing.assert_array_equal(got[-1][k].numpy(), want[-1][k].numpy(), err_msg=f"log2 {lg} {k}")


@pytest.mark.parametrize("case", list(CASES))
def test_trained_scene_is_a_scene(case):
    """The reference's rendering of the checkpoint against the analytic ground truth: a trained field, not noise (median range error
    below half a metre on rays out to 70 m, > 95 % of the labels right)."""
    g, mc, sd, batch = _setup(case)
    gt = nscene.cast(T(batch["origins"]), T(batch["directions"]), nlidar.seeded_rotation(0), 1 / 250)
    err_m = np.abs(g["out_depth"] - gt["depth"].numpy()) * 250
    assert np.median(err_m) < 0.5, np.median(err_m)
    assert (g["out_semantic"].argmax(-1) == gt["semantic"].numpy()).mean() > 0.95
    assert np.abs(g["out_intensity"] - gt["intensity"].numpy()).mean() < 0.03


# ---- GPU ------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu_models():
    from nerflidar_hip import _lib
    return {(c, p): nckpt.model_from_checkpoint(_ckpt(c), base=_setup(c)[1], device="cuda:0", precision=p)[0]
            for c in CASES for p in (_lib.PREC_F32, _lib.PREC_MIXED, _lib.PREC_FAST)}


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("precision", [0, 1, 2])
@pytest.mark.parametrize("path", ["history", "render"])
def test_trained_scene_max_gates(gpu_models, case, precision, path):
    """MAXIMUM gates on every ray of the fixture (north_star: depth / intensity within 1e-3, labels bit-exact)."""
    g, mc, sd, batch = _setup(case)
    model = gpu_models[(case, precision)]
    b = {k: T(np.ascontiguousarray(v)).cuda() for k, v in batch.items()}
    if path == "history":
        rend, hist = model(False, b, train_frac=1.0, compute_extras=True)
        r = rend[-1]
        labels = r["semantic"].argmax(-1).cpu().numpy()
    else:
        r, _ = model.render_rays(b, scale_factor=1 / 250)
        labels = r["labels"].cpu().numpy()
    npy = lambda t: t.detach().cpu().numpy()
    d = np.abs(npy(r["depth"]) - g["out_depth"])
    i = np.abs(npy(r["intensity"]) - g["out_intensity"])
    s = np.abs(npy(r["semantic"]) - g["out_semantic"])
    msg = (f"precision {precision} {path}: depth L1 {d.mean():.2e} max {d.max():.2e}; intensity max {i.max():.2e}; semantic max {s.max():.2e}; "
           f"labels differ {(labels != g['out_semantic'].argmax(-1)).sum()}")
    print(msg)
    assert d.max() <= 1e-3, msg
    assert d.mean() <= 1e-4, msg
    assert i.max() <= 1e-3, msg
    assert s.max() <= 5e-3, msg
    np.testing.assert_array_equal(labels, g["out_semantic"].argmax(-1), err_msg=msg)
    assert np.abs(npy(r["acc"]) - g["out_acc"]).max() <= 1e-5
    rgb_tol = 1e-3 if precision == 0 else 2e-2     # bf16 view MLP in MIXED / FAST
    assert np.abs(npy(r["rgb"]) - g["out_rgb"]).max() <= rgb_tol, msg
    if path == "history":  # per-sample history of the first rays: the inverse-CDF stage the white-noise tail was traced to
        K = g["hist0_sdist"].shape[0]
        for lvl in range(mc.num_levels):
            ds = np.abs(npy(hist[lvl]["sdist"][:K]) - g[f"hist{lvl}_sdist"])
            dw = np.abs(npy(hist[lvl]["weights"][:K]) - g[f"hist{lvl}_weights"])
            assert ds.max() <= 1e-4, f"level {lvl} sdist max {ds.max():.2e}"
            assert dw.max() <= 2e-3, f"level {lvl} weights max {dw.max():.2e}"


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_trained_full_sweep_against_oracle(gpu_models, case):
    """The whole held-out 32 x 1024 sweep (32 768 rays) on the GPU against the CPU oracle on 4 096 of its rays, maxima again."""
    g, mc, sd, _ = _setup(case)
    full = nlidar.synthetic_sweep(width=1024, seed=0, sweep_idx=int(g["sweep_idx"]))
    idx = np.linspace(0, full["origins"].shape[0] - 1, 4096 if case == "REFI" else 1024).astype(np.int64)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    rend, _ = orc.model_forward(sd, mc, {k: T(np.ascontiguousarray(v[idx])) for k, v in full.items()})
    ref = rend[-1]
    r, _ = gpu_models[(case, 2)].render_rays({k: T(v).cuda() for k, v in full.items()}, scale_factor=1 / 250)
    d = np.abs(r["depth"].cpu().numpy()[idx] - ref["depth"].numpy())
    i = np.abs(r["intensity"].cpu().numpy()[idx] - ref["intensity"].numpy())
    lab = r["labels"].cpu().numpy()[idx] != ref["semantic"].numpy().argmax(-1)
    # label ties: a ray whose two best classes are closer than the arithmetic's resolution is not a mismatch of the path
    sr = np.sort(ref["semantic"].numpy(), -1)
    margin = sr[:, -1] - sr[:, -2]
    msg = f"depth L1 {d.mean():.2e} max {d.max():.2e} intensity max {i.max():.2e} labels {lab.sum()} (min margin of mismatches {margin[lab].min() if lab.any() else None})"
    print(msg)
    assert d.max() <= 1e-3 and i.max() <= 1e-3, msg
    assert not (lab & (margin > 1e-3)).any(), msg


@pytest.mark.gpu
@pytest.mark.parametrize("case,width", [("C2", 256), ("REFI", 67)])
def test_wave_order_of_the_encode_kernels_does_not_change_the_render(gpu_models, case, width):
    """Which samples share a wave of the fused cast + encode kernels (8 adjacent rays at one sample index, 8 consecutive samples of one ray,
    or the per-level decision the device takes from the coherence of adjacent rays: `NlrRenderCfg.shuffled_rays`, NLR_DBG_RAY_GROUPS) is a
    performance choice: every output is bit-identical.  width 67: 32 x 67 = 2 144 rays = 268 groups of 8 (REFI), and the 820-ray fixtures
    above (102.5 groups) take the default path through the padded last group."""
    from nerflidar_hip import _lib
    model = gpu_models[(case, 2)]
    b = nlidar.synthetic_sweep(width=width, seed=0, sweep_idx=100)
    if case == "REFI":   # a ray count that is not a multiple of 8
        b = {k: np.ascontiguousarray(v[:-3]) for k, v in b.items()}
    tb = {k: T(v).cuda() for k, v in b.items()}
    outs = {}
    try:
        for name, key, shuffled in (("auto", 0, False), ("rays", 1, False), ("samples", 2, False), ("hint", 0, True)):
            _lib.lib().nlr_debug_set(_lib.DBG_RAY_GROUPS, key)
            model.shuffled_rays = shuffled
            r, hist = model.render_rays(tb, scale_factor=1 / 250, want_history=True)
            outs[name] = {k: v.clone() for k, v in r.items()} | {f"w{l}": h["weights"].clone() for l, h in enumerate(hist)}
    finally:
        _lib.lib().nlr_debug_set(_lib.DBG_RAY_GROUPS, 0)
        model.shuffled_rays = False
    for name in ("rays", "samples", "hint"):
        for k, v in outs["auto"].items():
            assert torch.equal(v, outs[name][k]), (name, k)
