"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the
reference-generated golden fixtures.  Tolerances (north_star: depth / intensity within 1e-3 of the
reference, semantic argmax bit-exact) are written next to each assert; most stages agree far tighter.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden
from oracle import nlr_oracle as orc
from nerflidar_hip import _lib, config as nconfig, lidar as nlidar, weights as nweights

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def _names(prefix):
    # (fwd_TRAINED_* belong to tests/test_trained_scene.py: their weights come from a checkpoint, not from a seed)
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")) if "TRAINED" not in p)


def cu(a):
    return T(np.ascontiguousarray(a)).to(DEV)


def npy(t):
    return t.detach().cpu().numpy()


_GATE_LOG = os.environ.get("NLR_GATE_LOG")


def _gate_distance(name, got, ref, key):
    """distance_mean / median / percentiles per ray (48-96 rays per fixture).  The percentiles interpolate the CDF: where it is flat
    (empty space between two surfaces) a 1e-6 change of a weight moves the crossing a long way on that ray, so they get a ray-count
    gate at 1e-2 (measured: at most 1 ray of a fixture beyond it, max 2.8e-2) instead of round 2's maximum of 2e-1."""
    if key in ("distance_mean", "distance_median"):
        gate(name, got, ref, 2e-4, 1e-2, thr=1e-3, frac=0.035)
    else:
        gate(name, got, ref, 1e-3, 1e-1, thr=1e-2, frac=0.035)


_LOOSE = 1.0  # x2 on mean / fraction gates for P_F32 (set by the forward tests): 16 levels up to resolution 524 288 = 2 mm cells of white
              # noise under the x1500 density gain; a 1-ulp coordinate difference moves a fine-level feature by 3e-2 of its amplitude


@pytest.fixture(autouse=True)
def _reset_looseness():
    global _LOOSE
    _LOOSE = 1.0
    yield
    _LOOSE = 1.0


def gate(name, got, ref, mean_tol, max_tol=None, thr=None, frac=0.0):
    mean_tol, frac = mean_tol * _LOOSE, frac * _LOOSE
    """mean |d| <= mean_tol; the FRACTION of elements with |d| > thr is <= frac (frac = 0: none); max |d| <= max_tol where a hard
    bound is meaningful.  Fractions replaced round 2's maxima that had been sized to pass (weights 5e-2, semantic 3e-2, percentiles
    2e-1): an element count beyond a tight threshold detects a defect a loose maximum does not (tests/test_fullsize_parity.py explains
    the tail).  NLR_GATE_LOG=<file> appends the measured statistics."""
    d = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).reshape(-1)
    if _GATE_LOG:
        with open(_GATE_LOG, "a") as f:
            f.write(f"{name} n={d.size} mean={d.mean():.3e} p99={np.percentile(d, 99):.3e} max={d.max():.3e} "
                    f"f>1e-4={np.mean(d > 1e-4):.4f} f>1e-3={np.mean(d > 1e-3):.4f} f>1e-2={np.mean(d > 1e-2):.4f}\n")
    msg = f"{name}: mean {d.mean():.3e} (<= {mean_tol})"
    ok = d.mean() <= mean_tol
    if thr is not None:
        f_ = float(np.mean(d > thr))
        msg += f", fraction > {thr}: {f_:.4f} (<= {frac})"
        ok = ok and f_ <= frac
    if max_tol is not None:
        msg += f", max {d.max():.3e} (<= {max_tol})"
        ok = ok and d.max() <= max_tol
    assert ok, msg


# ------------------------------------------------------------------------------------------------
# a-7 grid operator through gridencoder.GridEncoder / the C ABI
# ------------------------------------------------------------------------------------------------
def _points(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((n, 3)).astype(np.float32)
    x[:8] = np.array([[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0, 1, 0.25], [1, 0, 0], [0.999999, 0.5, 0.5],
                      [1e-7, 1e-7, 1e-7], [0.25, 0.75, 1.0]], np.float32)
    x[8:12, 0] = 1.0001   # out of range -> zeros (gridencoder.cu:110-135)
    x[12:16, 2] = -1e-6
    return x


@pytest.mark.parametrize("which", ["nerf", "prop0", "prop1"])
@pytest.mark.parametrize("log2_hashmap", [12, 21])
def test_grid_forward_bit_exact(which, log2_hashmap):
    mc = nconfig.workload("REF", log2_hashmap)
    cfg = {"nerf": mc.nerf_mlp, "prop0": mc.prop_cfg(0), "prop1": mc.prop_cfg(1)}[which]
    offsets, sizes, pls = nweights.grid_layout(cfg)
    from nerflidar_hip import synth
    table = synth.table_init(5, "t" + which, int(offsets[-1]), cfg.grid_level_dim, 1.0)
    x = _points(20000, 1)
    S, H = float(np.log2(pls)), cfg.grid_base_resolution
    ref, ref_dy = orc.grid_encode_c(x, table, offsets, S, H, want_dy_dx=True)
    L, Cc = len(offsets) - 1, cfg.grid_level_dim
    xd, td = cu(x), cu(table)
    off = np.ascontiguousarray(offsets, np.int32)
    for layout in (0, 1):
        out = torch.empty((L, len(x), Cc) if layout == 0 else (len(x), L * Cc), device=DEV)
        dy = torch.empty(len(x), L * 3 * Cc, device=DEV)
        rc = _lib.lib().nlr_grid_encode_forward(_lib.ptr(xd), _lib.ptr(td), 0, off.ctypes.data, _lib.ptr(out), len(x), 3, Cc,
                                                L, S, H, _lib.ptr(dy), 0, 0, 0, layout, None)
        _lib.check(rc)
        torch.cuda.synchronize()
        got = npy(out) if layout == 0 else npy(out).reshape(len(x), L, Cc).transpose(1, 0, 2)
        np.testing.assert_array_equal(got, ref)  # bit-exact: same fmaf order, host-computed level constants
        np.testing.assert_allclose(npy(dy), ref_dy, rtol=1e-5, atol=1e-4)
    assert (got[:, 8:16] == 0).all()
    # f16 tables (grid.py:43-44): values are the f16-rounded table, arithmetic still f32
    th = td.half()
    out = torch.empty(L, len(x), Cc, device=DEV)
    _lib.check(_lib.lib().nlr_grid_encode_forward(_lib.ptr(xd), _lib.ptr(th), 1, off.ctypes.data, _lib.ptr(out), len(x), 3, Cc,
                                                  L, S, H, None, 0, 0, 0, 0, None))
    ref16, _ = orc.grid_encode_c(x, npy(th.float()), offsets, S, H)
    np.testing.assert_array_equal(npy(out), ref16)


def test_gridencoder_module_and_backward():
    """Drop-in module (grid.py:96-174) incl. autograd against the CPU restatement of the backward."""
    from nerflidar_hip.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=6, level_dim=2, base_resolution=16, desired_resolution=512,
                      log2_hashmap_size=14, init_std=0.5).to(DEV)
    assert enc.output_dim == 12 and enc.grid_sizes.tolist() == [17, 33, 65, 129, 257, 513]
    x = cu(_points(5000, 2) * 2 - 1).requires_grad_(True)  # in [-1,1], bound=1
    y = enc(x, bound=1)
    g = torch.randn_like(y)
    y.backward(g)
    x01 = (npy(x) + 1) / 2
    S, H = float(np.log2(enc.per_level_scale)), enc.base_resolution
    ref, ref_dy = orc.grid_encode_c(x01.astype(np.float32), npy(enc.embeddings), npy(enc.offsets), S, H, want_dy_dx=True)
    np.testing.assert_array_equal(npy(y).reshape(len(x01), 6, 2).transpose(1, 0, 2), ref)
    gl = npy(g).reshape(len(x01), 6, 2).transpose(1, 0, 2)
    gt, gi = orc.grid_backward_c(gl, x01.astype(np.float32), npy(enc.offsets), enc.embeddings.shape[0], 2, S, H, dy_dx=ref_dy)
    np.testing.assert_allclose(npy(enc.embeddings.grad), gt, rtol=1e-4, atol=1e-5)  # atomics: order differs
    np.testing.assert_allclose(npy(x.grad), gi / 2, rtol=1e-4, atol=1e-4)          # d x01 / d x = 1/2
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        from nerflidar_hip.gridencoder import _backend
        _backend.grid_encode_forward(torch.zeros(4, 3), enc.embeddings, enc._offsets_host, torch.zeros(4, 12), 4, 3, 2, 6, 1.0, 16,
                                     None, 0, False, 0)


@pytest.mark.parametrize("C,layout", [(1, 0), (1, 1), (2, 1)])
def test_grid_backward_binned_scatter_against_oracle(C, layout):
    """`nlr_grid_encode_backward_ws` (round 4: levels beyond the LDS copy scattered through per-bucket bins in a workspace, accumulated per
    bucket in LDS) against the C restatement of kernel_grid_backward (gridencoder.cu:248-340): 2^17-entry hashed levels = 4 (C = 1) or 8
    (C = 2) buckets per level, ray-ordered points with runs of equal cells, corners / faces / out-of-range points, a batch that does not
    fill its last chunk, both gradient layouts; and the same call without a workspace (scattered atomics) gives the same table."""
    L, H, log2 = 6, 16, 17
    from nerflidar_hip import weights as nw
    offsets, sizes = nw.level_table(L, H, log2)[:2]
    B = (1 << 18) // C + 8192 * 2 + 37
    rng = np.random.default_rng(12)
    n_r = 256
    o = rng.random((n_r, 1, 3)) * 0.6 + 0.2
    d = rng.standard_normal((n_r, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = np.linspace(-0.25, 0.25, -(-B // n_r))[None, :, None]
    x = (o + d * t).reshape(-1, 3)[:B].astype(np.float32)
    x[:16] = _points(16, 5)
    grad = rng.standard_normal((L, B, C)).astype(np.float32)
    n_entries = int(offsets[-1])
    ref, _ = orc.grid_backward_c(grad, x, offsets, n_entries, C, 1.0, H)
    ref_abs, _ = orc.grid_backward_c(np.abs(grad), x, offsets, n_entries, C, 1.0, H)
    off = np.ascontiguousarray(offsets, np.int32)
    gl = grad if layout == 0 else np.ascontiguousarray(grad.transpose(1, 0, 2).reshape(B, L * C))
    xd, gd = cu(x), cu(gl)
    need = _lib.lib().nlr_grid_backward_workspace_bytes(B, C, L, 1.0, H, off.ctypes.data, 0, 0)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    got = {}
    for tag, w, nbytes in (("binned", ws, need), ("atomics", None, 0), ("short workspace", ws, need - 1)):
        gt = torch.zeros(n_entries, C, device=DEV)
        _lib.check(_lib.lib().nlr_grid_encode_backward_ws(_lib.ptr(gd), _lib.ptr(xd), off.ctypes.data, _lib.ptr(gt), B, 3, C, L, 1.0, H, None, None,
                                                          0, 0, 0, layout, _lib.ptr(w), nbytes, None))
        torch.cuda.synchronize()
        got[tag] = npy(gt)
        err = np.abs(got[tag] - ref)
        assert (err <= 2e-6 * ref_abs + 1e-6).all(), f"{tag}: max err / sum|terms| {np.max(err / (ref_abs + 1e-3)):.3e}"
    assert np.abs(ref[int(offsets[3]):]).max() > 0.5          # the hashed levels (the binned ones) really received the batch
    assert np.abs(got["binned"] - got["atomics"]).max() <= 1e-3 * np.abs(ref).max()


@pytest.mark.parametrize("C,order,log2", [(1, "ray", 14), (4, "ray", 14), (4, "random", 14), (2, "random", 14), (4, "ray", 19), (4, "ray-one-corner", 14)])
def test_grid_backward_lds_path_against_oracle(C, order, log2):
    """`nlr_grid_bwd_lds_kernel` (dense levels accumulated in an LDS copy of the table, taken when B * C >= 2^18: every real training
    step) against the C restatement of kernel_grid_backward (gridencoder.cu:248-340).  Ray-ordered points (runs of equal cells, the
    run-length aggregation of the non-LDS levels) and random points (the large-B path had only been compared with itself).
    C = 4 leaves through `nlr_grid_bwd_xpair_kernel` (both x-corners of a cell edge per atomic instruction; hashed levels at 2^14 entries,
    dense 33^3 / 65^3 levels beyond the LDS copy at 2^19) and, with NLR_DBG_NO_XPAIR_SCATTER, through the one-corner kernel of round 3."""
    L, H = 6, 16
    one_corner = order == "ray-one-corner"
    order = "ray" if one_corner else order
    from nerflidar_hip import weights as nw
    offsets, sizes = nw.level_table(L, H, log2)[:2]
    B = (1 << 18) // C + 4096 + 37           # B * C >= 2^18, not a multiple of anything
    rng = np.random.default_rng(11)
    if order == "ray":                       # 512 rays of B/512 consecutive points each
        n_r = 512
        o = rng.random((n_r, 1, 3)) * 0.6 + 0.2
        d = rng.standard_normal((n_r, 1, 3))
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        t = np.linspace(-0.2, 0.2, -(-B // n_r))[None, :, None]
        x = (o + d * t).reshape(-1, 3)[:B].astype(np.float32)
    else:
        x = rng.random((B, 3)).astype(np.float32)
    x[:16] = _points(16, 5)                  # corners, faces and out-of-range points
    grad = rng.standard_normal((L, B, C)).astype(np.float32)
    n_entries = int(offsets[-1])
    ref, _ = orc.grid_backward_c(grad, x, offsets, n_entries, C, 1.0, H)
    off = np.ascontiguousarray(offsets, np.int32)
    gt = torch.zeros(n_entries, C, device=DEV)
    xd, gd = cu(x), cu(grad)
    try:
        _lib.lib().nlr_debug_set(_lib.DBG_NO_XPAIR_SCATTER, int(one_corner))
        _lib.check(_lib.lib().nlr_grid_encode_backward(_lib.ptr(gd), _lib.ptr(xd), off.ctypes.data, _lib.ptr(gt), B, 3, C, L, 1.0, H, None, None,
                                                       0, 0, 0, 0, None))
        torch.cuda.synchronize()
    finally:
        _lib.lib().nlr_debug_set(_lib.DBG_NO_XPAIR_SCATTER, 0)
    got = npy(gt)
    # float atomics (LDS and global) add in a different order than the oracle's sequential loop: tolerance relative to the sum of
    # |contributions| of a cell, which for the hot cells of level 0 is thousands of terms
    ref_abs, _ = orc.grid_backward_c(np.abs(grad), x, offsets, n_entries, C, 1.0, H)
    err = np.abs(got - ref)
    assert (err <= 2e-6 * ref_abs + 1e-6).all(), f"max err / sum|terms| {np.max(err / (ref_abs + 1e-3)):.3e}"
    assert np.abs(ref[: int(offsets[1])]).max() > 1.0  # level 0 (17^3 cells, the LDS level) really received the batch
    # B * C beyond the 32-bit lane index is refused, not wrapped
    rc = _lib.lib().nlr_grid_encode_backward(_lib.ptr(gd), _lib.ptr(xd), off.ctypes.data, _lib.ptr(gt), (1 << 30) + 1, 3, 4, L, 1.0, H, None,
                                             None, 0, 0, 0, 0, None) if C == 4 else -1
    assert rc != 0


@pytest.mark.parametrize("which,log2_hashmap", [("nerf", 14), ("prop1", 12), ("prop0", 21)])
def test_grad_total_variation_matches_oracle(which, log2_hashmap):
    """`nlr_grad_total_variation` (gridencoder.h:15, cu:506-645) against the C restatement, through the C ABI and through
    `GridEncoder.grad_total_variation` (grid.py:176-198)."""
    mc = nconfig.workload("REF", log2_hashmap)
    cfg = {"nerf": mc.nerf_mlp, "prop0": mc.prop_cfg(0), "prop1": mc.prop_cfg(1)}[which]
    offsets, sizes, pls = nweights.grid_layout(cfg)
    from nerflidar_hip import synth
    Cc = cfg.grid_level_dim
    table = synth.table_init(9, "tv" + which, int(offsets[-1]), Cc, 1.0)
    x = _points(20000, 3)
    S, H = float(np.log2(pls)), cfg.grid_base_resolution
    g0 = (np.random.default_rng(2).standard_normal(table.shape) * 1e-3).astype(np.float32)
    ref = orc.grid_tv_c(x, table, g0, offsets, 1e-2, S, H)
    off = np.ascontiguousarray(offsets, np.int32)
    xd, td, gd = cu(x), cu(table), cu(g0)
    _lib.check(_lib.lib().nlr_grad_total_variation(_lib.ptr(xd), _lib.ptr(td), _lib.ptr(gd), off.ctypes.data, 1e-2, len(x), 3, Cc,
                                                   len(offsets) - 1, S, H, 0, 0, None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(npy(gd), ref, rtol=2e-4, atol=2e-6)   # float atomics + v_rsq_f32 for 1/sqrt
    assert np.abs(ref - g0).max() > 1e-3
    rc = _lib.lib().nlr_grad_total_variation(_lib.ptr(xd), _lib.ptr(td), _lib.ptr(gd), off.ctypes.data, 1e-2, len(x), 2, Cc,
                                             len(offsets) - 1, S, H, 0, 0, None)
    assert rc == -1 and b"D = 3" in _lib.lib().nlr_last_error()


def test_gridencoder_total_variation_method():
    from nerflidar_hip.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=6, level_dim=2, base_resolution=16, desired_resolution=512, log2_hashmap_size=14,
                      init_std=0.5).to(DEV)
    with pytest.raises(ValueError, match="grad is None"):
        enc.grad_total_variation(1e-3)
    pts = cu(_points(4000, 4) * 2 - 1)
    enc(pts).sum().backward()
    g0 = npy(enc.embeddings.grad).copy()
    enc.grad_total_variation(1e-3, inputs=pts, bound=1)
    x01 = ((npy(pts) + 1) / 2).astype(np.float32)
    ref = orc.grid_tv_c(x01, npy(enc.embeddings), g0, npy(enc.offsets), 1e-3, float(np.log2(enc.per_level_scale)), enc.base_resolution)
    np.testing.assert_allclose(npy(enc.embeddings.grad), ref, rtol=2e-4, atol=2e-6)
    enc.grad_total_variation(1e-3, B=1000)   # random cells (grid.py:188-190): runs, changes the gradient, stays finite
    g2 = npy(enc.embeddings.grad)
    assert np.isfinite(g2).all() and (g2 != ref).any()


# ------------------------------------------------------------------------------------------------
# a-2 / a-3 / a-4 resampling
# ------------------------------------------------------------------------------------------------
def _resample(prev_t, prev_w, dilation, S, near, far, jitter=None):
    n = prev_t.shape[0]
    npv = 0 if prev_w is None else prev_w.shape[1]
    sd = torch.empty(n, S + 1, device=DEV)
    td = torch.empty(n, S + 1, device=DEV)
    pt = cu(prev_t) if prev_w is not None else None
    pw = cu(prev_w) if prev_w is not None else None
    nr, fr = cu(near), cu(far)
    jt = cu(jitter) if jitter is not None else None
    rc = _lib.lib().nlr_resample_level(_lib.ptr(pt), _lib.ptr(pw), npv, float(dilation), 1.0, 0.0, S, _lib.ptr(jt), _lib.ptr(nr),
                                       _lib.ptr(fr), -1.5, n, _lib.ptr(sd), _lib.ptr(td), None)
    _lib.check(rc)
    torch.cuda.synchronize()
    return npy(sd), npy(td)


@pytest.mark.parametrize("name", _names("fn_max_dilate"))
@pytest.mark.parametrize("S", [32, 64, 128])
def test_resample_with_dilation(name, S):
    g = golden(name)
    t, w, d = T(g["t"]), T(g["w"]), float(g["dilation"])
    n = t.shape[0]
    near, far = np.full((n,), 0.008, np.float32), np.full((n,), 2.0, np.float32)
    sd, td = _resample(g["t"], g["w"], d, S, near, far)
    # reference chain: max_dilate_weights (golden) -> trim -> logits -> sample_intervals -> s_to_t (oracle, pinned)
    tdil, wdil = T(g["t_dilate"])[..., 1:-1], T(g["w_dilate"])[..., 1:-1]
    logits = torch.where(tdil[..., 1:] > tdil[..., :-1], torch.log(wdil), torch.full_like(wdil, -torch.inf))
    ref_s = orc.sample_intervals(tdil, logits, S, (0., 1.))
    _, s_to_t = orc.construct_ray_warps(T(near)[:, None], T(far)[:, None], -1.5)
    # inverse-CDF interpolation divides by bin mass: 1e-7 differences in the CDF are amplified in low-mass bins
    assert np.abs(sd - ref_s.numpy()).mean() <= 1e-6
    np.testing.assert_allclose(sd, ref_s.numpy(), atol=1e-4, rtol=0)
    np.testing.assert_allclose(td, s_to_t(ref_s).numpy(), atol=1e-4, rtol=1e-4)
    assert (np.diff(sd, axis=-1) >= 0).all() and sd.min() >= 0 and sd.max() <= 1


@pytest.mark.parametrize("n_prev,S", [(1, 16), (2, 8), (37, 48), (64, 64), (100, 32), (190, 64), (256, 128)])
def test_resample_dilation_ties_and_odd_sizes(n_prev, S):
    """The dilation stage ranks the merged fenceposts by binary search and takes window maxima from a range-max table:
    exercise ties (repeated fenceposts = zero-width bins, fenceposts exactly `dilation` apart, values pinned at 0 and 1),
    sizes that are not powers of two, and windows that span every interval, against the pinned oracle chain."""
    rng = np.random.default_rng(100 + n_prev)
    rows, d = 96, 0.03125  # a power of two: t_j - d == t_i and t_j + d == t_i happen exactly below
    grid = np.arange(0, 65, dtype=np.float32) / 64  # multiples of 1/64 = d/2, exact in float
    t = np.sort(rng.choice(grid, size=(rows, n_prev + 1), replace=True), axis=-1).astype(np.float32)
    t[: rows // 3] = np.sort(rng.random((rows // 3, n_prev + 1)).astype(np.float32), axis=-1)  # generic rows
    t[0, 0], t[0, -1] = 0.0, 1.0
    t[1] = np.linspace(0.4, 0.4 + 1e-6, n_prev + 1, dtype=np.float32)  # everything inside one dilation window
    w = rng.random((rows, n_prev)).astype(np.float32) + 1e-3
    w[2] = 0.0
    w[2, n_prev // 2] = 1.0  # one occupied bin
    w = w / w.sum(-1, keepdims=True)
    near, far = np.full((rows,), 0.008, np.float32), np.full((rows,), 2.0, np.float32)
    sd, td = _resample(t, w, d, S, near, far)
    tdil, wdil = orc.max_dilate_weights(T(t), T(w), d, (0., 1.), renormalize=True)
    tdil, wdil = tdil[..., 1:-1], wdil[..., 1:-1]
    if n_prev == 1:  # 3n-2 = 1 bin left after the trim
        assert tdil.shape[-1] == 2
    logits = torch.where(tdil[..., 1:] > tdil[..., :-1], torch.log(wdil), torch.full_like(wdil, -torch.inf))
    ref_s = orc.sample_intervals(tdil, logits, S, (0., 1.))
    ok = torch.isfinite(ref_s).all(-1).numpy()  # rows whose trimmed step function has no mass are NaN in the reference too
    assert ok.sum() >= rows // 2
    assert np.abs(sd[ok] - ref_s.numpy()[ok]).mean() <= 2e-6
    np.testing.assert_allclose(sd[ok], ref_s.numpy()[ok], atol=2e-4, rtol=0)
    assert (np.diff(sd[ok], axis=-1) >= 0).all()


@pytest.mark.parametrize("name", _names("fn_sample_intervals"))
def test_resample_plain(name):
    """No dilation: weights -> logits exactly as models.py:352-355 (fixtures carry log-weights)."""
    g = golden(name)
    S = g["sdist"].shape[-1] - 1
    n = g["t"].shape[0]
    near, far = np.full((n,), 0.05, np.float32), np.full((n,), 3.0, np.float32)
    if g["t"].shape[1] == 2:
        sd, td = _resample(g["t"], None, 0.0, S, near, far)
    else:
        w = np.exp(g["logits"]).astype(np.float32)  # exp(log w) is not bit-identical to w: 1e-6 tolerance
        sd, td = _resample(g["t"], w, 0.0, S, near, far)
    np.testing.assert_allclose(sd, g["sdist"], atol=2e-5, rtol=0)


def test_resample_random_jitter():
    """rand=True path (stepfun.py:211-216) with caller-provided uniform draws."""
    g = golden("fn_sample_intervals_s0_n64")
    n = g["t"].shape[0]
    u = np.random.default_rng(3).random((n, 1)).astype(np.float32)
    w = np.exp(g["logits"]).astype(np.float32)
    near, far = np.full((n,), 0.05, np.float32), np.full((n,), 3.0, np.float32)
    sd, _ = _resample(g["t"], w, 0.0, 64, near, far, jitter=u[:, 0])
    ref = orc.sample_intervals(T(g["t"]), T(g["logits"]), 64, (0., 1.), rand_u=T(u))
    np.testing.assert_allclose(sd, ref.numpy(), atol=2e-5, rtol=0)


# ------------------------------------------------------------------------------------------------
# a-13 / a-14 / a-16 compositing
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", _names("fn_composite"))
def test_composite(name):
    g = golden(name)
    n, S = g["density"].shape
    K = g["sem"].shape[-1]
    ins = {k: cu(g[k]) for k in ("density", "tdist", "dirs", "intensity", "far")}
    ins["rgbs"] = cu(np.ascontiguousarray(g["rgbs"].transpose(2, 0, 1)))  # channel-major [3,N,S]
    ins["sem"] = cu(np.ascontiguousarray(g["sem"].transpose(2, 0, 1)))    # class-major [K,N,S]
    origins = cu(np.zeros((n, 3), np.float32) + 0.25)
    out = _lib.NlrOut()
    res = {k: torch.empty(n, *sh, device=DEV) for k, sh in dict(rgb=(3,), depth=(), semantic=(K,), intensity=(), acc=(),
                                                                 distance_mean=(), distance_median=(), distance_percentile_5=(),
                                                                 distance_percentile_95=(), points=(3,)).items()}
    res["labels"] = torch.empty(n, dtype=torch.int32, device=DEV)
    for k, t in res.items():
        setattr(out, k, t.data_ptr())
    wts = torch.empty(n, S, device=DEV)
    rc = _lib.lib().nlr_composite_level(_lib.ptr(ins["density"]), _lib.ptr(ins["tdist"]), _lib.ptr(ins["dirs"]), _lib.ptr(ins["rgbs"]),
                                        _lib.ptr(ins["sem"]), _lib.ptr(ins["intensity"]), _lib.ptr(ins["far"]), _lib.ptr(origins), n, S, K,
                                        int("opaque" in name), 1.0, 1, 0.004, _lib.ptr(wts), C.byref(out), None, None)
    _lib.check(rc)
    torch.cuda.synchronize()
    np.testing.assert_allclose(npy(wts), g["weights"], atol=2e-6, rtol=1e-5)
    for k in ("rgb", "depth", "semantic", "intensity", "acc", "distance_mean", "distance_median", "distance_percentile_5",
              "distance_percentile_95"):
        np.testing.assert_allclose(npy(res[k]), g["out_" + k], atol=5e-6, rtol=2e-5, err_msg=k)
    np.testing.assert_array_equal(npy(res["labels"]), g["out_semantic"].argmax(-1))  # render_lidar.py:158-159
    pts = (npy(origins) + g["out_depth"][:, None] * g["dirs"]) / 0.004
    np.testing.assert_allclose(npy(res["points"]), pts, rtol=1e-5, atol=1e-3)


# ------------------------------------------------------------------------------------------------
# model-level: MLP rows and whole forward against reference-generated fixtures
# ------------------------------------------------------------------------------------------------
def _model(g, precision):
    from nerflidar_hip.models import Model
    lg = int(g["log2_hashmap"])
    mc = nconfig.workload(str(g["workload"]), None if lg < 0 else lg)
    sd = nweights.synth_state_dict(mc, seed=int(g["seed"]), trained_like=bool(g["trained_like"]))
    return mc, sd, Model(mc, sd, device=DEV, precision=precision)


@pytest.mark.parametrize("name", _names("mlp_"))
@pytest.mark.parametrize("precision", [_lib.PREC_F32, _lib.PREC_MIXED, _lib.PREC_FAST])
def test_mlp_level(name, precision):
    """Rows a-5..a-12: cast + contract + encode + MLP on the reference's own gaussians' tdist."""
    g = golden(name)
    fwd = golden("fwd_" + name[4:])
    mc, sd, model = _model(g, precision)
    KM = g["means"].shape[0]
    S = mc.num_nerf_samples
    batch_np = nlidar.synthetic_sweep(width=int(fwd["width"]), seed=int(fwd["seed"]), beams=list(fwd["beams"]))
    rays = _lib.NlrRays()
    keep = {k: cu(batch_np[k][:KM]) for k in ("origins", "directions", "viewdirs", "radii", "near", "far", "base_x", "base_y")}
    for k, t in keep.items():
        setattr(rays, k, t.data_ptr())
    tdist = cu(fwd["hist%d_tdist" % (mc.num_levels - 1)][:KM])
    F = mc.nerf_mlp.grid_num_levels * mc.nerf_mlp.grid_level_dim
    K = mc.nerf_mlp.class_num
    feat = torch.empty(KM * S, F, device=DEV)
    dens = torch.empty(KM, S, device=DEV)
    rgb = torch.empty(3, KM, S, device=DEV)   # channel-major, as the library writes per-sample heads
    sem = torch.empty(K, KM, S, device=DEV)   # class-major
    inten = torch.empty(KM, S, device=DEV) if mc.config.use_intensity else None
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    rc = _lib.lib().nlr_mlp_level(model._handle, mc.num_levels - 1, C.byref(rays), _lib.ptr(tdist), KM, 7, 3, None, _lib.ptr(feat),
                                  _lib.ptr(dens), _lib.ptr(rgb), _lib.ptr(sem), _lib.ptr(inten), _lib.ptr(ws), ws.numel(), None)
    _lib.check(rc)
    torch.cuda.synchronize()
    # features against the oracle's encode_features on the reference's means/stds
    enc = orc.make_encoders(sd, mc)["nerf_mlp"]
    ref_feat = orc.encode_features(enc, T(g["means"]), T(g["stds"])).numpy().reshape(KM * S, F)
    # fine levels (resolution 8192): a 1-ulp difference of a contracted coordinate (libm sqrt/pow, FMA use in
    # the basis product) moves the cell fraction by 8191 * 6e-8 = 5e-4, i.e. features by up to ~1e-4
    np.testing.assert_allclose(npy(feat), ref_feat, atol=2e-4, rtol=1e-4)
    # density against the reference: the pre-activation carries the feature error above times the x1500 "trained-like" gain of
    # density_layer.2 row 0, hence the loose gate here ...
    np.testing.assert_allclose(npy(dens), g["density"], atol=5e-2, rtol=2e-3)
    gate("density_level", npy(dens), g["density"], 5e-3, thr=1e-2, frac=0.05)
    # ... and the tight one on the MLP arithmetic itself: the trunk in float64 on the features the GPU produced, tolerance relative
    # to the gain sum_j |W2[0,j] h_j| of the raw density (split-bf16: ~2^-16 per product; exact-f32 MFMA: f32 rounding)
    f64 = npy(feat).astype(np.float64)
    W1, b1 = sd["nerf_mlp.density_layer.0.weight"].astype(np.float64), sd["nerf_mlp.density_layer.0.bias"].astype(np.float64)
    W2, b2 = sd["nerf_mlp.density_layer.2.weight"].astype(np.float64), sd["nerf_mlp.density_layer.2.bias"].astype(np.float64)
    hid = np.maximum(f64 @ W1.T + b1, 0.0)
    raw = hid @ W2[0] + b2[0] + mc.nerf_mlp.density_bias
    gain = np.abs(hid) @ np.abs(W2[0]) + np.abs(f64) @ np.abs(W1.T) @ np.abs(W2[0]) + 1.0
    ref_d = np.where(raw > 20, raw, np.log1p(np.exp(np.minimum(raw, 20))))
    rel = 3e-5 if precision == _lib.PREC_FAST else 2e-6
    err = np.abs(npy(dens).reshape(-1).astype(np.float64) - ref_d)
    assert (err <= rel * gain + 1e-6).all(), f"density vs f64 trunk on the GPU's features: max err/gain {np.max(err / gain):.3e} (allowed {rel})"
    np.testing.assert_allclose(npy(sem.permute(1, 2, 0)), g["semantic"], atol=2e-3, rtol=1e-3)
    if inten is not None:
        np.testing.assert_allclose(npy(inten), g["intensity"][..., 0], atol=1e-3, rtol=1e-3)
    rgb_tol = 1e-4 if precision == _lib.PREC_F32 else 2e-2  # bf16 view MLP (8 bits of mantissa per layer)
    np.testing.assert_allclose(npy(rgb.permute(1, 2, 0)), g["rgb"], atol=rgb_tol, rtol=0)
    # proposal network of level 0 on the same gaussians: same weights, a model view whose proposal level has the
    # NerfMLP sample count (weights do not depend on the sample count)
    import dataclasses
    from nerflidar_hip.models import Model
    mcp = dataclasses.replace(mc, num_prop_samples=tuple(S for _ in mc.num_prop_samples))
    pmodel = Model(mcp, sd, device=DEV, precision=precision)
    pd = torch.empty(KM, S, device=DEV)
    rc = _lib.lib().nlr_mlp_level(pmodel._handle, 0, C.byref(rays), _lib.ptr(tdist), KM, 7, 3, None, None, _lib.ptr(pd), None, None,
                                  None, None, 0, None)
    _lib.check(rc)
    torch.cuda.synchronize()
    np.testing.assert_allclose(npy(pd), g["prop_density"], atol=5e-2, rtol=2e-3)


@pytest.mark.parametrize("name", _names("fwd_"))
@pytest.mark.parametrize("precision", [_lib.PREC_F32, _lib.PREC_MIXED, _lib.PREC_FAST])
def test_model_forward(name, precision):
    """Whole Model.forward (rows a-1..a-16) against the reference run, via the drop-in `Model` class."""
    global _LOOSE
    _LOOSE = 2.0 if "P_F32" in name else 1.0
    g = golden(name)
    mc, sd, model = _model(g, precision)
    batch_np = nlidar.synthetic_sweep(width=int(g["width"]), seed=int(g["seed"]), beams=list(g["beams"]))
    batch = {k: cu(v) for k, v in batch_np.items()}
    rend, hist = model(False, batch, train_frac=1.0, compute_extras=True)
    r = rend[-1]
    K = g["hist0_sdist"].shape[0]
    # Error growth along the chain: 1-ulp libm differences in tdist -> cell-fraction differences at the fine grid
    # levels -> density (x1500 gain at "surfaces") -> weights -> next level's samples.  Gates therefore are the
    # north_star's: depth L1 (mean |d|) <= 1e-3, intensity <= 1e-3, semantic argmax bit-exact; per-sample
    # history is checked on its mean error and a loose max.
    for lvl in range(mc.num_levels):
        # (measured over all fixtures x precisions, r03: sdist max 7.9e-5, tdist 1.3e-4, weights 1.7e-3 with 0.07 % beyond 1e-3)
        gate(f"sdist{lvl}", npy(hist[lvl]["sdist"][:K]), g[f"hist{lvl}_sdist"], 1e-5, 1e-3, thr=1e-4, frac=0.002)
        gate(f"tdist{lvl}", npy(hist[lvl]["tdist"][:K]), g[f"hist{lvl}_tdist"], 1e-5, 1e-3, thr=1e-4, frac=0.004)
        gate(f"weights{lvl}", npy(hist[lvl]["weights"][:K]), g[f"hist{lvl}_weights"], 2e-5, 1e-2, thr=1e-3, frac=0.004)
        gate(f"depth{lvl}", npy(rend[lvl]["depth"]), g[f"lvl{lvl}_depth"], 2e-4, 1e-2, thr=1e-3, frac=0.035)
        if f"lvl{lvl}_rgb" in g:  # the whole rendering dict of a proposal level (ZI/models.py:514-531)
            gate(f"rgb{lvl}", npy(rend[lvl]["rgb"]), g[f"lvl{lvl}_rgb"], 1e-6, 1e-5)
            gate(f"acc{lvl}", npy(rend[lvl]["acc"]), g[f"lvl{lvl}_acc"], 1e-6, 1e-5)
            for k in ("distance_mean", "distance_median", "distance_percentile_5", "distance_percentile_95"):
                _gate_distance(f"{k}{lvl}", npy(rend[lvl][k]), g[f"lvl{lvl}_{k}"], k)
    gate("depth", npy(r["depth"]), g["out_depth"], 2e-4, 1e-2, thr=1e-3, frac=0.035)   # depth L1 within 1e-3 of the reference (measured <= 4e-5)
    assert np.percentile(np.abs(npy(r["depth"]) - g["out_depth"]), 95) <= 1e-3
    # percentiles interpolate the CDF: where it is flat (empty space between two surfaces) a 1e-6 change of a weight
    # moves the crossing point a long way, so only the mean is held tight
    for k in ("distance_mean", "distance_median", "distance_percentile_5", "distance_percentile_95"):
        _gate_distance(k, npy(r[k]), g["out_" + k], k)
    gate("acc", npy(r["acc"]), g["out_acc"], 1e-6, 1e-5)
    if "out_intensity" in g:
        gate("intensity", npy(r["intensity"]), g["out_intensity"], 1e-4, 1e-3)  # intensity within 1e-3
    if "out_semantic" in g:
        # measured: max 3.5e-3, 0.4 % beyond 1e-3 (P_F32: 1.4 %, max 2.1e-3, see _LOOSE)
        gate("semantic", npy(r["semantic"]), g["out_semantic"], 1e-4, 1e-2, thr=1e-3, frac=0.01)
        np.testing.assert_array_equal(npy(r["semantic"]).argmax(-1), g["out_semantic"].argmax(-1))  # bit-exact labels
    if precision == _lib.PREC_F32:
        gate("rgb", npy(r["rgb"]), g["out_rgb"], 2e-4, 5e-3, thr=1e-3, frac=0.02)
    else:
        gate("rgb", npy(r["rgb"]), g["out_rgb"], 2e-3, 2e-2)  # bf16 view MLP: 8 mantissa bits per layer, 8 layers (measured max 9.7e-3)


@pytest.mark.parametrize("name", _names("fwd_"))
def test_render_path_compositing_mode(name):
    """The render path proper (no ray_history: NLR_PREC_FAST, the MLP kernel composites inside 32-sample segments and
    nlr_composite_kernel combines the segment records) against the reference run, at the gates of test_model_forward; and
    against the ray_history path of the same library (same weights bit for bit, value sums to 2e-6)."""
    global _LOOSE
    _LOOSE = 2.0 if "P_F32" in name else 1.0
    g = golden(name)
    mc, sd, model = _model(g, _lib.PREC_FAST)
    batch_np = nlidar.synthetic_sweep(width=int(g["width"]), seed=int(g["seed"]), beams=list(g["beams"]))
    batch = {k: cu(v) for k, v in batch_np.items()}
    r, _ = model.render_rays(batch, scale_factor=1 / 250)
    ru, _ = model.render_rays(batch, scale_factor=1 / 250, want_history=True)

    gate("depth", npy(r["depth"]), g["out_depth"], 2e-4, 1e-2, thr=1e-3, frac=0.035)
    assert np.percentile(np.abs(npy(r["depth"]) - g["out_depth"]), 95) <= 1e-3
    gate("acc", npy(r["acc"]), g["out_acc"], 1e-6, 1e-5)
    if "out_intensity" in g:
        gate("intensity", npy(r["intensity"]), g["out_intensity"], 1e-4, 1e-3)
    if "out_semantic" in g:
        gate("semantic", npy(r["semantic"]), g["out_semantic"], 1e-4, 1e-2, thr=1e-3, frac=0.01)
        np.testing.assert_array_equal(npy(r["labels"]), g["out_semantic"].argmax(-1))  # bit-exact labels
    gate("rgb", npy(r["rgb"]), g["out_rgb"], 2e-3, 2e-2)
    for k in ("depth", "acc", "distance_median", "points"):
        np.testing.assert_array_equal(npy(r[k]), npy(ru[k]))
    for k in ("rgb", "semantic", "intensity"):
        if k in r:
            np.testing.assert_allclose(npy(r[k]), npy(ru[k]), rtol=0, atol=2e-6)
    if "labels" in r:
        np.testing.assert_array_equal(npy(r["labels"]), npy(ru["labels"]))


def test_camera_forward_c3():
    """BASELINE config 3 (camera novel view, hierarchical 64 + 128): orthonormal image-plane bases, per-ray radii."""
    from nerflidar_hip import camera as ncamera
    from nerflidar_hip.models import Model
    g = golden("camfwd_C3")
    mc = nconfig.workload("C3", int(g["log2_hashmap"]))
    sd = nweights.synth_state_dict(mc, seed=int(g["seed"]), trained_like=True)
    W, H, f = g["cam"]
    b = ncamera.synthetic_camera_batch(width=int(W), height=int(H), focal=float(f), seed=0, rows=g["rows"])
    model = Model(mc, sd, device=DEV, precision=_lib.PREC_FAST)
    rend, hist = model(False, {k: cu(v) for k, v in b.items()}, 1.0, True)
    r = rend[-1]
    assert np.abs(npy(r["depth"]) - g["out_depth"]).mean() <= 1e-3
    np.testing.assert_array_equal(npy(r["semantic"]).argmax(-1), g["out_semantic"].argmax(-1))
    assert np.abs(npy(r["rgb"]) - g["out_rgb"]).mean() <= 2e-3
    for lvl in range(mc.num_levels):
        assert np.abs(npy(hist[lvl]["sdist"][:24]) - g[f"hist{lvl}_sdist"]).mean() <= 1e-5


def test_random_sampling_against_oracle():
    """rand=True (training-time jitter, stepfun.py:211-216, render.py:149-150) with caller-provided uniform draws:
    the HIP path against the pinned oracle fed the same draws."""
    from nerflidar_hip.models import Model
    mc = nconfig.workload("REF", 14)
    sd = nweights.synth_state_dict(mc, seed=4, trained_like=True)
    batch_np = nlidar.synthetic_sweep(width=16, seed=4, beams=nlidar.LIDAR_ANGLES[::8])
    n = batch_np["origins"].shape[0]
    rng = np.random.default_rng(0)
    S = mc.level_samples()
    rj = [rng.random((n, 1)).astype(np.float32) for _ in S]
    rd = [rng.random((n, s, 7)).astype(np.float32) for s in S]
    model = Model(mc, sd, device=DEV, precision=_lib.PREC_F32)
    r, hist = model.render_rays({k: cu(v) for k, v in batch_np.items()}, want_history=True,
                                rand_jitter=[cu(x) for x in rj], rand_deg=[cu(x) for x in rd])
    ref, rh = orc.model_forward(sd, mc, {k: T(v) for k, v in batch_np.items()}, rand_jitter=[T(x) for x in rj],
                                rand_deg=[T(x) for x in rd])
    for lvl in range(mc.num_levels):
        assert np.abs(npy(hist[lvl]["sdist"]) - rh[lvl]["sdist"].numpy()).mean() <= 1e-5
    assert np.abs(npy(r["depth"]) - ref[-1]["depth"].numpy()).mean() <= 1e-3
    np.testing.assert_array_equal(npy(r["semantic"]).argmax(-1), ref[-1]["semantic"].numpy().argmax(-1))
    # and the draws matter: a deterministic render differs
    r0, _ = model.render_rays({k: cu(v) for k, v in batch_np.items()})
    assert np.abs(npy(r0["depth"]) - npy(r["depth"])).max() > 0


def test_render_image_driver_and_labels():
    """render_image (models.py:1379-1507) chunking == one-shot; labels/points post-step (render_lidar.py:142-161)."""
    from nerflidar_hip.models import render_image
    g = golden("fwd_REF_small")
    mc, sd, model = _model(g, _lib.PREC_MIXED)
    batch_np = nlidar.synthetic_sweep(width=int(g["width"]), seed=int(g["seed"]), beams=list(g["beams"]))
    batch = {k: cu(v) for k, v in batch_np.items()}
    cfg = nconfig.Config(render_chunk_size=40)
    out = render_image(model, None, batch, False, cfg, image=False)
    one, _ = model.render_rays(batch, scale_factor=1 / 250)
    for k in ("rgb", "depth", "semantic", "acc"):
        np.testing.assert_array_equal(npy(out[k]).reshape(npy(one[k]).shape), npy(one[k]))  # rays are independent: bit-identical
    np.testing.assert_array_equal(npy(one["labels"]), g["out_semantic"].argmax(-1))
    pts = (batch_np["origins"] + g["out_depth"][:, None] * batch_np["directions"]) * 250
    assert np.abs(npy(one["points"]) - pts).mean() <= 0.25  # 1e-3 depth L1 * 250 (1/scale_factor)


def test_full_size_properties():
    """BASELINE config C2 at its full size (32x1024 rays, 128 samples, 8x256): size-independent properties."""
    from nerflidar_hip.models import Model
    mc = nconfig.workload("C2")
    sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
    model = Model(mc, sd, device=DEV)
    batch_np = nlidar.synthetic_sweep(width=1024, seed=0)
    batch = {k: cu(v) for k, v in batch_np.items()}
    r1, h1 = model.render_rays(batch, want_history=True)
    r2, _ = model.render_rays(batch)
    r3, _ = model.render_rays(batch)
    torch.cuda.synchronize()
    for k in r2:
        np.testing.assert_array_equal(npy(r2[k]), npy(r3[k]))  # no atomics on the forward path: deterministic
    # with ray_history the per-sample heads go to HBM and nlr_composite_kernel sums them; without, the MLP kernel composites
    # inside its 32-sample segments (compositing mode): same weights, depth and acc bit for bit, the value sums in another order
    for k in ("depth", "acc", "distance_mean", "distance_median", "distance_percentile_5", "distance_percentile_95"):
        np.testing.assert_array_equal(npy(r1[k]), npy(r2[k]))
    for k in ("rgb", "semantic", "intensity"):
        np.testing.assert_allclose(npy(r1[k]), npy(r2[k]), rtol=0, atol=2e-6)
    for h in h1:
        s, w = npy(h["sdist"]), npy(h["weights"])
        assert (np.diff(s, axis=-1) >= 0).all() and s.min() >= 0 and s.max() <= 1
        assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-5).all()
    acc, sem = npy(r1["acc"]), npy(r1["semantic"])
    assert np.abs(acc - 1).max() < 1e-5  # opaque background: every ray is fully absorbed
    np.testing.assert_allclose(sem.sum(-1), acc, atol=1e-5)  # softmax rows composite to acc
    d = npy(r1["depth"])
    assert np.isfinite(d).all() and d.min() >= 0.008 - 1e-6 and d.max() <= 2.0 + 1e-6
    assert len(np.unique(sem.argmax(-1))) >= 5
    # a sector rendered alone equals the same rays inside the full sweep (what azimuth sharding relies on)
    sec, _ = nlidar.azimuth_sector(batch_np, 32, 1024, 3, 8)
    rs, _ = model.render_rays({k: cu(v) for k, v in sec.items()})
    idx = (np.arange(32)[:, None] * 1024 + np.arange(3 * 128, 4 * 128)[None, :]).reshape(-1)
    np.testing.assert_array_equal(npy(rs["depth"]), d[idx])


def test_c4_sectors_reassemble_the_sweep():
    """BASELINE config C4 on one GPU: the 8 azimuth sectors of ONE 32x1024 sweep, each rendered alone into its azimuth-major
    packed tile (written by the compositing kernel, NlrOut.packed), concatenated rank-major (what the all-gather does) ARE
    the packed image of the sweep rendered in one piece, bit for bit; and the records equal the named outputs."""
    from nerflidar_hip import sharding
    from nerflidar_hip.models import Model
    H, W, P = 32, 1024, 8
    mc = nconfig.workload("C2", 14)
    sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
    model = Model(mc, sd, device=DEV)
    batch_np = nlidar.synthetic_sweep(width=W, seed=0)
    whole = torch.empty(W, H, 7, device=DEV)
    r, _ = model.render_rays({k: cu(v) for k, v in batch_np.items()}, scale_factor=1 / 250, packed=whole)
    img = sharding.as_hw(whole)
    for k, sl in (("depth", 0), ("intensity", 1), ("acc", 2)):
        assert torch.equal(img[..., sl].reshape(-1), r[k])
    assert torch.equal(img[..., 3:6].reshape(-1, 3), r["rgb"])
    assert torch.equal(img[..., 6].reshape(-1).to(torch.int32), r["labels"])
    assert torch.equal(sharding.unpack_image(whole)["labels"].reshape(-1), r["labels"])
    tiles = []
    for p in range(P):
        sec, wp = nlidar.azimuth_sector(batch_np, H, W, p, P)
        t = torch.empty(wp, H, 7, device=DEV)
        model.render_rays({k: cu(v) for k, v in sec.items()}, scale_factor=1 / 250, packed=t)
        tiles.append(t)
    assert torch.equal(torch.cat(tiles), whole)
    flat = torch.empty(H * W, 7, device=DEV)  # ray-order records
    model.render_rays({k: cu(v) for k, v in batch_np.items()}, scale_factor=1 / 250, packed=flat)
    assert torch.equal(flat.reshape(H, W, 7), img)
    with pytest.raises(RuntimeError, match="packed"):
        model.render_rays({k: cu(v) for k, v in batch_np.items()}, packed=torch.empty(5, 7, device=DEV))


def test_reference_sweep_shape_1100_columns_in_8_padded_sectors():
    """The reference's real sweep, 32 beams x 1100 azimuths = 35 200 rays (ZI/lidar_utils.py:122-134,559-568; VERDICT r2 missing 2), in
    one piece and as 8 azimuth sectors of 138 columns (1104: the last sector carries 4 padded columns, repeats of the last real one),
    through the SweepGatherer's own buffers: the gathered-and-cropped image equals the one-piece image bit for bit, and the one-piece
    render agrees with the oracle on a sample of rays at the gates of the fixtures."""
    from nerflidar_hip import sharding
    from nerflidar_hip.models import Model
    H, W, P = 32, 1100, 8
    mc = nconfig.workload("C2", 14)
    sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
    model = Model(mc, sd, device=DEV)
    batch_np = nlidar.synthetic_sweep(width=W, seed=0)
    assert batch_np["origins"].shape[0] == 35200
    whole = torch.empty(W, H, 7, device=DEV)
    r, _ = model.render_rays({k: cu(v) for k, v in batch_np.items()}, scale_factor=1 / 250, packed=whole)
    wp = -(-W // P)
    assert wp == 138 and wp * P - W == 4
    tiles = []
    for p in range(P):
        sec, wq = nlidar.azimuth_sector(batch_np, H, W, p, P)
        assert wq == wp and sec["origins"].shape[0] == H * wp
        t = torch.empty(wp, H, 7, device=DEV)
        model.render_rays({k: cu(v) for k, v in sec.items()}, scale_factor=1 / 250, packed=t)
        tiles.append(t)
    gathered = torch.cat(tiles)                      # what all_gather_into_tensor leaves in SweepGatherer.images[b]: [P * wp, H, 7]
    assert gathered.shape[0] == 1104
    assert torch.equal(gathered[:W], whole)          # SweepGatherer.image crops the padded columns
    assert torch.equal(gathered[W:], gathered[W - 1:W].expand(4, H, 7))   # the padding repeats the last real column
    idx = np.linspace(0, 35199, 96).astype(np.int64)
    ref, _ = orc.model_forward(sd, mc, {k: T(np.ascontiguousarray(v[idx])) for k, v in batch_np.items()})
    gate("depth_1100", npy(r["depth"])[idx], ref[-1]["depth"].numpy(), 2e-4, 1e-2, thr=1e-3, frac=0.035)
    gate("intensity_1100", npy(r["intensity"])[idx], ref[-1]["intensity"].numpy(), 1e-4, 1e-3)
    np.testing.assert_array_equal(npy(r["labels"])[idx], ref[-1]["semantic"].numpy().argmax(-1))


def test_error_behaviour():
    from nerflidar_hip.models import Model
    mc = nconfig.workload("REF", 12)
    sd = nweights.synth_state_dict(mc, seed=0)
    model = Model(mc, sd, device=DEV)
    batch = {k: cu(v) for k, v in nlidar.synthetic_sweep(width=8, seed=0).items()}
    bad = dict(batch)
    bad.pop("base_x")
    with pytest.raises(RuntimeError, match="base_x"):
        model.render_rays(bad)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        model.render_rays({k: v.cpu() for k, v in batch.items()})
    mc2 = nconfig.workload("REF", 12)
    mc2.nerf_mlp.net_depth_viewdirs = 1
    with pytest.raises((RuntimeError, KeyError)):
        Model(mc2, nweights.synth_state_dict(mc2, seed=0), device=DEV)
    rc = _lib.lib().nlr_render_rays(model._handle, None, 4, None, None, None, 0, None)
    assert rc == -1 and b"NULL" in _lib.lib().nlr_last_error()


@pytest.mark.parametrize("wl", ["C2", "REF"])
def test_ragged_ray_counts_are_consistent(wl):
    """Rays are independent: any prefix of a batch must render bit-identically to the same rays inside the full batch, whatever
    the ray count does to tile / workgroup / wave boundaries (1 ray, non-multiples of 4, 32, 128; persistent-workgroup tails)."""
    mc = nconfig.workload(wl, 12)
    sd = nweights.synth_state_dict(mc, seed=11, trained_like=True)
    from nerflidar_hip.models import Model
    model = Model(mc, sd, device=DEV, precision=_lib.PREC_FAST)
    b = nlidar.synthetic_sweep(width=40, seed=11)   # 1280 rays
    full = {k: cu(v) for k, v in b.items()}
    rf, _ = model.render_rays(full, scale_factor=0.004)
    for n in (1, 3, 127, 129, 1000, 1279):
        rn, _ = model.render_rays({k: v[:n].contiguous() for k, v in full.items()}, scale_factor=0.004)
        for k in ("depth", "rgb", "semantic", "labels", "points", "acc", "distance_median"):
            assert torch.equal(rn[k], rf[k][:n]), (wl, n, k)


def test_static_sweep_replays_from_a_hip_graph():
    """`nlr_render_rays` reads nothing back to the host: the whole static sweep (resample / proposal / encode / MLP / composite of
    every level + the packed azimuth-major tile) is capturable, and a replay after the inputs were refilled in place is bit-identical to
    an eager call on the new inputs."""
    from nerflidar_hip.models import CapturedRender, Model
    mc = nconfig.workload("C2", 14)
    sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
    model = Model(mc, sd, device=DEV)
    H, wp = 8, 24
    sweeps = [nlidar.synthetic_sweep(width=wp, seed=s_, beams=nlidar.LIDAR_ANGLES[::4]) for s_ in (0, 1)]
    batch = {k: cu(v) for k, v in sweeps[0].items()}
    tile = torch.zeros(wp, H, 7, device=DEV)
    cap = CapturedRender(model, batch, compute_extras=True, scale_factor=1 / 250, packed=tile)
    for sw in (sweeps[1], sweeps[0]):
        for k, v in sw.items():
            batch[k].copy_(cu(v))               # refill the captured input buffers in place
        tile.zero_()
        for v in cap.out.values():
            v.zero_()
        out = cap.replay()
        torch.cuda.synchronize()
        eager_tile = torch.zeros(wp, H, 7, device=DEV)
        want, _ = model.render_rays({k: cu(v) for k, v in sw.items()}, compute_extras=True, scale_factor=1 / 250, packed=eager_tile)
        torch.cuda.synchronize()
        for k in ("depth", "intensity", "semantic", "labels", "rgb", "acc", "points", "distance_median"):
            assert torch.equal(out[k], want[k]), k
        assert torch.equal(tile, eager_tile)
        assert float(tile.abs().sum()) > 0


def test_captured_sweep_owns_its_workspace():
    """the graph carries the raw workspace address.  An eager render with MORE rays after the capture used to make the model
    replace (free) that arena; the capture now owns it, so the replay stays correct and the model's eager arena is a different tensor."""
    from nerflidar_hip.models import CapturedRender, Model
    mc = nconfig.workload("REF", 12)
    sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
    model = Model(mc, sd, device=DEV)
    small = {k: cu(v) for k, v in nlidar.synthetic_sweep(width=16, seed=0, beams=nlidar.LIDAR_ANGLES[::4]).items()}
    big = {k: cu(v) for k, v in nlidar.synthetic_sweep(width=512, seed=1).items()}
    want, _ = model.render_rays(small, scale_factor=1 / 250)
    want = {k: v.clone() for k, v in want.items()}
    cap = CapturedRender(model, small, scale_factor=1 / 250)
    assert model._ws is None and cap._ws is not None
    addr = cap._ws.data_ptr()
    model.render_rays(big, scale_factor=1 / 250)                       # larger n: the model allocates ITS arena
    junk = [torch.full((cap._ws.numel() // 4,), float("nan"), device=DEV) for _ in range(3)]  # would land in a freed arena
    assert model._ws is not None and model._ws.data_ptr() != addr and cap._ws.data_ptr() == addr
    for v in cap.out.values():
        v.zero_()
    out = cap.replay()
    torch.cuda.synchronize()
    for k in ("depth", "semantic", "labels", "rgb", "acc", "points"):
        assert torch.equal(out[k], want[k]), k
    del junk


@pytest.mark.parametrize("table_dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("workload,log2", [("C2", 14), ("REF", 21), ("C2", 10), ("P_F32", 15), ("P_F20", 12)])
def test_fast_level_body_is_bit_identical_to_the_generic_one(workload, log2, table_dtype):
    """`csrc/nlr_level_fast.h` (round 3: 32-bit offsets, reduced primes, scalar path for wave-uniform cells, x-pairs on dense levels,
    asm butterfly) claims the arithmetic of the generic body unchanged.  nlr_debug_set(NLR_DBG_FORCE_GENERIC, 1) routes the same launches through the
    round-2 kernels: every output of a whole render - which passes through nlr_prop8_kernel (C = 1) twice and nlr_encode8_kernel
    (C = 4) once, dense and hashed levels, fp32 and fp16 tables, 2^10 .. 2^21-entry hash maps - must be the same BITS.  The sweep is
    wide enough for uniform and non-uniform waves on every level.  P_F32: 16 levels x 2 features up to resolution 524 288, a grid round 3's
    envelope (written for a 24-bit multiply) sent to the generic kernels; P_F20: level_dim 2 (8-byte entries, two levels per feature piece)."""
    from nerflidar_hip.models import Model
    mc = nconfig.workload(workload, log2)
    off_, _, pls_ = nweights.grid_layout(mc.nerf_mlp)
    assert _lib.lib().nlr_grid_fast_path(np.ascontiguousarray(off_, np.int32).ctypes.data, mc.nerf_mlp.grid_num_levels, mc.nerf_mlp.grid_level_dim,
                                         float(np.log2(pls_)), 16, 0, 0, 0, 0) == 1
    sd = nweights.synth_state_dict(mc, seed=2, trained_like=True)
    model = Model(mc, sd, device=DEV, table_dtype=table_dtype)
    batch = {k: cu(v) for k, v in nlidar.synthetic_sweep(width=48, seed=4, beams=nlidar.LIDAR_ANGLES[::2]).items()}
    outs = []
    try:
        for generic in (0, 1):
            _lib.check(_lib.lib().nlr_debug_set(_lib.DBG_FORCE_GENERIC, generic))
            assert _lib.lib().nlr_debug_get(_lib.DBG_FORCE_GENERIC) == generic
            r, h = model.render_rays(batch, compute_extras=True, scale_factor=1 / 250, want_history=True)
            torch.cuda.synchronize()
            outs.append((r, h))
    finally:
        _lib.lib().nlr_debug_set(_lib.DBG_FORCE_GENERIC, 0)
    (r0, h0), (r1, h1) = outs
    for lvl in range(mc.num_levels):
        for k in ("sdist", "tdist", "density", "weights"):
            assert torch.equal(h0[lvl][k], h1[lvl][k]), f"level {lvl} {k}: fast and generic level bodies differ"
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
    assert float(h0[-1]["density"].max()) > 1.0
