"""Parity at the BENCHMARKED size (-m gpu): C2, full-size fp32 tables, one 32 x 1024 sweep, all three MLP precisions, against the
fp32 oracle on 8 192 rays of the sweep, with OUTLIER-FRACTION gates.

Why fractions and not maxima: the synthetic scene is white noise at every grid resolution under a x1500 density gain, so a few rays per
thousand are ill-conditioned - a 1-ulp difference anywhere upstream moves a surface crossing to another sample.  On those rays the
REFERENCE'S OWN fp32 arithmetic is as far from a float64 evaluation of its algorithm as the GPU is (second test below; full report:
profiles/r03_parity_tail.txt, made by tests/parity_tail.py).  The small fixtures (48-96 rays) cannot see a 0.5 % tail at all, and
they never run a workgroup over more than one tile: round 2's exact-f32 instance evaluated every later tile of a workgroup on its
first tile's features, which only this size shows.
"""
import os

import numpy as np
import pytest
import torch

import parity_tail as pt
from nerflidar_hip import _lib, config as nconfig, lidar as nlidar, weights as nweights
from nerflidar_hip.models import Model
from oracle import nlr_oracle as orc

pytestmark = pytest.mark.gpu
N_ORACLE = 8192


@pytest.fixture(scope="module")
def sweep():
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    mc = nconfig.workload("C2", None)
    sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
    full = nlidar.synthetic_sweep(width=1024, seed=0)
    idx = np.linspace(0, full["origins"].shape[0] - 1, N_ORACLE).astype(np.int64)  # the rays bench.py's `accuracy` samples
    ref, refh = pt.oracle_run(sd, mc, full, idx)
    dev = torch.device("cuda:0")
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in full.items()}
    out = {}
    for prec, name in pt.PRECS:
        model = Model(mc, sd, device=dev, precision=prec)
        r, h = model.render_rays(batch, compute_extras=True, scale_factor=1 / 250, want_history=True)
        r2, _ = model.render_rays(batch, compute_extras=True, scale_factor=1 / 250)  # the render path bench.py times
        torch.cuda.synchronize()
        keys = ("depth", "intensity", "semantic", "labels", "acc")
        out[name] = {k: r[k].cpu().numpy() for k in keys}
        out[name + "/render"] = {k: r2[k].cpu().numpy() for k in keys}
        out[name + "/density"] = h[-1]["density"].cpu().numpy()
        del model
    return dict(mc=mc, sd=sd, full=full, idx=idx, ref=ref, out=out)


def _fractions(g, ref):
    d = np.abs(g["depth"] - ref["depth"])
    i = np.abs(g["intensity"] - ref["intensity"])
    return dict(depth_l1=d.mean(), depth_p95=np.percentile(d, 95), f3=np.mean(d > 1e-3), f2=np.mean(d > 1e-2),
                int_l1=i.mean(), fi3=np.mean(i > 1e-3), labels=int((g["semantic"].argmax(-1) != ref["semantic"].argmax(-1)).sum()))


@pytest.mark.parametrize("name", ["F32", "MIXED", "FAST", "F32/render", "MIXED/render", "FAST/render"])
def test_fullsize_against_oracle(sweep, name):
    g = {k: v[sweep["idx"]] for k, v in sweep["out"][name].items()}
    s = _fractions(g, sweep["ref"])
    msg = f"{name}: {s}"
    # north_star: depth L1 within 1e-3 (measured 7.1e-5), intensity within 1e-3, labels bit-exact
    assert s["depth_l1"] <= 2e-4 and s["depth_p95"] <= 2e-4, msg
    assert s["int_l1"] <= 1e-4, msg
    assert s["labels"] == 0, msg
    assert (sweep["out"][name]["labels"][sweep["idx"]] == sweep["ref"]["semantic"].argmax(-1)).all(), msg
    # the tail, as fractions of the 8 192 rays (measured: 0.47 % / 0.085 % / 0.23 %; the oracle itself against float64 on evenly
    # spaced rays: 0.4 % / 0 / 0.2 %, see the next test)
    assert s["f3"] <= 0.01, msg
    assert s["f2"] <= 0.003, msg
    assert s["fi3"] <= 0.006, msg


def test_fullsize_precisions_agree(sweep):
    """All 32 768 rays: the three precisions render the same sweep (this is what catches a precision-specific kernel defect at a
    size the fixtures do not reach), and the render path equals the ray_history path of the same precision."""
    o = sweep["out"]
    for a, b in (("F32", "FAST"), ("MIXED", "FAST"), ("F32", "MIXED")):
        d = np.abs(o[a]["depth"] - o[b]["depth"])
        dd = np.abs(o[a + "/density"] - o[b + "/density"])
        lab = (o[a]["labels"] != o[b]["labels"]).mean()
        msg = f"{a} vs {b}: depth L1 {d.mean():.2e} frac>1e-3 {np.mean(d > 1e-3):.4f} density mean |d| {dd.mean():.2e} labels differ {lab:.5f}"
        assert d.mean() <= 2e-4 and np.mean(d > 1e-3) <= 0.01, msg
        assert dd.mean() <= 2e-3, msg  # densities reach ~150; split-bf16 trunk: ~2e-4 mean
        assert lab <= 2e-4, msg        # a handful of rays whose top-2 class margin is below the arithmetic's resolution
    for p in ("F32", "MIXED", "FAST"):
        np.testing.assert_array_equal(o[p]["depth"], o[p + "/render"]["depth"])
        np.testing.assert_array_equal(o[p]["acc"], o[p + "/render"]["acc"])
        np.testing.assert_allclose(o[p]["intensity"], o[p + "/render"]["intensity"], atol=2e-6, rtol=0)
        np.testing.assert_array_equal(o[p]["labels"], o[p + "/render"]["labels"])


def test_fullsize_gpu_is_as_close_to_float64_as_the_reference_arithmetic(sweep):
    """The same algorithm evaluated in float64 (oracle functions on float64 tensors, float64 grid interpolation) on 512 evenly
    spaced rays: the fp32 oracle's distance from it and the GPU's are the same distribution - the tail is conditioning, not a defect."""
    mc, sd, full, idx = sweep["mc"], sweep["sd"], sweep["full"], sweep["idx"]
    sub = np.linspace(0, len(idx) - 1, 512).astype(np.int64)
    torch.set_default_dtype(torch.float64)
    try:
        enc64 = {k: pt.GridEncoder64(e.table, e.offsets, e.grid_sizes.numpy(), 2.0 ** e.S, e.H) for k, e in orc.make_encoders(sd, mc).items()}
        ref64, _ = pt.oracle_run(sd, mc, full, idx[sub], dtype=torch.float64, chunk=256, enc=enc64)
    finally:
        torch.set_default_dtype(torch.float32)
    ref32 = {k: v[sub] for k, v in sweep["ref"].items()}
    so = _fractions(ref32, ref64)
    for name in ("F32", "MIXED", "FAST"):
        g = {k: v[idx[sub]] for k, v in sweep["out"][name].items()}
        sg = _fractions(g, ref64)
        msg = f"{name} vs float64 {sg}; oracle32 vs float64 {so}"
        assert sg["depth_l1"] <= 2.0 * so["depth_l1"] + 2e-5, msg
        assert sg["f3"] <= 1.5 * so["f3"] + 0.006, msg
        assert sg["f2"] <= 1.5 * so["f2"] + 0.004, msg
        assert sg["fi3"] <= 1.5 * so["fi3"] + 0.006, msg
        assert sg["labels"] == 0, msg
