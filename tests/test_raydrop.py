"""Ray-drop stage (scope row a-17, BASELINE config 5): the PyTorch-ROCm UNet against the reference's UNet outputs
(fixtures from tests/golden/make_golden.py importing NeRF_Lidar_code/src/unet), and the training step."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden
from nerflidar_hip import raydrop

sys.path.insert(0, GOLDEN)
from unet_fill import unet_fill  # noqa: E402  deterministic parameters shared with the fixture generator


@pytest.mark.parametrize("tag,reg", [("logits", False), ("regression", True)])
def test_unet_matches_reference(tag, reg):
    g = golden(f"unet_{tag}")
    m = raydrop.UNet(n_channels=6, n_classes=2, bilinear=True, regression=reg).eval()
    unet_fill(m, 7)
    with torch.no_grad():
        out = m(torch.from_numpy(g["x"]))
    logits = out[0] if reg else out
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=1e-5, rtol=1e-5)
    if reg:
        np.testing.assert_allclose(out[1].numpy(), g["reg"], atol=1e-6, rtol=1e-5)


def test_state_dict_keys_are_the_references():
    keys = set(raydrop.UNet(6, 2, bilinear=True).state_dict())
    assert {"inc.double_conv.0.weight", "down4.maxpool_conv.1.double_conv.4.running_var", "up1.conv.double_conv.3.weight",
            "outc.conv.bias"} <= keys
    assert len([k for k in keys if k.endswith("weight") and "double_conv.0" in k]) == 9


def test_train_step_cpu_small():
    torch.manual_seed(0)
    m = raydrop.UNet(6, 2, bilinear=True)
    opt = torch.optim.Adam(m.parameters())
    vl = raydrop.VGGLoss()
    img = torch.rand(2, 6, 32, 64)
    mask = (torch.rand(2, 32, 64) > 0.3).long()
    rng = img[:, 0] * mask
    l0, _ = raydrop.train_step(m, opt, vl, img, mask, rng)
    l1, _ = raydrop.train_step(m, opt, vl, img, mask, rng)
    assert torch.isfinite(l0) and torch.isfinite(l1)


def test_range_projection_oracle_matches_reference():
    """f-2 oracle (numpy restatement) against the reference's LaserScan.do_range_projection / pcs2img / real_to_var."""
    from oracle import nlr_oracle as orc
    g = golden("fn_range_image")
    o = orc.range_projection(g["points"], g["semantic"], g["rgb"], H=32, W=256)
    for k in ("proj_range", "proj_semantic", "proj_mask", "proj_rgb", "proj_xyz", "proj_idx"):
        np.testing.assert_array_equal(o[k], g[k], err_msg=k)
    lr = orc.log_range(o["proj_range"])
    np.testing.assert_array_equal(lr, g["log_range"])
    np.testing.assert_allclose(orc.real_to_var(lr, size=2), g["var2"], rtol=1e-12, atol=0)
    assert (g["proj_idx"] >= 0).sum() > 3000 and (g["proj_mask"] == 0).sum() > 0


def _apply_fixture():
    from nerflidar_hip import synth
    g = golden("fn_raydrop_apply")
    logits = synth.uniform(5, 63, (2, 32, 1024), -2.0, 2.0).astype(np.float32)  # what the fixture's stub runner returned
    return g, logits


@pytest.mark.parametrize("tag,place_car", [("plain", False), ("car", True)])
def test_apply_ray_drop_matches_reference_run(tag, place_car):
    """f-4: `apply_ray_drop` against the REFERENCE's drop_simulation_rays.drop_simulation run end to end on the same sweep and
    logits (tests/golden/make_golden.py:gen_raydrop_apply): same surviving points, same labels, same order.  The projection
    here is the oracle's (CPU); the GPU test below runs the HIP projection."""
    from oracle import nlr_oracle as orc
    g, logits = _apply_fixture()
    o = orc.range_projection(g["points"], g["semantic"], None, H=32, W=1024)
    proj = {k: torch.from_numpy(np.asarray(v)) for k, v in o.items() if v is not None}
    pts, lab = raydrop.apply_ray_drop(proj, torch.from_numpy(logits), mask_thre=0.5, place_car=place_car)
    np.testing.assert_array_equal(lab.numpy(), g[f"{tag}_labels"].astype(np.int64))
    np.testing.assert_array_equal(pts.numpy().astype(np.float32), g[f"{tag}_points"].astype(np.float32))
    assert 2000 < len(lab) < len(g["semantic"]) and (lab != 10).all()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,place_car", [("plain", False), ("car", True)])
def test_apply_ray_drop_on_gpu_matches_reference_run(tag, place_car):
    g, logits = _apply_fixture()
    dev = "cuda:0"
    proj = raydrop.range_projection(torch.from_numpy(g["points"]).to(dev), torch.from_numpy(g["semantic"]).to(dev), None, H=32, W=1024)
    pts, lab = raydrop.apply_ray_drop(proj, torch.from_numpy(logits).to(dev), mask_thre=0.5, place_car=place_car)
    np.testing.assert_array_equal(lab.cpu().numpy(), g[f"{tag}_labels"].astype(np.int64))
    np.testing.assert_array_equal(pts.cpu().numpy().astype(np.float32), g[f"{tag}_points"].astype(np.float32))


@pytest.mark.gpu
def test_range_projection_gpu_bit_exact():
    """nlr_range_project (HIP) against the reference's projection: every pixel picks the same (nearest) point."""
    g = golden("fn_range_image")
    dev = "cuda:0"
    out = raydrop.range_projection(torch.from_numpy(g["points"]).to(dev), torch.from_numpy(g["semantic"]).to(dev),
                                   torch.from_numpy(g["rgb"]).to(dev), H=32, W=256)
    np.testing.assert_array_equal(out["proj_idx"].cpu().numpy(), g["proj_idx"])       # index work: bit-exact
    np.testing.assert_array_equal(out["proj_mask"].cpu().numpy(), g["proj_mask"])
    np.testing.assert_array_equal(out["proj_semantic"].cpu().numpy(), g["proj_semantic"])
    np.testing.assert_array_equal(out["proj_range"].cpu().numpy(), g["proj_range"])   # f64 norm rounded to f32
    np.testing.assert_array_equal(out["proj_rgb"].cpu().numpy(), g["proj_rgb"].astype(np.float32))
    np.testing.assert_array_equal(out["proj_xyz"].cpu().numpy(), g["proj_xyz"])
    f = raydrop.unet_features(out)
    assert f.shape == (1, 6, 32, 256)
    np.testing.assert_allclose(f[0, 0].cpu().numpy(), g["log_range"], atol=1e-6)
    np.testing.assert_allclose(f[0, 5].cpu().numpy(), g["var2"], atol=1e-6)
    # empty input and the mask quirk
    e = raydrop.range_projection(torch.zeros(0, 3, dtype=torch.float64, device=dev), H=4, W=8)
    assert (e["proj_idx"] == -1).all() and (e["proj_range"] == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,reg", [("logits", False), ("regression", True)])
def test_unet_matches_reference_on_gpu(tag, reg):
    """The same fixtures as test_unet_matches_reference (outputs of the reference's UNet), through MIOpen on the GPU."""
    g = golden(f"unet_{tag}")
    m = raydrop.UNet(n_channels=6, n_classes=2, bilinear=True, regression=reg).eval()
    unet_fill(m, 7)
    m = m.to("cuda:0")
    with torch.no_grad():
        out = m(torch.from_numpy(g["x"]).to("cuda:0"))
    logits = out[0] if reg else out
    # MIOpen convolutions sum in another order than the CPU reference: 1e-4 absolute on logits of magnitude ~1
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], atol=1e-4, rtol=1e-4)
    if reg:
        np.testing.assert_allclose(out[1].cpu().numpy(), g["reg"], atol=1e-4, rtol=1e-4)
    assert (logits.argmax(1).cpu().numpy() == g["logits"].argmax(1)).mean() > 0.999  # the keep/drop decision itself


def _ce_step(dev):
    g = golden("unet_ce_step")
    m = raydrop.UNet(n_channels=6, n_classes=2, bilinear=True, regression=False).train()
    unet_fill(m, 9)
    m = m.to(dev)
    x, gt = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["gt_mask"]).to(dev)
    pred = m(x)
    loss = torch.nn.functional.cross_entropy(pred, gt)      # ray_drop_train.py:100-101 (mask_loss, weights = 1.)
    loss.backward()
    return g, m, pred, loss


def _ce_grads(dev, dtype):
    """Loss and the fixture's six gradients (first 4 096 values each, as float64 on the CPU) of the mask-loss step in `dtype` on `dev`."""
    g = golden("unet_ce_step")
    m = raydrop.UNet(n_channels=6, n_classes=2, bilinear=True, regression=False).train()
    unet_fill(m, 9)
    m = m.to(dev, dtype)
    x, gt = torch.from_numpy(g["x"]).to(dev, dtype), torch.from_numpy(g["gt_mask"]).to(dev)
    loss = torch.nn.functional.cross_entropy(m(x), gt)
    loss.backward()
    named = dict(m.named_parameters())
    return float(loss.detach()), {k[5:]: named[k[5:]].grad.reshape(-1)[:4096].detach().double().cpu() for k in g if k.startswith("grad_")}


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def test_unet_mask_loss_step_matches_reference():
    """The mask term of a ray-drop training iteration (ray_drop_train.py:96-101, 123-124) on the reference's UNet in train mode - loss
    and gradients of six parameters from the first to the last convolution - against `raydrop.UNet` on the CPU, same arithmetic: 1e-4.
    And how far that fp32 arithmetic is from the exact result: the same step in float64 differs from the REFERENCE's fp32 gradients by
    2e-3 of their norm on the first convolutions (nine train-mode BatchNorms and the whole backward chain behind them), 1e-7 on the last.
    Any fp32 implementation that sums in another order (MIOpen) is that far from the reference too; the GPU test below therefore holds the
    GPU to the float64 evaluation, not to one particular fp32 summation order."""
    g, m, pred, loss = _ce_step("cpu")
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-5)
    np.testing.assert_allclose(pred[:1].detach().cpu().numpy(), g["logits"], atol=1e-4, rtol=1e-4)
    named = dict(m.named_parameters())
    _, g64 = _ce_grads("cpu", torch.float64)
    noise = {}
    for k in [k[5:] for k in g if k.startswith("grad_")]:
        got = named[k].grad.reshape(-1).detach().cpu()
        want = torch.from_numpy(g["grad_" + k])
        assert _rel(got[:4096], want) <= 1e-4, k
        np.testing.assert_allclose(float(got.double().norm()), float(g["gnorm_" + k]), rtol=1e-4, err_msg=k)
        noise[k] = _rel(want.double(), g64[k])
    # measured: 2.2e-3, 2.0e-3, 2.5e-3, 1.3e-3, 1.7e-7, 2.7e-7
    assert 5e-4 < noise["inc.double_conv.0.weight"] < 6e-3 and noise["outc.conv.weight"] < 1e-5, noise


@pytest.mark.gpu
def test_unet_mask_loss_step_on_gpu_against_float64():
    """MIOpen against the exact result.  (a) The step in float64 on the GPU equals the float64 step on the CPU to 1e-9: the implementation
    is the reference's.  (b) The fp32 step on the GPU is as close to float64 as the reference's own fp32 step is (gate: 3x the reference's
    distance per parameter + 1e-5; measured on the GPU: the same 2e-3 scale), under the default and under the deterministic MIOpen algorithm
    selection.  Round 3 held the GPU to 2e-3 of the CPU fp32 gradients, then to 1e-2 after a red run: both numbers compared two roundings
    of an ill-conditioned sum with each other."""
    g = golden("unet_ce_step")
    l64c, g64c = _ce_grads("cpu", torch.float64)
    l64g, g64g = _ce_grads("cuda:0", torch.float64)
    assert abs(l64g - l64c) <= 1e-12 * abs(l64c) + 1e-12
    for k in g64c:
        assert _rel(g64g[k], g64c[k]) <= 1e-9, f"float64 GPU vs CPU {k}: {_rel(g64g[k], g64c[k]):.2e}"
    ref_noise = {k: _rel(torch.from_numpy(g["grad_" + k]).double(), g64c[k]) for k in g64c}
    report = []
    for det in (False, True):
        old = torch.backends.cudnn.deterministic
        torch.backends.cudnn.deterministic = det
        try:
            l32, g32 = _ce_grads("cuda:0", torch.float32)
        finally:
            torch.backends.cudnn.deterministic = old
        assert abs(l32 - l64c) <= 2e-6 * abs(l64c)
        for k in g64c:
            r = _rel(g32[k], g64c[k])
            report.append(f"{k} deterministic={det}: GPU fp32 vs f64 {r:.2e} (reference fp32 vs f64 {ref_noise[k]:.2e})")
            assert r <= 3 * ref_noise[k] + 1e-5, report[-1]
    print("\n".join(report))


@pytest.mark.gpu
def test_config5_unet_batch8_on_gpu():
    """BASELINE config 5: rendered sweep -> UNet(VGG-structured loss), batch 8, [8,6,32,1024], fwd + bwd on the GPU."""
    dev = "cuda:0"
    torch.manual_seed(0)
    m = raydrop.UNet(6, 2, bilinear=True).to(dev)
    opt = torch.optim.Adam(m.parameters())
    vl = raydrop.VGGLoss().to(dev)
    img = torch.rand(8, 6, 32, 1024, device=dev)
    mask = (torch.rand(8, 32, 1024, device=dev) > 0.3).long()
    rng = img[:, 0] * mask
    losses = [float(raydrop.train_step(m, opt, vl, img, mask, rng)[0]) for _ in range(3)]
    assert all(np.isfinite(losses))


@pytest.mark.gpu
def test_config5_chain_from_rendered_sweeps(tmp_path):
    """BASELINE config 5 END TO END: 8 sweeps rendered by the fused path from the trained checkpoint -> range
    projection -> UNet feature stack (all on the device: `render_lidar.raydrop_batch`) -> `raydrop.train_step` at batch 8 (CE + VGG-structured
    loss) -> the trained-for-a-few-steps UNet applied to a sweep -> `.bin` / `.label`.  Round 3 ran this configuration on `torch.rand` images."""
    import os
    from conftest import GOLDEN
    from nerflidar_hip import checkpoints as nckpt, config as nconfig, render_lidar as nrl
    dev = "cuda:0"
    model, step, _ = nckpt.model_from_checkpoint(os.path.join(GOLDEN, "ckpt_trained"), base=nconfig.workload("REFI", 13), device=dev)
    img, gt_mask, gt_range, projs = nrl.raydrop_batch(model, list(range(100, 108)))
    assert img.shape == (8, 6, 32, 1024) and gt_mask.shape == (8, 32, 1024) and gt_range.shape == (8, 32, 1024) and img.is_cuda
    # the rendered sweeps are a scene: (nearly) every pixel of the range image holds a point, several classes, ranges out to tens of metres
    assert float(projs[0]["proj_mask"].mean()) > 0.9
    assert len(torch.unique(projs[0]["proj_semantic"][projs[0]["proj_mask"] == 1])) >= 6
    assert 0.2 < float(gt_mask.float().mean()) < 0.95           # the analytic drop rule keeps some returns and drops others
    assert float(img[:, 0].max()) <= 1.0 and float(img[:, 0].mean()) > 0.3
    torch.manual_seed(0)
    m = raydrop.UNet(6, 2, bilinear=True).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    vl = raydrop.VGGLoss().to(dev)
    first = [float(raydrop.train_step(m, opt, vl, img, gt_mask, gt_range)[0]) for _ in range(3)]
    for _ in range(25):
        raydrop.train_step(m, opt, vl, img, gt_mask, gt_range)
    last = [float(raydrop.train_step(m, opt, vl, img, gt_mask, gt_range)[0]) for _ in range(3)]
    assert all(np.isfinite(first + last)) and np.mean(last) < 0.9 * np.mean(first), (first, last)
    m.eval()
    with torch.no_grad():
        logits = m(img[:1])
    pts, lab = raydrop.apply_ray_drop(projs[0], logits[0], mask_thre=0.5)
    kept = pts.shape[0] / float(projs[0]["proj_mask"].sum())
    assert 0.05 < kept < 1.0 and pts.shape[0] == lab.shape[0]
    raydrop.write_points_and_labels(0, str(tmp_path), pts, lab)
    assert os.path.getsize(tmp_path / "velodyne" / "000000.bin") == pts.shape[0] * 12
    assert os.path.getsize(tmp_path / "labels" / "000000.label") == pts.shape[0] * 4
