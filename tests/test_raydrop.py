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


def test_range_features_shape():
    d = torch.rand(32, 64) * 0.4
    f = raydrop.range_features(d, torch.randint(0, 19, (32, 64)), torch.rand(32, 64, 3), 1 / 250)
    assert f.shape == (1, 6, 32, 64) and torch.isfinite(f).all()


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
