"""Ray-drop stage on PyTorch-ROCm (scope row a-17, BASELINE config 5): the UNet and its training step.

north_star: "the ray-drop UNet (transfer_lidar_data.py) runs as a plain PyTorch-ROCm module on the rendered
output" -- so this is ordinary torch.nn (MIOpen convolutions), no custom kernel.  Same architecture and the same
`state_dict` keys as NeRF_Lidar_code/src/unet/unet_model.py:6-46 and unet_parts.py:8-77 (4 down / 4 up, bilinear
upsampling, BatchNorm, 2-class logits, optional sigmoid regression head), so a reference `.pth` loads unchanged.
The training step restates NeRF_Lidar_code/src/model/ray_drop_train.py:80-125 (azimuth roll augmentation, CE mask
loss, Gumbel-hard mask x range -> VGG-structured perceptual loss).  The reference's VGG19 ImageNet weights come from
torchvision (not installed, no network): `Vgg19Slices` keeps the structure (slices 0-2 / 2-7 / 7-12 / 12-21,
weights 1/16, 1/8, 1/4, 1 with the LAST feature skipped, VGG.py:13,28,50-56) and takes whatever weights it is
given (random here; a torchvision `vgg19().features.state_dict()` loads unchanged).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _double_conv(cin, cout, cmid=None):
    cmid = cmid or cout
    return nn.Sequential(nn.Conv2d(cin, cmid, 3, padding=1, bias=False), nn.BatchNorm2d(cmid), nn.ReLU(inplace=True),
                         nn.Conv2d(cmid, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DoubleConv(nn.Module):
    def __init__(self, cin, cout, cmid=None):
        super().__init__()
        self.double_conv = _double_conv(cin, cout, cmid)

    def forward(self, x):
        return self.double_conv(x)


class Down(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(cin, cout))

    def forward(self, x):
        return self.maxpool_conv(x)


class Up(nn.Module):
    def __init__(self, cin, cout, bilinear=True):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            self.conv = DoubleConv(cin, cout, cin // 2)
        else:
            self.up = nn.ConvTranspose2d(cin, cin // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(cin, cout)

    def forward(self, x1, x2):
        x1 = self.up(x1)
        dy, dx = x2.size(2) - x1.size(2), x2.size(3) - x1.size(3)
        x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(torch.cat([x2, x1], dim=1))


class OutConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=1)

    def forward(self, x):
        return self.conv(x)


class UNet(nn.Module):
    """unet_model.py:6-46.  Input [B, n_channels, 32, 1024] feature image, output logits [B, n_classes, 32, 1024]."""

    def __init__(self, n_channels, n_classes, bilinear=False, regression=False):
        super().__init__()
        self.n_channels, self.n_classes, self.bilinear, self.regression = n_channels, n_classes, bilinear, regression
        f = 2 if bilinear else 1
        self.inc = DoubleConv(n_channels, 64)
        self.down1, self.down2, self.down3 = Down(64, 128), Down(128, 256), Down(256, 512)
        self.down4 = Down(512, 1024 // f)
        self.up1, self.up2 = Up(1024, 512 // f, bilinear), Up(512, 256 // f, bilinear)
        self.up3, self.up4 = Up(256, 128 // f, bilinear), Up(128, 64, bilinear)
        self.outc = OutConv(64, n_classes)
        if regression:
            self.outr = OutConv(64, 1)

    def forward(self, x):
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        x = self.up4(self.up3(self.up2(self.up1(x5, x4), x3), x2), x1)
        logits = self.outc(x)
        if not self.regression:
            return logits
        return logits, torch.sigmoid(self.outr(x))


_VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512]  # features[0:21] of torchvision vgg19


class Vgg19Slices(nn.Module):
    """VGG.py:40-69: four feature slices of vgg19.features (indices 0-2, 2-7, 7-12, 12-21)."""

    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in _VGG19_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=False)]
                cin = v
        feats = nn.Sequential(*layers)  # 21 modules, indexed like torchvision's `features`
        self.slice1 = nn.Sequential(*[feats[i] for i in range(0, 2)])
        self.slice2 = nn.Sequential(*[feats[i] for i in range(2, 7)])
        self.slice3 = nn.Sequential(*[feats[i] for i in range(7, 12)])
        self.slice4 = nn.Sequential(*[feats[i] for i in range(12, 21)])
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        h1 = self.slice1(x)
        h2 = self.slice2(h1)
        h3 = self.slice3(h2)
        return [h1, h2, h3, self.slice4(h3)]


class VGGLoss(nn.Module):
    """VGG.py:7-38: per-pixel L1 between feature maps (upsampled to HxW), weights 1/16, 1/8, 1/4, last slice skipped."""

    def __init__(self, vgg: Vgg19Slices = None):
        super().__init__()
        self.vgg = vgg or Vgg19Slices()
        self.weights = [1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def forward(self, x, y):
        h, w = x.shape[-2:]
        x = x.unsqueeze(1).broadcast_to(x.shape[0], 3, h, w)
        y = y.unsqueeze(1).broadcast_to(x.shape[0], 3, h, w)
        fx, fy = self.vgg(x), self.vgg(y)
        loss = torch.zeros((x.shape[0], h, w), device=x.device)
        for i in range(len(fx) - 1):  # (sic) the deepest feature is never used: range(len(x_vgg)-1), VGG.py:28
            a, b = fx[i], fy[i].detach()
            if i > 0:
                a = F.interpolate(a, mode="bilinear", size=(h, w), align_corners=True)
                b = F.interpolate(b, mode="bilinear", size=(h, w), align_corners=True)
            loss = loss + (self.weights[i] * (a - b).abs()).mean(0).mean(0)
        return loss


def train_step(model: UNet, optim, vgg_loss: VGGLoss, img, gt_mask, gt_range, vgg_weight=0.5, roll=True, generator=None):
    """One iteration of ray_drop_train.py:80-125 (mask_loss=True, vgg=True, regression=False).
    img [B,F,H,W] (channel 0 = normalised range), gt_mask [B,H,W] int64 in {0,1}, gt_range [B,H,W]."""
    if roll:  # azimuth roll augmentation: the sweep is periodic in W
        d = int(torch.randint(0, img.shape[-1], (1,), generator=generator))
        img, gt_mask, gt_range = img.roll(d, dims=3), gt_mask.roll(d, dims=2), gt_range.roll(d, dims=2)
    pred = model(img)
    loss = F.cross_entropy(pred, gt_mask)
    mask = F.gumbel_softmax(pred, dim=1, hard=True)
    vl = vgg_loss(img[:, 0] * mask[:, 1], gt_range).mean()
    loss = loss + vgg_weight * vl
    optim.zero_grad()
    loss.backward()
    optim.step()
    return loss.detach(), vl.detach()


def range_projection(points, semantic=None, rgb=None, H=32, W=1024, fov_up=10.67, fov_down=-30.67):
    """LaserScan.do_range_projection on the GPU (libnerflidar_hip.so `nlr_range_project`; no CPU fallback).
    points [N,3] float64 CUDA tensor in the LiDAR frame; returns dict of [H,W,...] CUDA tensors."""
    import ctypes as C
    from . import _lib
    if not points.is_cuda:
        raise RuntimeError("points must be a CUDA tensor")
    pts = points.double().contiguous()
    n, dev = pts.shape[0], pts.device
    sem = None if semantic is None else semantic.float().contiguous()
    col = None if rgb is None else rgb.float().contiguous()
    out = dict(proj_range=torch.empty(H, W, device=dev), proj_xyz=torch.empty(H, W, 3, device=dev),
               proj_semantic=torch.empty(H, W, device=dev), proj_rgb=torch.empty(H, W, 3, device=dev),
               proj_idx=torch.empty(H, W, device=dev, dtype=torch.int32), proj_mask=torch.empty(H, W, device=dev))
    L = _lib.lib()
    ws = torch.empty(L.nlr_range_workspace_bytes(H, W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.nlr_range_project(_lib.ptr(pts), _lib.ptr(sem), _lib.ptr(col), n, H, W, float(fov_up), float(fov_down), _lib.ptr(ws),
                                 ws.numel(), _lib.ptr(out["proj_range"]), _lib.ptr(out["proj_xyz"]), _lib.ptr(out["proj_semantic"]),
                                 _lib.ptr(out["proj_rgb"]), _lib.ptr(out["proj_idx"]), _lib.ptr(out["proj_mask"]), _lib.current_stream())
    _lib.check(rc, "nlr_range_project")
    return out


def unet_features(proj, var=True):
    """Stack the UNet input as Generate_feature.py:44-49,161-166 does: [log-range, semantic, rgb x3, (variance over the
    4-column window of real_to_var(size=2))] -> [1, F, H, W]."""
    real = proj["proj_range"]
    lr = torch.clamp(torch.log2(torch.where(real < 0, torch.zeros_like(real), real) + 0.0001 + 1) / 6.5, 0, 1)
    feats = [lr, proj["proj_semantic"], proj["proj_rgb"][..., 0], proj["proj_rgb"][..., 1], proj["proj_rgb"][..., 2]]
    if var:
        win = torch.stack([torch.roll(lr, i, dims=1) for i in range(-2, 2)], dim=-1)
        feats.append(win.var(dim=-1, unbiased=False))
    return torch.stack(feats, 0)[None]


# ---- applying a trained ray-drop UNet to a rendered sweep (scope row f-4, second half) ----------------------------------
def apply_ray_drop(proj, logits, mask_thre=0.5, place_car=False, car_label=13, sky_label=10, road_label=0, road_z=-3.0):
    """NeRF_Lidar_code/src/drop_simulation_rays.py:88-166, the `save_near` branch without depth filter: keep the range-image
    pixels whose keep-probability `softmax(logits)[1]` exceeds `mask_thre` and that hold a point; with `place_car`, car
    pixels are first thresholded at their own median probability (:103-108).  Then drop sky points and road points below
    z = -3 m (:158-164).  proj: dict from `range_projection`; logits [2,H,W] (or [1,2,H,W]).  Returns (points [K,3] f32,
    labels [K] int64), both CUDA tensors, in row-major pixel order like the reference's boolean indexing."""
    logits = logits.reshape((2,) + tuple(proj["proj_range"].shape)).float()
    prob = torch.softmax(logits, dim=0)[1]
    sem = proj["proj_semantic"]
    if place_car:
        car = sem == car_label
        if bool(car.any()):
            thre = torch.quantile(prob[car], 0.5)  # np.percentile(., 50): linear interpolation, the torch default too
            prob = torch.where(car, (prob > thre).to(prob.dtype), prob)
    keep = (prob > mask_thre) & (proj["proj_mask"] == 1)
    pts, lab = proj["proj_xyz"][keep], sem[keep].to(torch.int64)
    ok = lab != sky_label
    pts, lab = pts[ok], lab[ok]
    ok = ~((lab == road_label) & (pts[:, 2] < road_z))
    return pts[ok], lab[ok]


def write_points_and_labels(index, savepath, points, labels):
    """KITTI-style pair as drop_simulation_rays.py:14-22 writes it: `velodyne/%06d.bin` = the points' float32 values back to
    back (three per point here), `labels/%06d.label` = uint32 per point."""
    import os
    import numpy as np
    os.makedirs(os.path.join(savepath, "velodyne"), exist_ok=True)
    os.makedirs(os.path.join(savepath, "labels"), exist_ok=True)
    p = points.detach().cpu().numpy() if isinstance(points, torch.Tensor) else np.asarray(points)
    l = labels.detach().cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
    p.astype(np.float32).tofile(os.path.join(savepath, "velodyne", "{:06d}.bin".format(index)))
    l.astype(np.uint32).tofile(os.path.join(savepath, "labels", "{:06d}.label".format(index)))
