"""BASELINE config 5: ray-drop UNet training step (CE + VGG-structured loss) on [8, 6, 32, 1024] range images, PyTorch-ROCm / MIOpen."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import torch
from nerflidar_hip import raydrop
torch.manual_seed(0)
dev = "cuda"
m = raydrop.UNet(6, 2, bilinear=True).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
vl = raydrop.VGGLoss().to(dev)
B = 8
img = torch.rand(B, 6, 32, 1024, device=dev)
mask = (torch.rand(B, 32, 1024, device=dev) > 0.3).long()
rng = img[:, 0] * mask
def step(): return raydrop.train_step(m, opt, vl, img, mask, rng)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): l, v = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"C5 train step (UNet fwd+bwd + VGG loss, batch {B}): {dt*1e3:.1f} ms -> {B/dt:.0f} sweeps/s, loss {float(l):.3f}")
m.eval()
with torch.no_grad():
    for _ in range(3): m(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m(img)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"C5 inference (UNet forward, batch {B}): {dt*1e3:.2f} ms -> {B/dt:.0f} sweeps/s")
