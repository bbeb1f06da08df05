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

# ---- the whole chain from rendered sweeps: render -> project -> stack -> UNet, nothing leaves the device ------------
from nerflidar_hip import checkpoints as nckpt, config as nconfig, render_lidar as nrl, weights as nweights
ck = os.path.join(ROOT, "tests", "golden", "ckpt_trained_c2")
import json
summ = json.load(open(os.path.join(ck, "train_summary.json")))["summary"]
sd, _ = nckpt.load_checkpoint(ck); sd, _ = nckpt.split_state_dict(sd)
mc = nckpt.infer_model_config(sd, nconfig.workload(summ["workload"], summ["log2_hashmap"]))
sd, mc = nweights.inflate_hashmaps(sd, mc, 21)
from nerflidar_hip.models import Model
model = Model(mc, sd, device=dev)
ids = list(range(100, 108))
sweeps = [{k: torch.from_numpy(v).to(dev) for k, v in nrl.nlidar.synthetic_sweep(width=1024, seed=0, sweep_idx=i).items()} for i in ids]
img8, gm, gr, projs = nrl.raydrop_batch(model, ids, batches=sweeps)            # warm-up; the "recorded" truth is made once
m.train()
n = 5
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    img8, _, _, projs = nrl.raydrop_batch(model, ids, batches=sweeps, truth=(gm, gr))
    l, v = raydrop.train_step(m, opt, vl, img8, gm, gr)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"C5 whole chain, training: 8 sweeps rendered (C2 trained checkpoint, full-size maps) + projected + stacked + UNet step (batch 8): {dt*1e3:.1f} ms -> {8/dt:.0f} sweeps/s")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    img8, _, _, projs = nrl.raydrop_batch(model, ids, batches=sweeps, truth=(gm, gr))
torch.cuda.synchronize(); dt_in = (time.perf_counter() - t0) / n
print(f"   of which input (render + project + stack, 8 sweeps): {dt_in*1e3:.1f} ms; the analytic stand-in for the recorded frames (made once) is not timed")
m.eval()
rot = torch.from_numpy(nrl.nlidar.seeded_rotation(0)).float().to(dev)
def apply_one(b):
    res = nrl.render_sweep_device(model, b, 1 / 250)
    img1, proj = nrl.sweep_unet_input(res, b["origins"][0] * 250, rot)
    with torch.no_grad():
        logits = m(img1)
    return raydrop.apply_ray_drop(proj, logits[0])
apply_one(sweeps[0])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    for b in sweeps:
        pts, lab = apply_one(b)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (3 * len(sweeps))
print(f"C5 whole chain, simulation: render -> project -> stack -> UNet forward -> drop mask -> kept points, per sweep: {dt*1e3:.2f} ms -> {1/dt:.0f} sweeps/s ({pts.shape[0]} of 32768 points kept in the last one)")
