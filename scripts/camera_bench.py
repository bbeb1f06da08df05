"""Config C3: novel-view render of 1024x768 pinhole images, hierarchical (64 + 128 samples), 8x256 NerfMLP,
through `render_image` with the reference's chunking (Config.render_chunk_size rays per call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import camera as ncamera, config as nconfig, weights as nweights
from nerflidar_hip.models import Model, render_image
mc = nconfig.workload("C3")
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
mc.config.render_chunk_size = chunk
model = Model(mc, nweights.synth_state_dict(mc, seed=0, trained_like=True), precision=2)
H, W = 768, 1024
batches = []
for cam in range(4):
    b = ncamera.synthetic_camera_batch(width=W, height=H, seed=cam)
    batches.append({k: torch.from_numpy(v).cuda().reshape(H, W, -1) for k, v in b.items()})
def run():
    for b in batches:
        r = render_image(model, None, b, False, mc.config)
    return r
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); r = run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
n = 4 * H * W
print(f"C3: 4 x {W}x{H} = {n} rays in {dt*1e3:.1f} ms ({n/dt/1e6:.2f} M rays/s, {dt/4*1e3:.1f} ms per image), chunk {chunk}; rgb {tuple(r['rgb'].shape)}, depth mean {float(r['depth'].mean()):.3f}")
