"""Where the hash-grid backward spends its time: per-level timing of nlr_grid_encode_backward on the multisample points of a LiDAR
training batch (4096 rays x 64 samples x 7 collinear multisamples)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import config as nconfig, lidar as nlidar, training as ntrain
from nerflidar_hip.gridencoder import GridEncoder
b = nlidar.synthetic_sweep(width=128, seed=0)
batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
n = batch["origins"].shape[0]
td = torch.sort(torch.rand(n, 65, device="cuda") * 1.5 + 0.01, dim=-1)[0]
means, stds = ntrain.cast_contract(batch, td)
pts = means.reshape(-1, 3)
print("points:", pts.shape[0])
for C, L, desired in ((1, 6, 512), (4, 10, 8192)):
    for lv in range(1, L + 1):
        enc = GridEncoder(3, lv, C, base_resolution=16, desired_resolution=None, per_level_scale=float(np.exp2(np.log2(desired / 16) / (L - 1))),
                          log2_hashmap_size=21).cuda()
        y = enc(pts)
        g = torch.randn_like(y)
        for _ in range(2):
            enc.embeddings.grad = None; y = enc(pts); y.backward(g)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            enc.embeddings.grad = None; y = enc(pts); torch.cuda.synchronize(); t1 = time.perf_counter(); y.backward(g); torch.cuda.synchronize(); t0 += (time.perf_counter() - t1) * 0  # noqa
        # time the backward alone with events
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        enc.embeddings.grad = None; y = enc(pts); ev0.record(); y.backward(g); ev1.record(); torch.cuda.synchronize()
        print(f"C={C} levels 0..{lv-1} (finest resolution {enc.grid_sizes[-1].item()}): backward {ev0.elapsed_time(ev1):7.3f} ms")
