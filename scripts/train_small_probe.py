"""30 training steps at 2^13-entry tables (the committed trained fixture's size) for a rocprofv3 kernel trace: why does a step take
~150 ms at 8 192 rays when full-size tables take ~30 ms?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd"))
import torch
from nerflidar_hip import config as nconfig, scene as nscene, training as ntrain
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 13
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
mc = nconfig.workload("REFI", lg)
tm = ntrain.TrainableModel(mc, fused_mlp=True).cuda()
opt, lr_fn = ntrain.create_optimizer(tm, max_steps=1000, lr_delay_steps=10)
for step in range(1, 31):
    if step == 11:
        torch.cuda.synchronize(); t0 = time.time()
    batch = nscene.supervise(nscene.random_lidar_rays(rays, 0, step, "cuda"))
    ntrain.training_step(tm, opt, batch, train_frac=0.5)
torch.cuda.synchronize()
print(f"log2 {lg}, {rays} rays: {(time.time() - t0) / 20 * 1e3:.1f} ms per step")
