import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import torch
from nerflidar_hip import config as nconfig, lidar as nlidar, weights as nweights, training as ntrain
mc = nconfig.workload(sys.argv[1]); sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
b = nlidar.synthetic_sweep(width=int(sys.argv[2]) // 32 if len(sys.argv) > 2 else 128, seed=0)
batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
n = batch["origins"].shape[0]
batch.update(rgb=torch.rand(n, 3, device="cuda"), depth=torch.rand(n, device="cuda") * 0.5 + 0.05, semantic=torch.randint(0, 19, (n,), device="cuda"))
if mc.config.use_intensity: batch["intensity"] = torch.rand(n, device="cuda")
tm = ntrain.TrainableModel(mc, fused_mlp=True).cuda().load_reference(sd)
opt = torch.optim.Adam(tm.parameters(), lr=1e-3, eps=1e-15)
for _ in range(3): ntrain.training_step(tm, opt, batch)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): ntrain.training_step(tm, opt, batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
