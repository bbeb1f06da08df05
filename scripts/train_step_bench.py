"""Time of one whole-model training step (nerflidar_hip.training.training_step: forward with jitter, losses, backward through the
HIP backward kernels, Adam) on a batch of rays, fused NerfMLP against torch Linear modules.
    python scripts/train_step_bench.py [workload=REF] [rays=4096]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import config as nconfig, lidar as nlidar, weights as nweights, training as ntrain
name = sys.argv[1] if len(sys.argv) > 1 else "REF"
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
mc = nconfig.workload(name)
sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
b = nlidar.synthetic_sweep(width=rays // 32, seed=0)
batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
n = batch["origins"].shape[0]
g = torch.Generator(device="cuda").manual_seed(0)
batch.update(rgb=torch.rand(n, 3, device="cuda", generator=g), depth=torch.rand(n, device="cuda", generator=g) * 0.5 + 0.05,
             semantic=torch.randint(0, 19, (n,), device="cuda", generator=g))
if mc.config.use_intensity:
    batch["intensity"] = torch.rand(n, device="cuda", generator=g)
print(f"workload {name}: {n} rays x {mc.level_samples()} samples, NerfMLP {mc.nerf_mlp.net_depth_viewdirs} x {mc.nerf_mlp.net_width_viewdirs}")
for fused in (False, True):
    tm = ntrain.TrainableModel(mc, fused_mlp=fused).cuda().load_reference(sd)
    opt = torch.optim.Adam(tm.parameters(), lr=1e-3, eps=1e-15)
    # the SAME step on both paths: same weights, deterministic sample positions, no update (lr irrelevant: loss is of the forward).
    # (Round 2 printed the loss after 13 randomized Adam steps drawn from one running RNG stream, i.e. of two different random
    # trajectories: the 3-8 % "gap" at 4 096 rays was jitter noise, 0.1 % at 65 536 rays.  tests/test_training.py pins the step itself
    # on the reference: terms to 2e-4 unfused / 3e-2 fused.)
    from nerflidar_hip import losses as nl
    with torch.no_grad():
        r0, h0 = tm(batch, randomized=False)
        same = float(sum(nl.total_loss(r0, h0, batch).values()))
    torch.manual_seed(0)
    for _ in range(3): ntrain.training_step(tm, opt, batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 10
    for _ in range(K): out = ntrain.training_step(tm, opt, batch, as_tensors=True)   # the loop reads the terms once, after the last step
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"  {'fused MFMA NerfMLP fwd+bwd' if fused else 'torch Linear NerfMLP      '}: {dt*1e3:8.2f} ms per step, {n/dt/1e3:8.1f} k rays/s, loss of the same deterministic step {same:.4f}, after 13 randomized steps {float(out['loss']):.4f}")
    del tm, opt
    torch.cuda.empty_cache()
