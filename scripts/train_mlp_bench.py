"""Step time of the NerfMLP Linear stack in training (row f-3): fused MFMA chains (nlr_mlp_train_forward / _backward + library GEMMs for
the weight gradients) against the torch Linear stack (hipBLASLt), fp32 and bf16 autocast.  Shapes of the reference's training step:
batch_size rays x num_nerf_samples (scripts/run/train_nusc.sh: 4096 rays on one GPU; shipped gin: 32 samples; C2: 128)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import config as nconfig, training
from nerflidar_hip.objects import _pos_enc
torch.manual_seed(0)
dev = "cuda"
F = torch.nn.functional


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for wl, N, S in (("REF", 4096, 32), ("C2", 4096, 128), ("C2", 16384, 128)):
    mc = nconfig.workload(wl, 12)
    cfg = mc.nerf_mlp
    M = N * S
    feats = (torch.randn(N, S, cfg.grid_num_levels * cfg.grid_level_dim, device=dev) * 0.3).requires_grad_(True)
    batch = {"viewdirs": F.normalize(torch.randn(N, 3, device=dev), dim=-1)}
    lv_t = training.TrainableNerfLevel(cfg).to(dev)
    lv_f = training.TrainableNerfLevel(cfg, fused_mlp=True).to(dev)
    lv_f.load_state_dict(lv_t.state_dict())

    def torch_stack(lvl):  # TrainableNerfLevel.forward from the features on
        x = lvl.density_layer(feats)
        outs = [F.softplus(x[..., 0] + cfg.density_bias)]
        if cfg.use_semantic: outs.append(torch.softmax(lvl.sem_layer(x), -1))
        if cfg.use_intensity: outs.append(lvl.intensity_layer(x)[..., 0])
        enc = _pos_enc(batch["viewdirs"], cfg.deg_view)
        h = torch.cat([x, enc[:, None, :].expand(-1, S, -1)], -1)
        inp = h
        for i in range(cfg.net_depth_viewdirs):
            h = F.relu(getattr(lvl, f"lin_second_stage_{i}")(h))
            if i == cfg.skip_layer_dir: h = torch.cat([h, inp], -1)
        outs.append(torch.sigmoid(lvl.rgb_layer(h)))
        return outs

    def step_torch(autocast):
        lv_t.zero_grad(set_to_none=True); feats.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            outs = torch_stack(lv_t)
        sum(o.float().sum() for o in outs).backward()

    def step_fused():
        lv_f.zero_grad(set_to_none=True); feats.grad = None
        o = lv_f._forward_fused(batch, feats)
        sum(v.sum() for v in o.values()).backward()

    def fwd_fused():
        with torch.no_grad():
            lv_f._forward_fused(batch, feats)

    t32, t16, tf, tff = timeit(lambda: step_torch(False)), timeit(lambda: step_torch(True)), timeit(step_fused), timeit(fwd_fused)
    fl = 3 * 2.0 * M * sum(p.numel() for n_, p in lv_t.named_parameters() if n_.endswith("weight") and "encoder" not in n_)
    print(f"{wl} {N} rays x {S} samples ({M / 1e6:.2f} M samples): torch fp32 {t32:.2f} ms, torch bf16 autocast {t16:.2f} ms, "
          f"fused {tf:.2f} ms ({fl / tf / 1e9:.0f} TFLOP/s fwd+bwd; tape pack + forward kernel alone {tff:.2f} ms)")
