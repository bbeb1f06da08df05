"""Which level body is right for fp16 tables on the 16 x 2 grid (test_fast_level_body... P_F32 / float16 differs)?  Features of the NerfMLP
level from the fast body, the generic body and the CPU oracle on the fp16-rounded table."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import _lib, config as nconfig, lidar as nlidar, weights as nweights
from nerflidar_hip.models import Model, _RAY_KEYS
from oracle import nlr_oracle as orc
mc = nconfig.workload("P_F32", 15)
sd = nweights.synth_state_dict(mc, seed=2, trained_like=True)
sd16 = dict(sd); sd16["nerf_mlp.encoder.embeddings"] = sd["nerf_mlp.encoder.embeddings"].astype(np.float16).astype(np.float32)
model = Model(mc, sd, device="cuda:0", table_dtype=torch.float16)
b = nlidar.synthetic_sweep(width=48, seed=4, beams=nlidar.LIDAR_ANGLES[::2])
batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
n = batch["origins"].shape[0]
r, h = model.render_rays(batch, want_history=True)
tdist = h[-1]["tdist"].contiguous()
S = tdist.shape[1] - 1
rays = _lib.NlrRays(); keep = []
for k in _RAY_KEYS:
    t = batch[k].reshape(n, -1).contiguous().float(); keep.append(t); setattr(rays, k, t.data_ptr())
F = 32
ws = torch.empty(_lib.lib().nlr_workspace_bytes(model._handle, n), dtype=torch.uint8, device="cuda")
feats = {}
for g in (0, 1):
    _lib.lib().nlr_debug_set(0, g)
    f = torch.zeros(n * S, F, device="cuda"); d = torch.empty(n, S, device="cuda")
    rgb = torch.empty(3, n, S, device="cuda"); sem = torch.empty(19, n, S, device="cuda")
    _lib.check(_lib.lib().nlr_mlp_level(model._handle, 2, C.byref(rays), _lib.ptr(tdist), n, 7, 3, None, _lib.ptr(f), _lib.ptr(d), _lib.ptr(rgb), _lib.ptr(sem),
                                        None, _lib.ptr(ws), ws.numel(), None))
    torch.cuda.synchronize(); feats[g] = f.cpu().numpy().reshape(n * S, 16, 2)
_lib.lib().nlr_debug_set(0, 0)
d01 = np.abs(feats[0] - feats[1])
print("fast vs generic: max |d| per level:", [f"{d01[:, l].max():.2e}" for l in range(16)])
bad = np.argwhere(d01 > 0)
print("differing (sample, level, ch):", bad[:10].tolist(), "of", int((d01 > 0).sum()))
if len(bad):
    s_, l_, c_ = bad[0]
    print("values fast / generic:", feats[0][s_, l_], feats[1][s_, l_])
