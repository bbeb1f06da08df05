"""Diagnostic: per-phase cycle shares of nlr_mlp_kernel from s_memtime stamps (stamp build only; never shipped)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import _lib, config as nconfig, lidar as nlidar, weights as nweights
from nerflidar_hip.models import Model
mc = nconfig.workload("C2"); sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
m = Model(mc, sd, precision=2)
b = nlidar.synthetic_sweep(width=1024, seed=0)
batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
N = 32768
need = _lib.lib().nlr_workspace_bytes(m._handle, N)
m._ws = torch.zeros(need + (4 << 20), dtype=torch.uint8, device="cuda")
al = lambda x: (x + 255) & ~255
off = 0
for S in (64, 64):
    off += 2 * al(N * (S + 1) * 4) + 2 * al(N * S * 4)
S = 128
off += 2 * al(N * (S + 1) * 4) + 2 * al(N * S * 4)   # sdist, tdist, weights, density
off += al(N * S * 40 * 4) + al(N * 32 * 4) + al(N * S * 3 * 4) + al(N * S * 19 * 4)   # feat, enc, rgb, sem
inten_off = off
for _ in range(3): m.render_rays(batch)
torch.cuda.synchronize()
raw = m._ws[inten_off + N * S * 4: inten_off + N * S * 4 + 256 * 128].cpu().numpy().view(np.uint64).reshape(256, 16)  # one row per persistent workgroup (its last tile)
d = np.diff(raw[:, :7].astype(np.int64), axis=1)
clk = (raw[:, 7].astype(np.int64) - raw[:, 0].astype(np.int64)) / np.maximum(raw[:, 9].astype(np.int64) - raw[:, 8].astype(np.int64), 1) * 100.0
print("in-kernel clock (MHz, median over blocks):", np.median(clk))
ok = (d > 0).all(1) & (d < 10**7).all(1)
print("blocks with sane stamps:", ok.sum())
names = ["prologue+trunk+heads", "softmax/stores", "V0 (144 mfma)", "V1 (272 mfma)", "hidden x6 (768 mfma)", "rgb (16 mfma)"]
med = np.median(d[ok], axis=0)
for n, v in zip(names, med): print(f"  {n:24s} {v:10.0f} ticks ({100*v/med.sum():5.1f} %)")
print("  total", med.sum())
