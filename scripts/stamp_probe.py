"""Diagnostic: per-phase cycle shares of nlr_mlp_kernel from s_memtime stamps (a -DNLR_STAMPS build only; never shipped).
    TAG=stamps EXTRA=-DNLR_STAMPS scripts/diag_build.sh;  NLR_LIB_PATH=nerf-lidar_amd/build/var/lib_stamps.so python scripts/stamp_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import _lib, config as nconfig, lidar as nlidar, weights as nweights
from nerflidar_hip.models import Model
mc = nconfig.workload("C2"); sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
m = Model(mc, sd, precision=2)
b = nlidar.synthetic_sweep(width=1024, seed=0)
batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
for _ in range(200): m.render_rays(batch)   # ~2 s of back-to-back sweeps so that the clock settles
torch.cuda.synchronize()
NS = 24
raw = np.zeros(1024 * NS, np.uint64)
L = _lib.lib()
assert L.nlr_debug_stamps(raw.ctypes.data_as(C.c_void_p), C.c_size_t(raw.size)) == 0
raw = raw.reshape(1024, NS)[:256].astype(np.int64)
names = ["input reads (half A)", "feature split (A)", "D0 (A)", "D2 (A)", "H1 (A)", "H2 (A)", "heads out / hand-over (A)", "-",
         "V0 + V1 (A) [+ compositing pieces]", "trunk + heads + V0 + V1 (B)", "hidden layer pairs", "rgb layer", "rgb tail", "tape padding"]
d = np.diff(raw[:, :15], axis=1)
ok = (d >= 0).all(1) & (d < 10**7).all(1) & (raw[:, 0] > 0)
clk = (raw[:, 14] - raw[:, 0]) / np.maximum(raw[:, 23] - raw[:, 22], 1) * 100.0
print("blocks with sane stamps:", int(ok.sum()), " in-kernel clock (MHz, median):", float(np.median(clk[ok])))
med = np.median(d[ok], axis=0)
for n, v in zip(names, med): print(f"  {n:28s} {v:10.0f} cycles ({100 * v / med.sum():5.1f} %)")
print(f"  total per 256-sample tile     {med.sum():10.0f} cycles")
tot = d[ok].sum(1); us = tot / clk[ok]
q = lambda a: " / ".join(f"{v:.1f}" for v in np.percentile(a, [0, 5, 50, 95, 100]))
print(f"  per-workgroup spread (min / p5 / median / p95 / max): cycles per tile {q(tot / 1e3)} k, clock {q(clk[ok])} MHz, time per tile {q(us)} us")
