#!/usr/bin/env python3
"""Diagnostic: at which problem size does PREC_F32 / PREC_MIXED leave PREC_FAST (C2 architecture)?  Prints the density / depth gap per (rays, table size)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import _lib, config as nconfig, lidar as nlidar, weights as nweights
from nerflidar_hip.models import Model
dev = torch.device("cuda:0")
for log2 in (14, None):
    mc = nconfig.workload("C2", log2)
    sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
    models = {p: Model(mc, sd, device=dev, precision=p) for p in (_lib.PREC_F32, _lib.PREC_MIXED, _lib.PREC_FAST)}
    for width in (2, 8, 32, 128, 256, 512, 1024):
        full = nlidar.synthetic_sweep(width=width, seed=0)
        batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in full.items()}
        res = {}
        for p, m in models.items():
            r, h = m.render_rays(batch, compute_extras=True, scale_factor=1 / 250, want_history=True)
            torch.cuda.synchronize()
            res[p] = (r["depth"].cpu().numpy(), h[-1]["density"].cpu().numpy(), h[1]["density"].cpu().numpy())
        n = full["origins"].shape[0]
        line = f"log2 {log2} rays {n:6d}:"
        for p, name in ((_lib.PREC_F32, "F32"), (_lib.PREC_MIXED, "MIXED")):
            dd = np.abs(res[p][1] - res[_lib.PREC_FAST][1])
            bad_rays = np.where(dd.max(-1) > 1.0)[0]
            line += f"  {name}: depth L1 {np.abs(res[p][0] - res[_lib.PREC_FAST][0]).mean():.2e} density |d| mean {dd.mean():.2e} max {dd.max():.2e} bad rays {len(bad_rays)}"
            if len(bad_rays):
                line += f" (first {bad_rays[:4]}, last {bad_rays[-2:]})"
        print(line, flush=True)
