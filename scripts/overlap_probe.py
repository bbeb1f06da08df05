"""Probe: do the gather/VALU-bound kernels (encode, prop) overlap with the MFMA kernel when two half-sweeps are
rendered on two streams?  Prints sequential vs two-stream wall time per sweep."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import _lib, config as nconfig, lidar as nlidar, weights as nweights
from nerflidar_hip.models import Model
mc = nconfig.workload("C2"); sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
models = [Model(mc, sd, precision=2) for _ in range(nsplit)]
for m in models[1:]:
    m.tables = models[0].tables
b = nlidar.synthetic_sweep(width=1024, seed=0)
full = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
n = 32768 // nsplit
parts = [{k: v[i * n:(i + 1) * n].contiguous() for k, v in full.items()} for i in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]
def seq():
    for i in range(nsplit): models[0].render_rays(parts[i])
def par():
    for i in range(nsplit):
        with torch.cuda.stream(streams[i]): models[i].render_rays(parts[i])
for name, fn in (("sequential", seq), ("streams", par), ("sequential", seq), ("streams", par)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name:10s} nsplit={nsplit}: {dt*1e3:.2f} ms per sweep -> {32768/dt/1e6:.2f} M rays/s")
