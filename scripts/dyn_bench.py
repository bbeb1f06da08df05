"""Throughput of the dynamic-object path (row f-1) next to the static fused path, REF architecture, full sweep."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerflidar_hip import config as nconfig, lidar as nlidar, weights as nweights, objects as nobj
from nerflidar_hip.models import Model
n_tracks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mc = nconfig.workload("REF"); sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
b = nlidar.synthetic_sweep(width=1024, seed=0); n = b["origins"].shape[0]
b["timestamp"] = nobj.synthetic_timestamps(n, 0)
tracks = nobj.synthetic_tracks(b, n_tracks, 20, 0, size=(0.02, 0.01, 0.008), depth=(0.02, 0.2))  # ~5 x 2.5 x 2 m at scale 1/250
names = ["vehicle.car", "vehicle.truck", "vehicle.bus.rigid"] * n_tracks
cids = sorted({nobj.query_class(c) for c in names[:n_tracks]})
sd.update(nweights.synth_object_state_dict({c: nconfig.obj_mlp_config(c, 128, 21) for c in cids}, n_tracks, seed=0))
mc.config.instance_obj = True
dyn = nobj.DynamicModel(mc, sd, tracks, names[:n_tracks], precision=2)
static = Model(nconfig.workload("REF"), sd, precision=2)
batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
runs = (("static fused (nlr_render_rays)", static.render_rays),
        (f"dynamic, {n_tracks} tracks, on the device (nlr_render_rays_dynamic)", dyn.render_rays),
        (f"dynamic, {n_tracks} tracks, stage loop + torch ObjMLPs (round 1)", dyn.render_rays_torch))
for name, fn in runs:
    for _ in range(3): r, h = fn(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): r, h = fn(batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    extra = f", samples in boxes per level {[int(x['obj_mask'].sum()) for x in h]}" if "obj_mask" in h[0] else ""
    print(f"{name}: {dt*1e3:.2f} ms per sweep, {n/dt/1e6:.2f} M rays/s{extra}")
if len(sys.argv) > 2:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3): dyn.render_rays(batch)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
    print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=12, max_name_column_width=60))
