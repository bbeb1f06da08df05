"""Where a training step goes on a field that IS a scene: a trained checkpoint (tests/golden/ckpt_trained*, hash maps re-laid out to
full size by `weights.inflate_hashmaps`) is loaded into `TrainableModel`, steps are taken on fresh analytic-scene rays exactly as
`nerflidar_hip.train_scene` takes them, and the torch profiler lists the kernels.  scripts/train_step_bench.py times the same step on
the white-noise field, where every ray is absorbed in the cells around the sensor.
    python scripts/train_scene_profile.py [ckpt_dir=tests/golden/ckpt_trained_c2] [rays=16384] [log2=21]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd")); sys.path.insert(0, ROOT)
import torch
from nerflidar_hip import checkpoints as nckpt, config as nconfig, scene as nscene, training as ntrain, weights as nweights

ck = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ckpt_trained_c2")
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
log2 = int(sys.argv[3]) if len(sys.argv) > 3 else 21
summ = json.load(open(os.path.join(ck, "train_summary.json")))["summary"]
sd, _ = nckpt.load_checkpoint(ck); sd, _ = nckpt.split_state_dict(sd)
mc = nckpt.infer_model_config(sd, nconfig.workload(summ["workload"], summ["log2_hashmap"]))
sd, mc = nweights.inflate_hashmaps(sd, mc, log2)
dev = torch.device("cuda", 0)
tm = ntrain.TrainableModel(mc, fused_mlp=True).to(dev).load_reference(sd)
opt, lr_fn = ntrain.create_optimizer(tm, 0.01, 0.001, 3000, 600)
for g in opt.param_groups:
    g["lr"] = 1e-4                                                # late in a run: the field stays the scene it is
sf = 1.0 / 250.0
def draw(step): return nscene.supervise(nscene.random_lidar_rays(rays, 0, step, dev, 0, sf), 0, sf)
def step(i, batch=None, as_tensors=True):
    return ntrain.training_step(tm, opt, batch if batch is not None else draw(i), train_frac=0.9, randomized=True, hash_decay_mult=0.1, as_tensors=as_tensors)
for i in range(5): out = step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for i in range(K): out = step(100 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K): b = draw(200 + i)
torch.cuda.synchronize(); dt_draw = (time.perf_counter() - t0) / K
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K): out = step(0, b)
torch.cuda.synchronize(); dt_fixed = (time.perf_counter() - t0) / K
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K): step(0, b, as_tensors=False)
torch.cuda.synchronize(); dt_read = (time.perf_counter() - t0) / K
print(f"{summ['workload']} trained checkpoint, maps 2^{log2}, {rays} rays x {mc.level_samples()} samples: {dt*1e3:.1f} ms per step incl. drawing + supervising the rays "
      f"({dt_draw*1e3:.1f} ms of it), {dt_fixed*1e3:.1f} ms on a batch already drawn -> {rays/dt_fixed/1e3:.0f} k rays/s "
      f"({dt_read*1e3:.1f} ms when every step reads its loss terms back); loss {float(out['loss']):.5f}")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(3): step(0, b)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=32, max_name_column_width=90))

# per-level cost of the C = 4 scatter on these points (NLR_DBG_SCATTER_LEVELS: the kernel leaves the other levels out)
if os.environ.get("NLR_SCATTER_LEVELS", "1") != "0":
    from nerflidar_hip import _lib
    L_ = _lib.lib()
    rows = []
    nl = mc.nerf_mlp.grid_num_levels if hasattr(mc.nerf_mlp, "grid_num_levels") else 10
    for mask in [1 << l for l in range(nl)] + [0]:
        L_.nlr_debug_set(_lib.DBG_SCATTER_LEVELS, mask)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for i in range(2): step(0, b)
            torch.cuda.synchronize()
        t = [e for e in prof.key_averages() if "xpair" in e.key]
        rows.append((mask, t[0].self_device_time_total / t[0].count / 1e3 if t else float("nan")))
    L_.nlr_debug_set(_lib.DBG_SCATTER_LEVELS, 0)
    print("nlr_grid_bwd_xpair_kernel<4> per level (ms; levels inside the LDS copy leave at once): " +
          ", ".join(f"{'all' if m == 0 else 'L' + str(m.bit_length() - 1)} {v:.2f}" for m, v in rows))
