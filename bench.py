#!/usr/bin/env python3
"""bench.py -- LiDAR rays/s of the MI355X-native render hot path (BASELINE.json metric).

A step = one pass of the whole hot path (all proposal levels + NerfMLP level + compositing + the packed range-image
record per ray, outputs resident in HBM) over one synthetic nuScenes-shaped sweep of 32 beams x 1024 azimuth columns
(32 768 rays), configuration C2 of BASELINE.json: (64, 64, 128) samples per ray, 8x256 view MLP, semantic + intensity
heads, full-size hash tables (229 MiB NerfMLP table, fp32).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (BASELINE config C4): the ONE 32 x 1024 sweep is split into N azimuth sectors, GPU p renders sector p and one
all-gather (RCCL over xGMI, issued on a side stream under the next sweep's first kernels) puts the packed [1024, 32, 7]
range image on every rank: "scaling": "strong" (total work fixed).  `--scaling weak` renders 1024*N columns instead
(per-GPU work fixed).  Without a launcher environment (RANK/WORLD_SIZE unset) bench.py starts its own N ranks as child
processes (`python -m torch.distributed.run`, rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU,
and forwards their exit code; under `torch.distributed.run` it is a rank.

Prints ONE JSON line on rank 0; carries `roofline` for the dominant kernel (nlr_mlp_kernel, timed with HIP events on its
launch stream inside the timed region), `cpu_baseline` (the CPU oracle on a bounded sample) and `accuracy` (the timed
precision against the oracle on that same sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
H_BEAMS, W_COLS = 32, 1024
PMC_PROFILE = "r04_pmc_traffic.json"  # scripts/pmc_traffic.sh on the final binary of the round
MFMA_BF16_SUSTAINED_TFLOPS = 1776.0  # measured, scripts/micro/mfma_shape4 sustained 10 s (profiles/r04_power_trace.txt)
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"
KNAMES = ["resample", "prop", "encode", "direnc", "mlp", "composite"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--width", type=int, default=W_COLS, help="azimuth columns of the sweep (1024 = BASELINE config; 1100 = the reference's own "
                    "sweep, ZI/lidar_utils.py:122-134: 35 200 rays)")
    ap.add_argument("--precision", type=int, default=2, help="0 f32, 1 mixed, 2 fast (default; see include/nerflidar_hip.h)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = ONE 32x1024 sweep in N azimuth sectors (C4); weak = 1024*N columns")
    ap.add_argument("--log2-hashmap", type=int, default=None, help="shrink the hash tables (debug only)")
    ap.add_argument("--cpu-rays", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--table-dtype", choices=["f32", "f16"], default="f32", help="hash-table storage; f32 is the benchmark "
                    "configuration, f16 is what the reference uses under autocast (Z/gridencoder/grid.py:43-44)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: self-launch, initialise RCCL and run the collective also with one rank")
    ap.add_argument("--emulate-world", type=int, default=0, help="diagnostic, one GPU: time the step one rank of a P-way azimuth split "
                    "runs (sector 0 of P, no collective); the JSON line says so and is not a benchmark result")
    ap.add_argument("--chunk", type=int, default=0, help="rays per nlr_render_rays call (0 = the whole sector at once)")
    ap.add_argument("--history", action="store_true", help="also write the per-sample heads of the last level (ray_history)")
    ap.add_argument("--graph", action="store_true", help="replay the sweep from a HIP graph (models.CapturedRender), one graph per tile buffer: "
                    "one graph launch per step instead of ~10 kernel launches; the per-kernel HIP events are not part of a captured "
                    "sweep, so kernel_ms / roofline are not reported in this mode")
    ap.add_argument("--ckpt", default=None, help="render a TRAINED checkpoint (reference format, nerflidar_hip.checkpoints) instead of the seeded "
                    "synthetic weights; the architecture comes from the file, sampling counts from --workload (or the train_summary.json "
                    "`python -m nerflidar_hip.train_scene` leaves beside it)")
    ap.add_argument("--weight-scale", type=float, default=None, help="diagnostic (profiles/r04_power_trace.txt): multiply every synthetic "
                    "parameter by this factor; 0 renders all-zero weights and tables - the same instruction stream with no operand toggling")
    ap.add_argument("--mlp-workgroups", type=int, default=0, help="diagnostic (scripts/mlp_cu_sweep.sh): cap the persistent MLP grid at n workgroups "
                    "(nlr_debug_set; recorded in the line as `debug_switches`)")
    ap.add_argument("--inflate-log2", type=int, default=None, help="with --ckpt: re-lay the checkpoint's hash maps out at 2^N rows per hashed "
                    "level (nerflidar_hip.weights.inflate_hashmaps: the same field bit for bit, with the footprint and the access pattern "
                    "of the larger maps; 21 = the full-size configuration)")
    ap.add_argument("--no-trained-leg", action="store_true", help="skip the `trained_scene` object (the default one-GPU run also renders "
                    "the committed trained checkpoint, inflated to full-size maps, for a second, scene-shaped measurement)")
    ap.add_argument("--static-origin", action="store_true", help="render the SAME sweep every step (round 1-3 behaviour).  Default: step i "
                    "renders synthetic_sweep(sweep_idx = i mod 64) from pre-uploaded ray batches, as a LiDAR replay moves the sensor every "
                    "sweep (Z/train.py:484-485 counts rays of distinct batches), so the caches hold what a replay leaves, not the "
                    "previous step's identical access pattern")
    ap.add_argument("--ray-groups", choices=["auto", "rays", "samples"], default="auto", help="A/B: which samples share a wave of the encode kernels. auto "
                    "(default): decided per level on the device from the coherence of adjacent rays' samples; rays: always 8 adjacent rays x one "
                    "sample index; samples: always 8 consecutive samples of one ray (rounds 1-3) (nlr_debug_set NLR_DBG_RAY_GROUPS)")
    ap.add_argument("--selftest-cpu", action="store_true", help="launcher + partition + collective logic on CPU (gloo) with a "
                    "stand-in renderer; for tests/, measures nothing")
    ap.add_argument("--selftest-hw", type=int, nargs=2, default=[4, 64], metavar=("H", "W"),
                    help="beams x azimuth columns of the --selftest-cpu sweep (32 1100 = the reference's sweep, ZI/lidar_utils.py:122-134: "
                         "1100 columns over 8 ranks is 138 per rank with 4 padded columns)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """Start the N ranks as children of this (GPU-free) process and hand back their exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def host_cores():
    """CPU share of this process: affinity, capped by the cgroup quota, and by 16 (the one-GPU box's share)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(mc, sd, batch_np, idx, threads, passes=3):
    """The oracle (a port: PyTorch-CPU restatement pinned on reference fixtures) timed on this host's cores.  Returns the
    cpu_baseline object and the oracle's renderings of the sampled rays (for `accuracy`)."""
    import numpy as np
    import torch
    from oracle import nlr_oracle as orc
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    n_rays = len(idx)
    b = {k: torch.from_numpy(np.ascontiguousarray(v[idx])) for k, v in batch_np.items()}
    enc = orc.make_encoders(sd, mc)
    sdt = orc.to_torch_sd(sd)
    chunk = 4096  # chunks of 4 096 rays
    n_pass, passes = passes, []
    with torch.no_grad():
        orc.model_forward(sd, mc, {k: v[:256] for k, v in b.items()}, encoders=enc, sd_t=sdt)  # warm-up
        for _ in range(n_pass):  # median of 3 passes over the sample (median of 5 over all 32 768 rays = ~150 s; bounded here)
            outs = []
            t0 = time.perf_counter()
            for i in range(0, n_rays, chunk):
                outs.append(orc.model_forward(sd, mc, {k: v[i:i + chunk] for k, v in b.items()}, encoders=enc, sd_t=sdt)[0][-1])
            passes.append(time.perf_counter() - t0)
    dt = sorted(passes)[len(passes) // 2]
    ref = {k: torch.cat([o[k] for o in outs]).numpy() for k in ("depth", "intensity", "semantic") if k in outs[0]}
    return dict(value=n_rays / dt, unit="rays/s", cores=threads, kind="port",
                sample=f"{n_rays} rays of the same sweep (every {len(batch_np['origins']) // n_rays}th ray), same weights, chunks of {chunk}, "
                       f"median of 3 passes after a 256-ray warm-up ({', '.join(f'{p_:.1f}' for p_ in passes)} s), fp32 PyTorch-CPU + OpenMP C grid "
                       f"oracle; SURVEY 8d's protocol (all 32 768 rays, median of 5) is ~150 s of CPU work and is bounded to this sample so that "
                       f"the default run stays within minutes"), ref


def accuracy(ref, r, idx):
    """The timed GPU outputs against the oracle on the sampled rays (north_star: depth / intensity within 1e-3, labels exact).
    Besides L1 / percentiles the OUTLIER FRACTIONS are reported: on this synthetic scene (white-noise tables under a x1500 density gain)
    a few rays per thousand are ill-conditioned - the reference's own fp32 arithmetic is as far from a float64 evaluation of its
    algorithm on them as the GPU is (profiles/r03_parity_tail.txt) - so a maximum says nothing, the fraction beyond a threshold does."""
    import numpy as np
    g = {k: r[k].detach().cpu().numpy()[idx] for k in ("depth", "intensity", "semantic", "labels") if k in r}
    d = np.abs(g["depth"] - ref["depth"])
    out = dict(rays=int(len(idx)), depth_l1=float(d.mean()), depth_p95=float(np.percentile(d, 95)), depth_p99=float(np.percentile(d, 99)),
               depth_max=float(d.max()), depth_frac_gt_1e3=float(np.mean(d > 1e-3)), depth_frac_gt_1e2=float(np.mean(d > 1e-2)))
    if "intensity" in ref and "intensity" in g:
        di = np.abs(g["intensity"] - ref["intensity"])
        out.update(intensity_l1=float(di.mean()), intensity_max=float(di.max()), intensity_frac_gt_1e3=float(np.mean(di > 1e-3)))
    if "semantic" in ref and "labels" in g:
        sr = np.sort(ref["semantic"], axis=-1)
        out["label_mismatches"] = int((g["labels"] != ref["semantic"].argmax(-1)).sum())
        out["min_top2_margin"] = float((sr[:, -1] - sr[:, -2]).min())
    return out


TRAINED_CKPT = os.path.join(ROOT, "tests", "golden", "ckpt_trained_c2")


def trained_scene_leg(dev, precision, threads, steps=20, warmup=5, oracle_rays=2048):
    """Second measurement of the default one-GPU run: the SAME configuration (C2: (64, 64, 128) samples, 8x256 NerfMLP + semantic +
    intensity heads, full-size fp32 maps, one 32 x 1024 sweep per step, moving origin) on a TRAINED field instead of seeded white noise.
    The committed checkpoint (tests/golden/ckpt_trained_c2: `python -m nerflidar_hip.train_scene` on the analytic street scene, small
    hash maps so that it fits the repository) is re-laid out on 2^21-row maps by `weights.inflate_hashmaps` - the same function value
    for value, with the 229 MiB footprint and the scattered fine-level accesses of the full-size configuration.  Samples follow the
    scene's surfaces, so the gather kernels see the line traffic of a real replay (the white-noise scene terminates every ray within
    a few cells of the sensor: cache-friendly far beyond any real scene).  Returns the `trained_scene` object, or None without the file."""
    import ctypes as C
    import numpy as np
    import torch
    from nerflidar_hip import _lib, checkpoints as nckpt, config as nconfig, flops as nflops, lidar as nlidar, weights as nweights
    from nerflidar_hip.models import Model
    if not os.path.isdir(TRAINED_CKPT):
        return None
    summ = json.load(open(os.path.join(TRAINED_CKPT, "train_summary.json")))["summary"]
    sd_all, ck_step = nckpt.load_checkpoint(TRAINED_CKPT)
    sd, _ = nckpt.split_state_dict(sd_all)
    mc = nckpt.infer_model_config(sd, nconfig.workload(summ["workload"], summ["log2_hashmap"]))
    for prefix, cfg_ in nweights.mlp_names(mc):
        sd[f"{prefix}.encoder.offsets"], sd[f"{prefix}.encoder.grid_sizes"], _ = nweights.grid_layout(cfg_)
    sd, mc = nweights.inflate_hashmaps(sd, mc, 21)
    model = Model(mc, sd, device=dev, precision=precision)
    n_sw = 32
    secs = [nlidar.synthetic_sweep(width=W_COLS, seed=0, sweep_idx=100 + si) for si in range(n_sw)]  # sensor positions no training ray used
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in secs[0].items()}
    origins = [torch.from_numpy(np.ascontiguousarray(s_["origins"])).to(dev) for s_ in secs]
    n_rays = batch["origins"].shape[0]
    tile = torch.zeros(W_COLS, H_BEAMS, 7, device=dev)
    sf = 1.0 / 250.0
    L = _lib.lib()

    def step(i):
        batch["origins"] = origins[i % n_sw]
        return model.render_rays(batch, compute_extras=True, scale_factor=sf, packed=tile)[0]

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    _lib.check(L.nlr_profile_begin_kinds(model._handle, (1 << 4) | (1 << 2)), "nlr_profile_begin_kinds")
    t0 = time.perf_counter()
    for i in range(steps):
        r = step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = (C.c_float * _lib.NLR_K_COUNT)()
    cnt = (C.c_uint32 * _lib.NLR_K_COUNT)()
    _lib.check(L.nlr_profile_end(model._handle, _lib.current_stream(), ms, cnt), "nlr_profile_end")
    ms2 = (C.c_float * _lib.NLR_K_COUNT)()
    cnt2 = (C.c_uint32 * _lib.NLR_K_COUNT)()
    _lib.check(L.nlr_profile_begin(model._handle), "nlr_profile_begin")
    for i in range(4):
        step(warmup + steps + i)
    torch.cuda.synchronize()
    _lib.check(L.nlr_profile_end(model._handle, _lib.current_stream(), ms2, cnt2), "nlr_profile_end")
    kern = {}
    for k_ in range(_lib.NLR_K_COUNT):
        m_, c_ = (ms[k_], cnt[k_]) if k_ in (2, 4) else (ms2[k_], cnt2[k_])
        kern[KNAMES[k_]] = round(m_ / c_, 4) if c_ else 0.0
    last = (warmup + steps - 1) % n_sw
    r = step(last)            # the sweep the accuracy leg checks (the 4 untimed sweeps above moved on)
    torch.cuda.synchronize()
    fl = 2.0 * nflops.macs_per_sample(mc.nerf_mlp) * n_rays * mc.level_samples()[-1]
    idx = np.linspace(0, n_rays - 1, oracle_rays).astype(np.int64)
    _, ref = cpu_baseline(mc, sd, secs[last], idx, threads, passes=1)
    return {"value": n_rays * steps / dt, "unit": "rays/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "kernel_ms": kern, "mlp_roofline_frac": (fl / (kern["mlp"] * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS) if kern["mlp"] else None,
            "accuracy": accuracy(ref, r, idx),
            "workload": f"C2 architecture and sampling on the trained analytic street scene: checkpoint step {ck_step} "
                        f"({os.path.relpath(TRAINED_CKPT, ROOT)}, trained with 2^{summ['log2_hashmap']}-row maps), hash maps inflated to 2^21 rows "
                        "(same field, full-size footprint), held-out sensor positions, moving origin",
            "scene_fit": summ.get("held_out_sweep")}


def pmc_traffic(profile_path, binary_sha, sources_sha, kernel, applicable=True):
    """HBM bytes per launch of `kernel` from a committed rocprofv3 --pmc profile (scripts/pmc_traffic.sh), or (None, why not).
    Quoted only when the profile was taken on THIS binary (the hash compiled into libnerflidar_hip.so) and the binary is not stale
    against the sources in the tree."""
    if binary_sha != sources_sha:
        return None, (f"libnerflidar_hip.so was built from kernel source {binary_sha[:12]} but the tree holds {sources_sha[:12]}: "
                      "stale binary, traffic withheld")
    try:
        tj = json.load(open(profile_path))
    except Exception:
        return None, f"no PMC profile at {os.path.relpath(profile_path, ROOT)}"
    if tj.get("kernel_source_sha") != binary_sha:
        return None, (f"{os.path.relpath(profile_path, ROOT)} was measured on kernel source {str(tj.get('kernel_source_sha'))[:12]}, "
                      f"this binary is {binary_sha[:12]}: traffic withheld")
    if not applicable:
        return None, "the committed profile is of the default one-GPU C2 sweep, not of this configuration"
    for k, v in tj["kernels"].items():
        if kernel in k and "hbm_read_bytes_corrected" in v:
            return v["hbm_read_bytes_corrected"] + v.get("hbm_write_bytes", 0.0), (
                f"bytes per launch, {os.path.relpath(profile_path, ROOT)} (separate rocprofv3 --pmc passes: FETCH_SIZE x2 gfx950 "
                "correction + WRITE_SIZE)")
    return None, f"{kernel} not in {os.path.relpath(profile_path, ROOT)}"


def selftest_cpu(args):
    """Partition + collective + reassembly with a stand-in renderer on CPU ranks (gloo).  No product code is timed."""
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd"))
    from nerflidar_hip import lidar as nlidar, sharding
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W = args.selftest_hw[0], args.selftest_hw[1] * (world if args.scaling == "weak" else 1)
    full = nlidar.synthetic_sweep(width=W, seed=0, beams=nlidar.LIDAR_ANGLES[:H])

    def fake(b, packed=None):
        d = b["directions"]
        depth = (d * torch.tensor([1.0, 2.0, 3.0])).sum(-1)
        return dict(depth=depth, intensity=depth * 0.5, acc=torch.ones_like(depth), rgb=d.abs(), labels=(depth.abs() * 7).to(torch.int32) % 19)

    g = sharding.SweepGatherer(H, W, "cpu", force=args.force_dist)
    for i in range(args.steps):
        img = sharding.render_sweep_sharded(fake, full, H, W, "cpu", gatherer=g, index=i)
    one = sharding.pack_tile(fake({k: torch.from_numpy(v) for k, v in full.items()}), H, W)
    ok = bool(torch.equal(img, one))
    seen = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "steps": args.steps, "scaling": args.scaling, "image_equal": ok,
                          "shape": list(img.shape), "ranks_seen": int(seen.item()), "columns_per_rank": g.wp,
                          "padded_columns": g.wp * world - W}))
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    args = parse_args()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and (args.gpus > 1 or args.force_dist):
        sys.exit(self_launch(args))  # nothing above this line imports torch or touches the GPU
    if args.selftest_cpu:
        sys.exit(selftest_cpu(args))

    sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd"))
    sys.path.insert(0, ROOT)
    import ctypes as C
    import numpy as np
    import torch
    from nerflidar_hip import _lib, buildinfo, config as nconfig, flops as nflops, lidar as nlidar, sharding, weights as nweights
    from nerflidar_hip.models import Model

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = launched and (world > 1 or args.force_dist)
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    if args.mlp_workgroups:
        _lib.check(_lib.lib().nlr_debug_set(_lib.DBG_MLP_WORKGROUPS, args.mlp_workgroups))
    tdt = torch.float16 if args.table_dtype == "f16" else torch.float32
    ckpt_note = None
    if args.ckpt:
        from nerflidar_hip import checkpoints as nckpt
        wl, lg = args.workload, args.log2_hashmap
        summ_path = os.path.join(args.ckpt if os.path.isdir(args.ckpt) else os.path.dirname(args.ckpt), "train_summary.json")
        if os.path.exists(summ_path):
            summ = json.load(open(summ_path))["summary"]
            wl, lg = summ["workload"], summ["log2_hashmap"]
            ckpt_note = {k: summ[k] for k in ("workload", "log2_hashmap", "steps", "rays_per_step", "held_out_sweep") if k in summ}
        args.workload = wl
        sd_all, ck_step = nckpt.load_checkpoint(args.ckpt)
        sd, _ignored = nckpt.split_state_dict(sd_all)
        mc = nckpt.infer_model_config(sd, nconfig.workload(wl, lg))
        for prefix, cfg_ in nweights.mlp_names(mc):  # buffers the oracle reads (re-derived from the config, grid.py:137-142)
            sd[f"{prefix}.encoder.offsets"], sd[f"{prefix}.encoder.grid_sizes"], _ = nweights.grid_layout(cfg_)
        ckpt_note = dict(ckpt_note or {}, path=os.path.relpath(args.ckpt, ROOT), restored_step=ck_step)
        if args.inflate_log2:
            sd, mc = nweights.inflate_hashmaps(sd, mc, args.inflate_log2)
            ckpt_note["hash_maps_inflated_to_log2"] = args.inflate_log2
            lg = args.inflate_log2
        args.log2_hashmap = None if lg in (None, 21) else lg
    else:
        mc = nconfig.workload(args.workload, args.log2_hashmap)
        sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
        if args.weight_scale is not None:
            sd = {k: (v * np.float32(args.weight_scale) if v.dtype == np.float32 else v) for k, v in sd.items()}
    model = Model(mc, sd, device=dev, precision=args.precision, table_dtype=tdt)
    _lib.lib().nlr_debug_set(_lib.DBG_RAY_GROUPS, {"auto": 0, "rays": 1, "samples": 2}[args.ray_groups])
    width = args.width * (world if args.scaling == "weak" else 1)
    emul = args.emulate_world if (args.emulate_world > 1 and world == 1) else 0
    n_sweeps = 1 if args.static_origin else 64
    secs = []
    for si in range(n_sweeps):  # the sweeps of a replay differ in the sensor position only (lidar.synthetic_sweep: origin = o0(sweep_idx))
        full = nlidar.synthetic_sweep(width=width, seed=0, sweep_idx=si)
        sec, wp = nlidar.azimuth_sector(full, H_BEAMS, width, rank, emul or world)
        secs.append(sec)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in secs[0].items()}
    origins = [torch.from_numpy(np.ascontiguousarray(s_["origins"])).to(dev) for s_ in secs]  # pre-uploaded: resident when the timed region starts
    n_rays = H_BEAMS * wp
    sf = 1.0 / 250.0
    if emul:
        width = wp  # the "image" of the emulation is the rank's own tile
    gat = sharding.SweepGatherer(H_BEAMS, width, dev, force=args.force_dist)
    last = {}

    caps = {}
    if args.graph:
        if args.chunk:
            raise SystemExit("--graph renders the sector in one call (no --chunk)")
        from nerflidar_hip.models import CapturedRender
        for b_ in range(2 if gat.collective else 1):  # SweepGatherer double-buffers the tile when a collective reads it
            caps[b_] = CapturedRender(model, batch, compute_extras=True, scale_factor=sf, want_history=args.history, packed=gat.tiles[b_])

    def step(i):
        tile = gat.tile(i if gat.collective else 0)  # without a collective nothing reads the tile behind the render: one buffer
        last["sweep"] = i % n_sweeps
        if caps:
            if n_sweeps > 1:
                batch["origins"].copy_(origins[i % n_sweeps])  # a captured sweep reads its static input buffers: refill in place
            last["r"] = caps[(i & 1) if len(caps) == 2 else 0].replay()
            gat.submit(i)
            return
        batch["origins"] = origins[i % n_sweeps]
        if args.chunk and args.chunk < n_rays:  # chunked like the reference's driver; records land in ray order
            flat = torch.empty(n_rays, 7, device=dev)
            for a in range(0, n_rays, args.chunk):
                model.render_rays({k: v[a:a + args.chunk] for k, v in batch.items()}, compute_extras=True, scale_factor=sf,
                                  want_history=args.history, packed=flat[a:a + args.chunk])
            tile.copy_(flat.reshape(H_BEAMS, wp, 7).permute(1, 0, 2))
        else:
            last["r"], _ = model.render_rays(batch, compute_extras=True, scale_factor=sf, want_history=args.history, packed=tile)
        gat.submit(i)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    L = _lib.lib()
    # HIP events on the launch stream inside the timed region, around the two kernels the roofline objects are about (nlr_mlp_kernel,
    # nlr_encode8_kernel): 4 event records per step.  Bracketing all ten launches of a sweep costs 0.1 ms of a 7 ms step
    # (profiles/r03_emulated_sector_steps.txt), so the other kernels' durations come from a short untimed pass after the loop.
    K_MLP, K_ENC = 4, 2  # NLR_K_MLP, NLR_K_ENCODE (include/nerflidar_hip.h)
    prof = not os.environ.get("NLR_BENCH_NOPROF") and not caps  # (diagnostic switch: what do the HIP events themselves cost?)
    if prof:
        _lib.check(L.nlr_profile_begin_kinds(model._handle, (1 << K_MLP) | (1 << K_ENC)), "nlr_profile_begin_kinds")
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    img = gat.image((args.warmup + args.steps - 1) if gat.collective else 0)
    barrier()
    dt = time.perf_counter() - t0
    ms = (C.c_float * _lib.NLR_K_COUNT)()
    cnt = (C.c_uint32 * _lib.NLR_K_COUNT)()
    _lib.check(L.nlr_profile_end(model._handle, _lib.current_stream(), ms, cnt), "nlr_profile_end")
    if prof:  # untimed: every kernel bracketed, a few sweeps, for the `kernel_ms` breakdown of the launches outside the roofline objects
        ms2 = (C.c_float * _lib.NLR_K_COUNT)()
        cnt2 = (C.c_uint32 * _lib.NLR_K_COUNT)()
        _lib.check(L.nlr_profile_begin(model._handle), "nlr_profile_begin")
        for i in range(min(args.steps, 8)):
            step(args.warmup + args.steps + i)
        barrier()
        _lib.check(L.nlr_profile_end(model._handle, _lib.current_stream(), ms2, cnt2), "nlr_profile_end")
        for k_ in range(_lib.NLR_K_COUNT):
            if k_ not in (K_MLP, K_ENC):
                ms[k_], cnt[k_] = ms2[k_], cnt2[k_]
    ag_ms = None
    ranks_seen, rank_ms = 1, [dt / args.steps * 1e3]
    if use_dist:
        # evidence that the collective library really spans `world` ranks: a SUM of ones and every rank's own step time
        own = torch.tensor([dt], device=dev, dtype=torch.float64)
        t = own.clone()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        one = torch.ones(1, device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(one, op=torch.distributed.ReduceOp.SUM)
        every = [torch.zeros_like(own) for _ in range(world)]
        torch.distributed.all_gather(every, own)
        ranks_seen, rank_ms = int(round(one.item())), [float(x.item()) / args.steps * 1e3 for x in every]
        dt = float(t.item())
        # the collective alone (not overlapped), for the record
        reps = max(args.steps, 20)
        barrier()
        ta = time.perf_counter()
        for _ in range(reps):
            torch.distributed.all_gather_into_tensor(gat.images[0], gat.tiles[0])
        barrier()
        ag_ms = (time.perf_counter() - ta) / reps * 1e3
    assert img.shape == (width, H_BEAMS, 7)

    if rank == 0:
        rays_total = n_rays * world * args.steps
        kern = {KNAMES[i]: (ms[i] / cnt[i] if cnt[i] else 0.0) for i in range(_lib.NLR_K_COUNT)}
        # dominant kernel: nlr_mlp_kernel.  Algorithmic FLOPs per launch = 2 * MACs/sample (SURVEY 8d:
        # 657 408 for the 8x256 NerfMLP + heads) * samples per launch (rays * 128).
        S_last = mc.level_samples()[-1]
        launch_rays = min(args.chunk, n_rays) if args.chunk else n_rays
        fl_launch = 2.0 * nflops.macs_per_sample(mc.nerf_mlp) * launch_rays * S_last
        mlp_s = kern["mlp"] * 1e-3
        achieved = fl_launch / mlp_s / 1e12 if mlp_s > 0 else 0.0
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes (scripts/pmc_traffic.sh), which record the source hash
        # COMPILED INTO the binary they measured: a profile of other code is not a measurement of this binary.
        bsha, ssha = buildinfo.binary_sha(), buildinfo.kernel_source_sha()
        prof = os.path.join(ROOT, "profiles", PMC_PROFILE)
        plain = (world == 1 and args.workload == "C2" and not args.chunk and not emul and args.width == W_COLS and args.table_dtype == "f32"
                 and args.precision == 2 and args.log2_hashmap is None and not args.ckpt and not args.static_origin and args.weight_scale is None
                 and not args.mlp_workgroups and args.ray_groups == "auto")
        traffic, tnote = pmc_traffic(prof, bsha, ssha, "nlr_mlp_kernel", plain)
        # second ceiling: the gather side, priced in bytes that really cross the L2's memory-side port.  achieved = PMC counter
        # bytes per launch (FETCH_SIZE [x 1 for these random-line kernels, see scripts/pmc_traffic.sh] + WRITE_SIZE; the committed profile of THIS binary) / the
        # kernel's launch duration measured here with HIP events; peak = 8 TB/s HBM (the guide's measured random-gather rates from tables
        # in the Infinity Cache are 7.4-8.6 TB/s, streamed HBM 6.0-6.3 TB/s).  The algorithmic gather bytes (samples x 7 multisamples x L
        # levels x 8 corners x C channels x 4 B) are kept as a note: most of them are served by the scalar cache, by lanes sharing a
        # look-up and by L1/L2 hits, so their rate is not a roofline fraction.
        ncfg = mc.nerf_mlp
        tb = 4 if args.table_dtype == "f32" else 2
        g_bytes = float(launch_rays) * S_last * 7 * ncfg.grid_num_levels * 8 * ncfg.grid_level_dim * tb
        HBM_PEAK_GBS = 8000.0

        def gather_roofline(kernel, key, alg_bytes, launches_per_step):
            t_s = kern[key] * 1e-3
            tr, note = pmc_traffic(prof, bsha, ssha, kernel, plain)
            ach = (tr / t_s / 1e9) if (tr is not None and t_s > 0) else None
            return {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (ach / HBM_PEAK_GBS) if ach is not None else None, "traffic": tr, "traffic_note": note,
                    "launch_ms": round(kern[key], 4), "launches_per_step": launches_per_step,
                    "algorithmic_gather_bytes": alg_bytes, "algorithmic_gather_GBps": (alg_bytes / t_s / 1e9) if t_s > 0 else None,
                    "note": "achieved = PMC bytes past L2 per launch (FETCH_SIZE as tallied: for random 64-byte lines it equals the line bytes, "
                            "calibrated with scripts/micro/gather_rand.hip, profiles/r04_fetch_size_calibration.txt - the x2 gfx950 correction "
                            "applies to wide coalesced streams only - + WRITE_SIZE; Infinity-Cache hits included: the counter sits on the L2's "
                            "fabric side) / HIP-event launch duration, against 8 TB/s.  The guide's gather ceilings are 7.4-8.6 TB/s for "
                            "1 152-byte rows; for random 64-byte LINES this chip delivers 3.5-4.2 TB/s (55-66 G lines/s, "
                            "profiles/r04_gather_rand_microbench.txt).  "
                            "algorithmic_gather_* counts every corner read of every multisample and is NOT a roofline figure (scalar-cache "
                            "path, shared look-ups, L1/L2 hits)"}

        samples = mc.level_samples()
        props = [mc.prop_cfg(i) for i in range(mc.num_levels - 1)]
        p_bytes = (sum(float(launch_rays) * samples[i] * 7 * c_.grid_num_levels * 8 * c_.grid_level_dim * tb for i, c_ in enumerate(props)) / len(props)
                   if props else 0.0)
        rg = gather_roofline("nlr_encode8_kernel", "encode", g_bytes, 1)
        rp = gather_roofline("nlr_prop8_kernel", "prop", p_bytes, len(props)) if props else None
        out = {
            "metric": "LiDAR rays/sec @128 samples/ray, 8x256 MLP; depth L1 vs reference",
            "value": rays_total / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": {_lib.PREC_F32: "f32 (f32 MFMA everywhere)",
                      _lib.PREC_MIXED: "f32 (sampling, hash grid, density/semantic/intensity layers on f32 MFMA) + bf16 MFMA (view MLP)",
                      _lib.PREC_FAST: "f32 (sampling, hash grid) + split-bf16 x3 MFMA (density/semantic/intensity) + bf16 MFMA (view MLP)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: nuScenes 32-beam sweep, {H_BEAMS}x{width} rays per sweep, samples "
                                   f"({','.join(str(x) for x in mc.level_samples())}), "
                                   f"{mc.nerf_mlp.net_depth_viewdirs}x{mc.nerf_mlp.net_width_viewdirs} NerfMLP + semantic"
                                   f"{' + intensity' if mc.config.use_intensity else ''} heads, "
                                   f"{'full-size' if args.log2_hashmap is None else f'2^{args.log2_hashmap}-entry'} "
                                   f"{'fp32' if args.table_dtype == 'f32' else 'fp16'} hash tables",
                       "rays_per_gpu_per_step": n_rays, "azimuth_columns_total": width,
                       "parallelism": f"azimuth-sector x{world}" + (" + 1 all_gather of the packed range image on a side stream" if use_dist else ""),
                       "per_sample_history": bool(args.history), "hip_graph_replay": bool(caps),
                       "weights": ("trained checkpoint" if args.ckpt else "seeded synthetic (white-noise tables, x1500 density gain)"),
                       "sweep_origins": ("static: the same sweep every step" if args.static_origin else
                                         f"moving: step i renders sweep i mod {n_sweeps} (pre-uploaded ray batches)"),
                       "flops_per_ray": nflops.flops_per_ray(mc), "gather_bytes_per_ray": nflops.gather_bytes_per_ray(mc)},
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
            "roofline": {"kernel": "nlr_mlp_kernel", "bound": "mfma", "achieved": achieved,
                         "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": traffic, "traffic_note": tnote,
                         "sustained_peak": MFMA_BF16_SUSTAINED_TFLOPS, "frac_of_sustained": achieved / MFMA_BF16_SUSTAINED_TFLOPS,
                         "sustained_note": "what this chip holds on LDS-fed bf16 MFMAs with random operands at its power limit (1 332 W, 2.12 GHz; "
                                           "2 056 TFLOP/s at 2.39 GHz on all-zero operands): profiles/r04_power_trace.txt"},
            "roofline_gather": rg,
            "roofline_prop": rp,
            "debug_switches": {"force_generic_level_body": L.nlr_debug_get(_lib.DBG_FORCE_GENERIC), "mlp_workgroups": L.nlr_debug_get(_lib.DBG_MLP_WORKGROUPS),
                               "ray_groups": L.nlr_debug_get(_lib.DBG_RAY_GROUPS)},
            "kernel_source_sha": bsha[:16],
            "binary_stale": buildinfo.stale(),
        }
        if ckpt_note:
            out["config"]["checkpoint"] = ckpt_note
        if caps:  # a captured sweep carries no per-kernel HIP events: nothing to report, rather than zeros
            out["kernel_ms"] = None
            out["roofline"].update(achieved=None, frac=None, traffic=None, traffic_note="HIP-graph replay: per-kernel durations are not measured in this mode")
            out["roofline_gather"] = out["roofline_prop"] = None
        if emul:  # not a benchmark result: what ONE rank of an `emul`-way split does per step, measured on one GPU
            out["emulated_world"] = emul
            out["metric"] = f"DIAGNOSTIC per-rank step of a {emul}-way azimuth split (one GPU, no collective)"
            out["projected_rays_per_s_at_world"] = n_rays * emul * args.steps / dt
            out["roofline"]["traffic"], out["roofline"]["traffic_note"] = None, "full-sweep profile does not apply to a sector"
        if ag_ms is not None:
            out["allgather_ms"] = round(ag_ms, 4)
            out["allgather_bytes_per_rank"] = int(gat.tiles[0].numel() * 4)
            out["ranks_seen"] = ranks_seen  # all-reduce (SUM) of 1 over the process group: must equal n_gpus
            out["rank_ms_per_step"] = {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4), "all": [round(x, 4) for x in rank_ms]}
            out["collective_backend"] = torch.distributed.get_backend()
        if world == 1 and not args.no_cpu_baseline and "r" in last:
            threads = host_cores()
            idx = np.linspace(0, n_rays - 1, args.cpu_rays).astype(np.int64)
            out["cpu_baseline"], ref = cpu_baseline(mc, sd, secs[last["sweep"]], idx, threads)  # the sweep the last step rendered
            out["accuracy"] = accuracy(ref, last["r"], idx)
            if plain and not args.no_trained_leg:
                del model, last["r"]
                torch.cuda.empty_cache()
                out["trained_scene"] = trained_scene_leg(dev, args.precision, threads)
        print(json.dumps(out))
    if use_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
