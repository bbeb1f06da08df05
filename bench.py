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
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"
KNAMES = ["resample", "prop", "encode", "direnc", "mlp", "composite"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C2")
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
    ap.add_argument("--selftest-cpu", action="store_true", help="launcher + partition + collective logic on CPU (gloo) with a "
                    "stand-in renderer; for tests/, measures nothing")
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


def cpu_baseline(mc, sd, batch_np, idx, threads):
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
    chunk = 1024
    outs = []
    with torch.no_grad():
        orc.model_forward(sd, mc, {k: v[:256] for k, v in b.items()}, encoders=enc, sd_t=sdt)  # warm-up
        t0 = time.perf_counter()
        for i in range(0, n_rays, chunk):
            outs.append(orc.model_forward(sd, mc, {k: v[i:i + chunk] for k, v in b.items()}, encoders=enc, sd_t=sdt)[0][-1])
        dt = time.perf_counter() - t0
    ref = {k: torch.cat([o[k] for o in outs]).numpy() for k in ("depth", "intensity", "semantic") if k in outs[0]}
    return dict(value=n_rays / dt, unit="rays/s", cores=threads, kind="port",
                sample=f"{n_rays} rays of the same sweep (every {len(batch_np['origins']) // n_rays}th ray), "
                       f"same weights, chunks of {chunk}, {dt:.1f} s of wall time, fp32 PyTorch-CPU + OpenMP C grid oracle"), ref


def accuracy(ref, r, idx):
    """The timed GPU outputs against the oracle on the sampled rays (north_star: depth / intensity within 1e-3, labels exact)."""
    import numpy as np
    g = {k: r[k].detach().cpu().numpy()[idx] for k in ("depth", "intensity", "semantic", "labels") if k in r}
    d = np.abs(g["depth"] - ref["depth"])
    out = dict(rays=int(len(idx)), depth_l1=float(d.mean()), depth_p95=float(np.percentile(d, 95)), depth_max=float(d.max()))
    if "intensity" in ref and "intensity" in g:
        out["intensity_max"] = float(np.abs(g["intensity"] - ref["intensity"]).max())
    if "semantic" in ref and "labels" in g:
        s = np.sort(ref["semantic"], axis=-1)
        out["label_mismatches"] = int((g["labels"] != ref["semantic"].argmax(-1)).sum())
        out["min_top2_margin"] = float((s[:, -1] - s[:, -2]).min())
    return out


def selftest_cpu(args):
    """Partition + collective + reassembly with a stand-in renderer on CPU ranks (gloo).  No product code is timed."""
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd"))
    from nerflidar_hip import lidar as nlidar, sharding
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W = 4, 64 * (world if args.scaling == "weak" else 1)
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
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "steps": args.steps, "scaling": args.scaling, "image_equal": ok,
                          "shape": list(img.shape)}))
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

    mc = nconfig.workload(args.workload, args.log2_hashmap)
    sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
    model = Model(mc, sd, device=dev, precision=args.precision,
                  table_dtype=torch.float16 if args.table_dtype == "f16" else torch.float32)
    width = W_COLS * (world if args.scaling == "weak" else 1)
    full = nlidar.synthetic_sweep(width=width, seed=0)
    emul = args.emulate_world if (args.emulate_world > 1 and world == 1) else 0
    sec, wp = nlidar.azimuth_sector(full, H_BEAMS, width, rank, emul or world)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sec.items()}
    n_rays = H_BEAMS * wp
    sf = 1.0 / 250.0
    if emul:
        width = wp  # the "image" of the emulation is the rank's own tile
    gat = sharding.SweepGatherer(H_BEAMS, width, dev, force=args.force_dist)
    last = {}

    def step(i):
        tile = gat.tile(i)
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
    if not os.environ.get("NLR_BENCH_NOPROF"):  # (diagnostic switch: what do the HIP events themselves cost?)
        L.nlr_profile_begin(model._handle)  # HIP events on the launch stream around every kernel of the timed steps
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    img = gat.image(args.warmup + args.steps - 1)
    barrier()
    dt = time.perf_counter() - t0
    ms = (C.c_float * _lib.NLR_K_COUNT)()
    cnt = (C.c_uint32 * _lib.NLR_K_COUNT)()
    _lib.check(L.nlr_profile_end(model._handle, _lib.current_stream(), ms, cnt), "nlr_profile_end")
    ag_ms = None
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
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
        # HBM traffic of the dominant kernel per launch comes from separate rocprofv3 --pmc passes (scripts/pmc_traffic.sh), which
        # record the hash of the kernel sources they measured: a profile of other code is not a measurement of this binary.
        traffic, tnote = None, "no PMC profile of this kernel source committed"
        src = buildinfo.kernel_source_sha()
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")))
            if tj.get("kernel_source_sha") != src:
                tnote = f"profiles/r02_pmc_traffic.json was measured on kernel source {str(tj.get('kernel_source_sha'))[:12]}, this binary is {src[:12]}: traffic withheld"
            elif world == 1 and args.workload == "C2" and not args.chunk:
                for k, v in tj["kernels"].items():
                    if "nlr_mlp_kernel" in k:
                        traffic = v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
                        tnote = "bytes per launch, profiles/r02_pmc_traffic.json (separate rocprofv3 --pmc passes: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)"
        except Exception:
            pass
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
                       "per_sample_history": bool(args.history),
                       "flops_per_ray": nflops.flops_per_ray(mc), "gather_bytes_per_ray": nflops.gather_bytes_per_ray(mc)},
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
            "roofline": {"kernel": "nlr_mlp_kernel", "bound": "mfma", "achieved": achieved,
                         "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": traffic, "traffic_note": tnote},
            "kernel_source_sha": src[:16],
        }
        if emul:  # not a benchmark result: what ONE rank of an `emul`-way split does per step, measured on one GPU
            out["emulated_world"] = emul
            out["metric"] = f"DIAGNOSTIC per-rank step of a {emul}-way azimuth split (one GPU, no collective)"
            out["projected_rays_per_s_at_world"] = n_rays * emul * args.steps / dt
            out["roofline"]["traffic"], out["roofline"]["traffic_note"] = None, "full-sweep profile does not apply to a sector"
        if ag_ms is not None:
            out["allgather_ms"] = round(ag_ms, 4)
            out["allgather_bytes_per_rank"] = int(gat.tiles[0].numel() * 4)
        if world == 1 and not args.no_cpu_baseline and "r" in last:
            threads = host_cores()
            idx = np.linspace(0, n_rays - 1, args.cpu_rays).astype(np.int64)
            out["cpu_baseline"], ref = cpu_baseline(mc, sd, sec, idx, threads)
            out["accuracy"] = accuracy(ref, last["r"], idx)
        print(json.dumps(out))
    if use_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
