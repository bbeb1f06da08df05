#!/usr/bin/env python3
"""bench.py -- LiDAR rays/s of the MI355X-native render hot path (BASELINE.json metric).

A step = one pass of the whole hot path (all proposal levels + NerfMLP level + compositing, outputs resident in
HBM) over one synthetic nuScenes-shaped sweep sector of 32 beams x 1024 azimuth columns (32 768 rays) per GPU,
configuration C2 of BASELINE.json: (64, 64, 128) samples per ray, 8x256 view MLP, semantic + intensity heads,
full-size hash tables (229 MiB NerfMLP table, fp32).  With N > 1 the sweep has 1024*N azimuth columns, GPU p
renders sector p and ONE all-gather (RCCL) reassembles the packed [32, 1024*N, 7] range image on every rank
(weak scaling: per-GPU work fixed).

    python bench.py --gpus N --steps K --warmup W         (N > 1: under torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0; carries `roofline` for the dominant kernel (nlr_mlp_kernel, timed with HIP
events on its own stream inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nerf-lidar_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from nerflidar_hip import _lib, config as nconfig, flops as nflops, lidar as nlidar, sharding, weights as nweights
from nerflidar_hip.models import Model

H_BEAMS, W_COLS = 32, 1024
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"
KNAMES = ["resample", "prop", "encode", "direnc", "mlp", "composite"]


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


def cpu_baseline(mc, sd, batch_np, n_rays, threads):
    """The oracle (a port: PyTorch-CPU restatement pinned on reference fixtures) timed on this host's cores."""
    from oracle import nlr_oracle as orc
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    idx = np.linspace(0, batch_np["origins"].shape[0] - 1, n_rays).astype(np.int64)
    b = {k: torch.from_numpy(np.ascontiguousarray(v[idx])) for k, v in batch_np.items()}
    enc = orc.make_encoders(sd, mc)
    sdt = orc.to_torch_sd(sd)
    chunk = 1024
    def run():
        for i in range(0, n_rays, chunk):
            orc.model_forward(sd, mc, {k: v[i:i + chunk] for k, v in b.items()}, encoders=enc, sd_t=sdt)
    with torch.no_grad():
        orc.model_forward(sd, mc, {k: v[:256] for k, v in b.items()}, encoders=enc, sd_t=sdt)  # warm-up
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
    return dict(value=n_rays / dt, unit="rays/s", cores=threads, kind="port",
                sample=f"{n_rays} rays of the same sweep (every {len(batch_np['origins']) // n_rays}th ray), "
                       f"same weights, chunks of {chunk}, {dt:.1f} s of wall time, fp32 PyTorch-CPU + OpenMP C grid oracle")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--precision", type=int, default=_lib.PREC_FAST)
    ap.add_argument("--log2-hashmap", type=int, default=None, help="shrink the hash tables (debug only)")
    ap.add_argument("--cpu-rays", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--table-dtype", choices=["f32", "f16"], default="f32", help="hash-table storage; f32 is the benchmark "
                    "configuration, f16 is what the reference uses under autocast (Z/gridencoder/grid.py:43-44)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: initialise RCCL and run the collectives also with one rank")
    ap.add_argument("--chunk", type=int, default=0, help="rays per nlr_render_rays call (0 = the whole sector at once; the "
                    "reference's driver uses Config.render_chunk_size = 16384, ZI/configs.py)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if world == 1:  # rehearsal without a launcher
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    mc = nconfig.workload(args.workload, args.log2_hashmap)
    sd = nweights.synth_state_dict(mc, seed=0, trained_like=True)
    model = Model(mc, sd, device=dev, precision=args.precision,
                  table_dtype=torch.float16 if args.table_dtype == "f16" else torch.float32)
    width = W_COLS * world
    full = nlidar.synthetic_sweep(width=width, seed=0)
    sec, wp = nlidar.azimuth_sector(full, H_BEAMS, width, rank, world)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sec.items()}
    n_rays = H_BEAMS * wp
    sf = 1.0 / 250.0

    def step():
        if args.chunk and args.chunk < n_rays:
            parts = [model.render_rays({k: v[i:i + args.chunk] for k, v in batch.items()}, compute_extras=True, scale_factor=sf)[0]
                     for i in range(0, n_rays, args.chunk)]
            r = {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
        else:
            r, _ = model.render_rays(batch, compute_extras=True, scale_factor=sf)
        tile = sharding.pack_tile(r, H_BEAMS, wp)
        return sharding.gather_tiles(tile, width, force=args.force_dist)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    L = _lib.lib()
    if not os.environ.get("NLR_BENCH_NOPROF"):  # (diagnostic switch: what do the HIP events themselves cost?)
        L.nlr_profile_begin(model._handle)  # HIP events on the launch stream around every kernel of the timed steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = step()
    barrier()
    dt = time.perf_counter() - t0
    ms = (C.c_float * _lib.NLR_K_COUNT)()
    cnt = (C.c_uint32 * _lib.NLR_K_COUNT)()
    _lib.check(L.nlr_profile_end(model._handle, _lib.current_stream(), ms, cnt), "nlr_profile_end")
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    assert img.shape == (H_BEAMS, width, 7)

    if rank == 0:
        rays_total = n_rays * world * args.steps
        kern = {KNAMES[i]: (ms[i] / cnt[i] if cnt[i] else 0.0) for i in range(_lib.NLR_K_COUNT)}
        # dominant kernel: nlr_mlp_kernel.  Algorithmic FLOPs per launch = 2 * MACs/sample (SURVEY 8d:
        # 657 408 for the 8x256 NerfMLP + heads) * samples per launch (rays * 128).
        S_last = mc.level_samples()[-1]
        fl_launch = 2.0 * nflops.macs_per_sample(mc.nerf_mlp) * n_rays * S_last
        mlp_s = kern["mlp"] * 1e-3
        achieved = fl_launch / mlp_s / 1e12 if mlp_s > 0 else 0.0
        # HBM traffic of the dominant kernel per launch: not measurable from inside the process; taken from the last
        # rocprofv3 --pmc run of scripts/pmc_traffic.sh (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), if committed.
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_final3_pmc_traffic.json")))
            for k, v in tj.items():
                if "nlr_mlp_kernel" in k and world == 1 and args.workload == "C2":
                    traffic = v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {_lib.PREC_F32: "f32 (f32 MFMA everywhere)",
                      _lib.PREC_MIXED: "f32 (sampling, hash grid, density/semantic/intensity layers on f32 MFMA) + bf16 MFMA (view MLP)",
                      _lib.PREC_FAST: "f32 (sampling, hash grid) + split-bf16 x3 MFMA (density/semantic/intensity) + bf16 MFMA (view MLP)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: nuScenes 32-beam sweep, 32x1024 rays per GPU, samples "
                                   f"({','.join(str(x) for x in mc.level_samples())}), "
                                   f"{mc.nerf_mlp.net_depth_viewdirs}x{mc.nerf_mlp.net_width_viewdirs} NerfMLP + semantic"
                                   f"{' + intensity' if mc.config.use_intensity else ''} heads, "
                                   f"{'full-size' if args.log2_hashmap is None else f'2^{args.log2_hashmap}-entry'} "
                                   f"{'fp32' if args.table_dtype == 'f32' else 'fp16'} hash tables",
                       "rays_per_gpu_per_step": n_rays, "azimuth_columns_total": width,
                       "parallelism": f"azimuth-sector x{world}" + (" + 1 all_gather of the packed range image" if world > 1 else ""),
                       "flops_per_ray": nflops.flops_per_ray(mc), "gather_bytes_per_ray": nflops.gather_bytes_per_ray(mc)},
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
            "roofline": {"kernel": "nlr_mlp_kernel", "bound": "mfma", "achieved": achieved,
                         "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": "bytes per launch from profiles/r01_final3_pmc_traffic.json (separate rocprofv3 --pmc passes)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = host_cores()
            out["cpu_baseline"] = cpu_baseline(mc, sd, sec, args.cpu_rays, threads)
        print(json.dumps(out))
    if use_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
