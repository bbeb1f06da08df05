#!/bin/bash
# One-stop evidence run for profiles/ (all from the SAME build): bench line, rocprofv3 kernel trace + stats, SQ counters and HBM
# traffic of every render kernel (separate --pmc passes: gpurun refuses --pmc together with tracing domains), the FETCH_SIZE calibration
# on random 64-byte lines, the other workloads, the trained-scene counters.
# usage (on the GPU box): TAG=r04_final scripts/profile_all.sh     -> gpurun_out/$TAG/*; copy what matters into profiles/
set -u
TAG=${TAG:-r04_final}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
python3 -c "import sys; sys.path.insert(0, \"nerf-lidar_amd\"); from nerflidar_hip import buildinfo; print(buildinfo.binary_sha()); print(\"stale:\", buildinfo.stale())" > $OUT/kernel_source_sha.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trained-leg > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_tr -- python3 $R/bench.py --ckpt $R/tests/golden/ckpt_trained_c2 --inflate-log2 21 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_trained_under_rocprof.json 2> $OUT/trace_tr.err
# FETCH_SIZE calibration on a known count of random 64-byte lines (scripts/micro/gather_rand.hip prints the line count per launch)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- $R/nerf-lidar_amd/build/gather_rand > $OUT/gather_rand_under_pmc.txt 2> $OUT/calib.err
cd $R
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$STATS" ] && cp $STATS $OUT/kernel_stats.csv
STATS=$(find $OUT/trace_tr -name "*kernel_stats.csv" | head -1); [ -n "$STATS" ] && cp $STATS $OUT/kernel_stats_trained.csv
python3 - <<PY > $OUT/fetch_size_calibration.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
lines = 256 * 64 * 256 * 32 * 4   # blocks x threads x iterations x pairs: distinct 64-byte lines per launch of gather_rand
print("FETCH_SIZE (KiB) per launch of scripts/micro/gather_rand.hip; every launch reads", lines, "distinct random 64-byte lines =", lines * 64 / 1e9, "GB")
for k, v in acc.items():
    for i in range(0, len(v), 3):   # 3 repetitions per (table, mode)
        m = sum(v[i:i + 3]) / len(v[i:i + 3])
        print(f"{k:50s} launch group {i // 3:2d}: FETCH_SIZE x 1024 = {m * 1024 / 1e9:7.2f} GB = {m * 1024 / (lines * 64):.3f} of the line bytes")
PY
scripts/pmc_mlp.sh 2 > $OUT/pmc_sq.txt 2>&1
scripts/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_traffic/summary.json $OUT/pmc_traffic.json 2>/dev/null
scripts/pmc_trained.sh > $OUT/pmc_trained.txt 2>&1
scripts/workloads.sh > $OUT/workloads.txt 2>&1
rm -rf $OUT/trace $OUT/trace_tr $OUT/calib gpurun_out/pmc gpurun_out/pmc_traffic
ls -la $OUT
