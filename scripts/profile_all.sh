#!/bin/bash
# One-stop evidence run for profiles/ (all from the SAME build): bench line, rocprofv3 kernel trace + stats, SQ counters and HBM
# traffic of every render kernel (separate --pmc passes: gpurun refuses --pmc together with tracing domains), phase stamps.
# usage (on the GPU box): TAG=r02_final scripts/profile_all.sh     -> gpurun_out/$TAG/*; copy what matters into profiles/
set -u
TAG=${TAG:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
python3 -c "import sys; sys.path.insert(0, \"nerf-lidar_amd\"); from nerflidar_hip import buildinfo; print(buildinfo.binary_sha()); print(\"stale:\", buildinfo.stale())" > $OUT/kernel_source_sha.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cd $R
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && cp $STATS $OUT/kernel_stats.csv
scripts/pmc_mlp.sh 2 > $OUT/pmc_sq.txt 2>&1
scripts/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_traffic/summary.json $OUT/pmc_traffic.json 2>/dev/null
if [ -f nerf-lidar_amd/build/var/lib_stamps.so ]; then
  NLR_LIB_PATH=nerf-lidar_amd/build/var/lib_stamps.so timeout -k 10 200 python3 scripts/stamp_probe.py > $OUT/mlp_stamps.txt 2>&1
fi
rm -rf $OUT/trace gpurun_out/pmc gpurun_out/pmc_traffic
ls -la $OUT
