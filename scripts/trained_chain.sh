#!/bin/bash
# train the BENCHMARK architecture (C2, full-size fp32 tables) on the analytic scene on this box, write the
# checkpoint in the reference's format, read it back, and bench / profile the fused render on the trained field next to the white-noise
# default.  The 300 MB checkpoint stays on the box (/tmp); what comes back is gpurun_out/$TAG/*.
# usage (GPU box): TAG=r04_trained STEPS=3000 scripts/trained_chain.sh
set -u
TAG=${TAG:-r04_trained}
STEPS=${STEPS:-3000}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
CK=/tmp/ckpt_c2
mkdir -p $OUT
cd $R/nerf-lidar_amd
python3 -m nerflidar_hip.train_scene --workload C2 --steps $STEPS --rays 16384 --depth-lam 1.0 --log-every 500 --out $CK > $OUT/train.log 2> $OUT/train.err || exit 1
cp $CK/train_summary.json $OUT/train_summary.json
cd $R
python3 bench.py --ckpt $CK > $OUT/bench_trained.json 2> $OUT/bench_trained.err || exit 1
python3 bench.py --ckpt $CK --static-origin --no-cpu-baseline > $OUT/bench_trained_static_origin.json 2>> $OUT/bench_trained.err
python3 bench.py > $OUT/bench_noise.json 2> $OUT/bench_noise.err
python3 bench.py --static-origin --no-cpu-baseline > $OUT/bench_noise_static_origin.json 2>> $OUT/bench_noise.err
# L2 hit rates and bytes past L2 on the trained field (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  D=$OUT/pmc_$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/bench.py --ckpt $CK --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        if "nlr_" in k:
            res[k][c] = sum(v) / len(v)
for k, v in res.items():
    if "TCC_HIT_sum" in v: v["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    if "FETCH_SIZE" in v: v["fetch_bytes_x2"] = v["FETCH_SIZE"] * 2048
json.dump(res, open("$OUT/pmc_trained.json", "w"), indent=1)
PY
rm -rf $OUT/pmc_TCC_HIT_sum $OUT/pmc_FETCH_SIZE
ls -la $OUT
