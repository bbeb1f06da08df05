#!/bin/bash
# HBM traffic of the render kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section).
# Writes gpurun_out/pmc_traffic/summary.json = {"kernel_source_sha": <the source hash compiled into the measured binary, nlr_build_sha()>,
# "kernels": {name: {...}}}; copy it to profiles/r04_pmc_traffic.json.  bench.py quotes it only while the hash matches.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trained-leg > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trained-leg > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trained-leg > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, "nerf-lidar_amd")
from nerflidar_hip import buildinfo
res = collections.defaultdict(dict)
for d in ("fetch", "write", "l2"):
    for f in glob.glob(f"gpurun_out/pmc_traffic/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            if "nlr_" in k:
                res[k][c] = sum(v) / len(v)
for k, v in res.items():
    # FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies half of a wide coalesced stream (MI355X_MICROARCH.md, HBM section):
    # doubled for the streaming kernels.  For RANDOM 64-byte lines it tallies the line bytes exactly - calibrated on this access pattern with
    # scripts/micro/gather_rand.hip under --pmc FETCH_SIZE (profiles/r04_fetch_size_calibration.txt: 0.98-1.01 of lines x 64 B) - so the
    # gather kernels (encode / proposal: their traffic is scattered 16- and 4-byte reads) are NOT doubled.
    gather = "encode8" in k or "prop8" in k
    if "FETCH_SIZE" in v: v["hbm_read_bytes_corrected"] = v["FETCH_SIZE"] * 1024 * (1 if gather else 2)
    if "FETCH_SIZE" in v: v["fetch_correction"] = "x1 (random lines, calibrated)" if gather else "x2 (wide coalesced stream)"
    if "WRITE_SIZE" in v: v["hbm_write_bytes"] = v["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in v: v["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    print(k, json.dumps(v))
json.dump({"kernel_source_sha": buildinfo.binary_sha(), "stale": buildinfo.stale(), "kernels": res}, open("gpurun_out/pmc_traffic/summary.json", "w"), indent=1)
PY
