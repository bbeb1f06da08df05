#!/bin/bash
# Counters of the gather kernels on the TRAINED-scene workload (committed C2 checkpoint inflated to full-size maps): separate --pmc passes.
# usage (GPU box): scripts/pmc_trained.sh > gpurun_out/pmc_trained.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_tr
mkdir -p $OUT
B="python3 $R/bench.py --ckpt $R/tests/golden/ckpt_trained_c2 --inflate-log2 21 --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
         "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
         "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- $B > /dev/null 2> $OUT/p$i.err
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_tr/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "encode8" in k or "prop8" in k:
                print(d.split("/")[-2], k, {c: f"{x / max(1, n[(k, c)]):.4g}" for c, x in v.items()})
PY
rm -rf $OUT
