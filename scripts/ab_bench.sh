#!/bin/bash
# Same-box A/B of two builds of libnerflidar_hip.so (MI355X devices differ by several % in sustained clock, so two
# gpurun calls cannot resolve small changes).  usage: scripts/ab_bench.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    NLR_LIB_PATH=$L timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$L', round(d['ms_per_step'],3), {a: round(b,4) for a,b in k.items()})"
  done
done
