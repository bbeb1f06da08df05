#!/bin/bash
# Same-box A/B of builds of libnerflidar_hip.so (MI355X devices differ by several % in sustained clock, so separate
# gpurun calls cannot resolve small changes).  usage: ROUNDS=3 scripts/ab_bench.sh libA.so libB.so [libC.so ...]
R=${ROUNDS:-3}
for i in $(seq $R); do
  for L in "$@"; do
    NLR_LIB_PATH=$L timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$L', round(d['ms_per_step'],3), {a: round(b,4) for a,b in k.items()})"
  done
done
