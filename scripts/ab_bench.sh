#!/bin/bash
# Same-box A/B of builds of libnerflidar_hip.so (MI355X devices differ by several % in sustained clock, so separate
# gpurun calls cannot resolve small changes).  usage: ROUNDS=3 scripts/ab_bench.sh libA.so libB.so [libC.so ...]
# stderr of every run is kept in gpurun_out/ab_bench.err (round 3 dropped it and two variants that printed no JSON went unexplained).
R=${ROUNDS:-3}
ERR=${GRAFT_REPO_ROOT:-.}/gpurun_out/ab_bench.err
mkdir -p $(dirname $ERR)
for i in $(seq $R); do
  for L in "$@"; do
    NLR_LIB_PATH=$L timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline $ARGS 2>>$ERR | tail -1 | \
      python -c "
import sys, json
line = sys.stdin.read()
try:
    d = json.loads(line); k = d['kernel_ms']; print('$L', round(d['ms_per_step'], 3), {a: round(b, 4) for a, b in k.items()})
except Exception as e:
    print('$L', 'NO JSON LINE (see gpurun_out/ab_bench.err):', repr(line[:200]))
"
  done
done
