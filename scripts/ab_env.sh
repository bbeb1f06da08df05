#!/bin/bash
# Same-box A/B of one build under two environments.  usage: ROUNDS=3 scripts/ab_env.sh "NLR_STATIC_TILES=1" ""
R=${ROUNDS:-3}
for i in $(seq $R); do
  for E in "$@"; do
    env $E timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('[$E]', round(d['ms_per_step'],3), {a: round(b,4) for a,b in k.items()}, round(d['roofline']['frac'],4))"
  done
done
