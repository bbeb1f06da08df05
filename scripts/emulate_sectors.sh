#!/bin/bash
# What ONE rank of a P-way azimuth split does per step, on one GPU (no collective): eager with the per-kernel HIP events (what bench.py
# does), eager without them, and replayed from a HIP graph (bench.py --graph).  For profiles/rNN_emulated_sector_steps.txt.
one() {
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'ms/step', {k: round(v,4) for k,v in (d['kernel_ms'] or {}).items()})"
}
for P in 1 2 4 8; do
  E=""; [ $P -gt 1 ] && E="--emulate-world $P"
  echo "P=$P eager + events : $(one $E)"
  echo "P=$P eager, no events: $(NLR_BENCH_NOPROF=1 one $E)"
  echo "P=$P graph replay    : $(one $E --graph)"
done
