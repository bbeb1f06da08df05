#!/bin/bash
# A/B on one box, production build: which samples share a wave of the fused cast + encode kernels - decided per level on the device (auto, the
# default), always 8 adjacent rays x one sample index (rays), always 8 consecutive samples of one ray (samples = rounds 1-3).
mkdir -p gpurun_out
run() {  # label, args...
  L=$1; shift
  timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-trained-leg "$@" 2>>gpurun_out/ray_groups.err | tail -1 | LBL="$L" python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print(os.environ['LBL'].ljust(44), round(d['value'] / 1e6, 3), 'M rays/s', round(d['ms_per_step'], 3), 'ms', {a: round(b, 4) for a, b in k.items() if a != 'direnc'})"
}
for i in 1 2; do
  for G in auto rays samples; do
    run "C2 white noise, $G" --ray-groups $G
    run "C2 trained, $G" --ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21 --ray-groups $G
  done
done
for G in auto rays samples; do
  run "REF white noise, $G" --workload REF --ray-groups $G
  run "REFI trained, $G" --ckpt tests/golden/ckpt_trained --inflate-log2 21 --ray-groups $G
  run "C1 (uniform samples), $G" --workload C1 --ray-groups $G
done
run "C2 --width 1100, auto" --width 1100
run "C2 f16 tables trained, auto" --ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21 --table-dtype f16
