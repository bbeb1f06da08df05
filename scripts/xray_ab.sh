#!/bin/bash
# Experiment (diagnostic build nerf-lidar_amd/build/var/lib_xray.so, made from a patched COPY of csrc/nlr_encode.hip): a wave of the fused
# cast + encode kernels = 8 adjacent rays x one sample index (NLR_ENC_XRAY=1) against 8 consecutive samples of one ray (0), same box.
L=$PWD/nerf-lidar_amd/build/var/lib_xray.so
mkdir -p gpurun_out
for i in 1 2; do
 for X in 0 1; do
  for W in "white" "trained"; do
    if [ $W = trained ]; then A="--ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21"; else A=""; fi
    NLR_ENC_XRAY=$X NLR_LIB_PATH=$L timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-trained-leg $A 2>>gpurun_out/xray.err | tail -1 | \
      X=$X W=$W python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('xray', os.environ['X'], os.environ['W'], round(d['ms_per_step'], 3), {a: round(b, 4) for a, b in k.items()}, 'depth_l1', d.get('accuracy', {}).get('depth_l1'))"
  done
 done
done
