#!/bin/bash
# experiment (patched copy of nlr_encode.hip, diagnostic build): rays per group of the ray-fastest slot order (8 = shipped: a wave = 8 adjacent rays;
# 32: a WORKGROUP = 32 adjacent rays at one sample index), with two XCD chunk sizes
L=$PWD/nerf-lidar_amd/build/var/lib_grp.so
for R in 8 16 32 64; do
  for CH in 32 128; do
    A="--ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21"
    NLR_ENC_GROUPR=$R NLR_ENC_CHUNK=$CH NLR_LIB_PATH=$L timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-trained-leg $A 2>/dev/null | tail -1 | \
      R=$R CH=$CH python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('rays/group', os.environ['R'].rjust(3), 'chunk', os.environ['CH'].rjust(3), 'trained', round(d['ms_per_step'], 3), {a: round(b, 4) for a, b in k.items() if a in ('prop','encode','mlp')}, 'depth_max', d.get('accuracy', {}).get('depth_max'))"
  done
done
