#!/bin/bash
# diagnostic build (scripts/diag_encode.sh, -DNLR_DBG_ENV): XCD chunk size of the encode / proposal launches under the device-decided wave order
L=$PWD/nerf-lidar_amd/build/var/lib_encdbg.so
for CH in 32 64 128 256 16; do
  for W in white trained; do
    if [ $W = trained ]; then A="--ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21"; else A=""; fi
    NLR_ENC_CHUNK=$CH NLR_LIB_PATH=$L timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-trained-leg $A 2>/dev/null | tail -1 | \
      CH=$CH W=$W python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('chunk', os.environ['CH'].rjust(3), os.environ['W'].ljust(7), round(d['ms_per_step'], 3), {a: round(b, 4) for a, b in k.items() if a in ('prop','encode','mlp')})"
  done
done
