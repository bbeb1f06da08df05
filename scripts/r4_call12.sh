#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  env NLR_LIB_PATH=nerf-lidar_amd/build/var/lib_encdbg.so "$@" timeout -k 10 200 python3 bench.py $ARGS --steps 20 --warmup 5 --no-cpu-baseline --no-trained-leg 2>>gpurun_out/r4_encexp5.err | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$*', '| step', round(d['ms_per_step'],3), 'encode', k['encode'], 'prop', k['prop'], 'mlp', k['mlp'])"
}
for W in "--ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21" ""; do
  ARGS="$W"; echo "== $W"
  run A=0
  run NLR_ENC_SWEEPW=1024 NLR_ENC_TILEB=1 NLR_ENC_TILEA=8
  run NLR_ENC_SWEEPW=1024 NLR_ENC_TILEB=2 NLR_ENC_TILEA=8 NLR_ENC_CHUNK=64
  run NLR_ENC_SWEEPW=1024 NLR_ENC_TILEB=4 NLR_ENC_TILEA=8 NLR_ENC_CHUNK=128
  run NLR_ENC_SWEEPW=1024 NLR_ENC_TILEB=4 NLR_ENC_TILEA=4 NLR_ENC_CHUNK=64
  run NLR_ENC_SWEEPW=1024 NLR_ENC_TILEB=8 NLR_ENC_TILEA=4 NLR_ENC_CHUNK=128
  run NLR_ENC_SWEEPW=1024 NLR_ENC_TILEB=4 NLR_ENC_TILEA=16 NLR_ENC_CHUNK=256
  run NLR_ENC_SWEEPW=1024 NLR_ENC_TILEB=32 NLR_ENC_TILEA=1 NLR_ENC_CHUNK=128
done
