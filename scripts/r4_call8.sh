#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_trained_scene.py -m gpu -x -q > gpurun_out/r4_parity8.txt 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r4_parity8.txt
run() {
  env "$@" timeout -k 10 200 python3 bench.py $ARGS --steps 20 --warmup 5 --no-cpu-baseline --no-trained-leg 2>>gpurun_out/r4_encexp4.err | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$*', '| step', round(d['ms_per_step'],3), 'encode', k['encode'], 'prop', k['prop'], 'mlp', k['mlp'])"
}
D=NLR_LIB_PATH=nerf-lidar_amd/build/var/lib_encdbg.so
for W in "--ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21" ""; do
  ARGS="$W"; echo "== $W"
  run PRODUCTION=1
  run $D NLR_ENC_CHUNK=0
  run $D
  run $D NLR_ENC_PERSIST=8
  run $D NLR_ENC_PERSIST=6
  run $D NLR_ENC_PERSIST=4
  run $D NLR_ENC_PERSIST=16
done
