#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  env NLR_LIB_PATH=nerf-lidar_amd/build/var/lib_encdbg.so "$@" timeout -k 10 200 python3 bench.py $ARGS --steps 20 --warmup 5 --no-cpu-baseline --no-trained-leg 2>>gpurun_out/r4_encexp2.err | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$*', '| step', round(d['ms_per_step'],3), 'encode', k['encode'], 'prop', k['prop'], 'mlp', k['mlp'])"
}
echo "== trained (C2 checkpoint, inflated maps)"
ARGS="--ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21"
run A=0
run NLR_ENC_XCD=1
run NLR_ENC_XCD=3
run NLR_ENC_XCD=1 NLR_ENC_NT=0x200
run NLR_ENC_XCD=1 NLR_ENC_NT=0x300
run NLR_ENC_XCD=1 NLR_ENC_NT=0x3c0
run NLR_ENC_XCD=1 NLR_ENC_NTST=1
run NLR_ENC_XCD=3 NLR_ENC_NT=0x300 NLR_ENC_NTST=1
echo "== white-noise default"
ARGS=""
run A=0
run NLR_ENC_XCD=1
run NLR_ENC_XCD=3
run NLR_ENC_XCD=1 NLR_ENC_NTST=1
echo "== REF architecture, trained (REFI checkpoint, inflated maps)"
ARGS="--ckpt tests/golden/ckpt_trained --inflate-log2 21"
run A=0
run NLR_ENC_XCD=3
