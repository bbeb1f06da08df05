#!/bin/bash
# encode-kernel experiments on the trained-scene workload (diagnostic build, switches from the environment)
cd $GRAFT_REPO_ROOT
run() {
  env NLR_LIB_PATH=nerf-lidar_amd/build/var/lib_encdbg.so "$@" timeout -k 10 200 python3 bench.py --ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21 --steps 20 --warmup 5 --no-cpu-baseline 2>>gpurun_out/r4_encexp.err | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$*', '| step', round(d['ms_per_step'],3), 'encode', k['encode'], 'prop', k['prop'], 'mlp', k['mlp'])"
}
run A=0
run NLR_ENC_XCD=1
run NLR_ENC_NOSCALAR=1
run NLR_ENC_LO=0 NLR_ENC_HI=3
run NLR_ENC_LO=3 NLR_ENC_HI=6
run NLR_ENC_LO=6 NLR_ENC_HI=10
run NLR_ENC_LO=0 NLR_ENC_HI=6
run NLR_ENC_LO=8 NLR_ENC_HI=10
run NLR_ENC_LO=9 NLR_ENC_HI=10
run NLR_ENC_LO=10 NLR_ENC_HI=10
