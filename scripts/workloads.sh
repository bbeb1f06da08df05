#!/bin/bash
# The other workloads of SURVEY 8d on the CURRENT binary (one line each: rays/s, ms per step, kernel ms, MLP TFLOP/s), for profiles/rNN_workloads.txt:
# C2 (the benchmark), C2 with fp16 tables, the reference's 32 x 1100 sweep, the shipped architecture (REF), C1, C2 as one uniform level (C2S),
# the exact-f32 modes, the TRAINED checkpoints (C2 and the shipped architecture, hash maps inflated to full size; C2 also with its own
# small maps), and config C3 (4 cameras 1024 x 768) through render_image.
run() {
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trained-leg "$@" 2>>gpurun_out/workloads.err | tail -1 | \
    ARGS="$*" python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['ARGS'] or 'C2', '|', int(d['value']), 'rays/s', round(d['ms_per_step'],3), 'ms', d['kernel_ms'], 'mlp', round(d['roofline']['achieved'],1), 'TFLOP/s', 'encode', round(d['roofline_gather']['algorithmic_gather_GBps']/1e3,2), 'TB/s algorithmic')"
}
python -c "import sys; sys.path.insert(0,'nerf-lidar_amd'); from nerflidar_hip import buildinfo; print('# binary', buildinfo.binary_sha()[:16], 'stale:', buildinfo.stale())"
run
run --static-origin
run --table-dtype f16
run --width 1100
run --workload REF
run --workload C1
run --workload C2S
run --precision 1
run --ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21
run --ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21 --table-dtype f16
run --ckpt tests/golden/ckpt_trained_c2
run --ckpt tests/golden/ckpt_trained --inflate-log2 21
run --precision 0 --steps 5 --warmup 2
timeout -k 10 300 python scripts/camera_bench.py 32768 2>/dev/null | tail -1
