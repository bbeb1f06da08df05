#!/bin/bash
# sync-free loss terms: the training tests, then the step timings with / without per-step reads
set -o pipefail
mkdir -p gpurun_out/r4_sync
timeout -k 10 600 python -m pytest tests/test_training.py tests/test_trained_scene.py tests/test_losses.py -q -m gpu -x > gpurun_out/r4_sync/tests.txt 2>&1 || { tail -40 gpurun_out/r4_sync/tests.txt; exit 1; }
tail -2 gpurun_out/r4_sync/tests.txt
NLR_SCATTER_LEVELS=0 timeout -k 10 200 python3 scripts/train_scene_profile.py tests/golden/ckpt_trained_c2 16384 21 > gpurun_out/r4_sync/c2.txt 2>&1 || { tail -20 gpurun_out/r4_sync/c2.txt; exit 1; }
NLR_SCATTER_LEVELS=0 timeout -k 10 200 python3 scripts/train_scene_profile.py tests/golden/ckpt_trained 65536 21 > gpurun_out/r4_sync/refi.txt 2>&1 || { tail -20 gpurun_out/r4_sync/refi.txt; exit 1; }
grep -h "trained checkpoint" gpurun_out/r4_sync/c2.txt gpurun_out/r4_sync/refi.txt
timeout -k 10 200 python3 scripts/train_step_bench.py REF 65536 2>&1 | grep -v amdgpu | tee gpurun_out/r4_sync/train_step_bench.txt
timeout -k 10 200 python3 scripts/train_step_bench.py C2 16384 2>&1 | grep -v amdgpu | tee -a gpurun_out/r4_sync/train_step_bench.txt
