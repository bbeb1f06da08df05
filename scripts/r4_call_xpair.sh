#!/bin/bash
# scatter work of round 4 (x-pair atomics, tagged LDS cache for the coarse levels): parity tests of the grid backward and of training, the A/B,
# and the training step on the trained scene.
set -o pipefail
mkdir -p gpurun_out/r4_xpair
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_training.py -q -m gpu -k "grid_backward or training or encode_features" -x > gpurun_out/r4_xpair/tests.txt 2>&1 || { tail -30 gpurun_out/r4_xpair/tests.txt; exit 1; }
tail -2 gpurun_out/r4_xpair/tests.txt
timeout -k 10 300 python scripts/grid_bwd_ab.py > gpurun_out/r4_xpair/ab.txt 2>&1 || { tail -20 gpurun_out/r4_xpair/ab.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r4_xpair/ab.txt
timeout -k 10 280 python scripts/train_scene_profile.py tests/golden/ckpt_trained_c2 16384 21 > gpurun_out/r4_xpair/c2.txt 2>&1 || { tail -20 gpurun_out/r4_xpair/c2.txt; exit 1; }
grep "trained checkpoint\|per level" gpurun_out/r4_xpair/c2.txt
timeout -k 10 280 python scripts/train_scene_profile.py tests/golden/ckpt_trained 65536 21 > gpurun_out/r4_xpair/refi.txt 2>&1 || { tail -20 gpurun_out/r4_xpair/refi.txt; exit 1; }
grep "trained checkpoint\|per level" gpurun_out/r4_xpair/refi.txt
