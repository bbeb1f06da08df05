#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -k "grid_backward or gridencoder" tests/test_raydrop.py tests/test_training.py -m gpu -x -q > gpurun_out/r4_tests9.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4_tests9.txt
timeout -k 10 300 python3 scripts/grid_bwd_ab.py > gpurun_out/r4_grid_bwd_ab.txt 2>&1; cat gpurun_out/r4_grid_bwd_ab.txt | tail -5
timeout -k 10 300 python3 scripts/raydrop_bench.py > gpurun_out/r4_raydrop_bench.txt 2>&1; tail -6 gpurun_out/r4_raydrop_bench.txt
