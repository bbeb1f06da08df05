#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_trained_scene.py -m gpu -x -q -s > gpurun_out/r4_trained_tests2.txt 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4_trained_tests2.txt
timeout -k 10 120 nerf-lidar_amd/build/gather_rand > gpurun_out/r4_gather_rand.txt 2>&1; cat gpurun_out/r4_gather_rand.txt
scripts/pmc_trained.sh > gpurun_out/r4_pmc_trained.txt 2>&1; cat gpurun_out/r4_pmc_trained.txt
python3 bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; python3 -c "
import json; d=json.load(open('gpurun_out/r4_bench_default.json')); print(int(d['value']), d['ms_per_step'], d['kernel_ms']); t=d['trained_scene']; print('trained', int(t['value']), t['ms_per_step'], t['kernel_ms'], t['accuracy'])"
