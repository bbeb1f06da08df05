#!/bin/bash
# final evidence of round 4, final binary; two calls (the whole does not fit one 1200 s call):
#   scripts/r4_final.sh a    GPU tests, smoke, scripts/profile_all.sh (kernel trace, PMC passes, bench lines, workloads)
#   scripts/r4_final.sh b    training / ray-drop / sector / scatter benches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_final
if [ "$1" = "a" ]; then
  python -m pytest tests -m gpu -q > gpurun_out/r4_gputest_final.txt 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r4_gputest_final.txt
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  TAG=r04_final scripts/profile_all.sh > gpurun_out/r04_final_log.txt 2>&1
  ls gpurun_out/r04_final | wc -l; tail -2 gpurun_out/r04_final/bench.json | cut -c1-400
else
  timeout -k 10 200 python3 scripts/train_step_bench.py REF 65536 > gpurun_out/r04_final/train_step_bench.txt 2>&1
  timeout -k 10 200 python3 scripts/train_step_bench.py C2 16384 >> gpurun_out/r04_final/train_step_bench.txt 2>&1
  NLR_SCATTER_LEVELS=0 timeout -k 10 200 python3 scripts/train_scene_profile.py tests/golden/ckpt_trained_c2 16384 21 > gpurun_out/r04_final/train_scene_profile_c2.txt 2>&1
  NLR_SCATTER_LEVELS=0 timeout -k 10 200 python3 scripts/train_scene_profile.py tests/golden/ckpt_trained 65536 21 > gpurun_out/r04_final/train_scene_profile_refi.txt 2>&1
  timeout -k 10 200 python3 scripts/raydrop_bench.py > gpurun_out/r04_final/raydrop_bench.txt 2>&1
  timeout -k 10 400 scripts/emulate_sectors.sh > gpurun_out/r04_final/emulated_sector_steps.txt 2>&1
  timeout -k 10 200 python3 scripts/grid_bwd_ab.py > gpurun_out/r04_final/grid_scatter_ab.txt 2>&1
  grep -v amdgpu gpurun_out/r04_final/train_step_bench.txt | tail -6; grep "trained checkpoint" gpurun_out/r04_final/train_scene_profile_*.txt; tail -3 gpurun_out/r04_final/raydrop_bench.txt; cat gpurun_out/r04_final/emulated_sector_steps.txt
fi
