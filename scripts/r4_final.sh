#!/bin/bash
# final evidence of round 4, one box, final binary
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r4_gputest_final.txt 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r4_gputest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
TAG=r04_final scripts/profile_all.sh > gpurun_out/r04_final_log.txt 2>&1
timeout -k 10 300 python3 scripts/train_step_bench.py REF 65536 > gpurun_out/r04_final/train_step_bench.txt 2>&1
timeout -k 10 300 python3 scripts/train_step_bench.py C2 16384 >> gpurun_out/r04_final/train_step_bench.txt 2>&1
timeout -k 10 300 python3 scripts/raydrop_bench.py > gpurun_out/r04_final/raydrop_bench.txt 2>&1
timeout -k 10 600 scripts/emulate_sectors.sh > gpurun_out/r04_final/emulated_sector_steps.txt 2>&1
timeout -k 10 300 python3 scripts/grid_bwd_ab.py > gpurun_out/r04_final/grid_scatter_ab.txt 2>&1
ls gpurun_out/r04_final | wc -l; tail -4 gpurun_out/r04_final/train_step_bench.txt; tail -3 gpurun_out/r04_final/raydrop_bench.txt; cat gpurun_out/r04_final/emulated_sector_steps.txt
