#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_bwd_trace -- python3 $GRAFT_REPO_ROOT/scripts/grid_bwd_ab.py > $GRAFT_REPO_ROOT/gpurun_out/r4_grid_bwd_ab_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/r4_bwd_trace -name "*kernel_stats.csv" | head -1) gpurun_out/r4_bwd_kernel_stats.csv; rm -rf gpurun_out/r4_bwd_trace
head -12 gpurun_out/r4_bwd_kernel_stats.csv | cut -c1-180
