python -m pytest tests/test_trained_scene.py "tests/test_hip_parity.py::test_captured_sweep_owns_its_workspace" "tests/test_hip_parity.py::test_static_sweep_replays_from_a_hip_graph" -m gpu -x -q -s > gpurun_out/r4_trained_tests.txt 2>&1
echo "tests rc=$?" 
TAG=r04_trained STEPS=3000 scripts/trained_chain.sh > gpurun_out/r4_chain.txt 2>&1
echo "chain rc=$?"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_smalltrain_trace -- python3 $GRAFT_REPO_ROOT/scripts/train_small_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r4_smalltrain.txt 2>&1
cd $GRAFT_REPO_ROOT && cp $(find gpurun_out/r4_smalltrain_trace -name "*kernel_stats.csv" | head -1) gpurun_out/r4_smalltrain_kernel_stats.csv; rm -rf gpurun_out/r4_smalltrain_trace
tail -3 gpurun_out/r4_trained_tests.txt; tail -2 gpurun_out/r4_chain.txt
