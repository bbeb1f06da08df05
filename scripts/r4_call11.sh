#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_training.py -m gpu -x -q -s -k "dynamic_objects or whole_training" > gpurun_out/r4_tests11.txt 2>&1; echo "rc=$?"; tail -25 gpurun_out/r4_tests11.txt
