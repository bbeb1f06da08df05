#!/bin/bash
# Diagnostic: nlr_mlp_kernel time against the number of persistent workgroups (= busy CUs): is the launch power-limited, i.e. do fewer
# CUs clock higher?  usage: scripts/mlp_cu_sweep.sh 256 224 192 ...
for W in "$@"; do
  timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trained-leg --mlp-workgroups $W 2>/dev/null | tail -1 | \
    W=$W python -c "import sys,json,os; d=json.loads(sys.stdin.read()); w=int(os.environ['W']); m=d['kernel_ms']['mlp']; print(w, 'step', round(d['ms_per_step'],3), 'mlp', m, 'mlp x W/256 =', round(m*w/256,4))"
done
