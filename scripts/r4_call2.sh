#!/bin/bash
# call 2: is the training loop still host-bound?  then train the C2 architecture with small maps (the committable trained checkpoint),
# and bench it inflated to full-size maps.
cd $GRAFT_REPO_ROOT
python3 scripts/train_small_probe.py 13 8192 2>&1 | grep "ms per step" > gpurun_out/r4_probe.txt
cat gpurun_out/r4_probe.txt
MS=$(sed -n 's/.*: \([0-9.]*\) ms per step/\1/p' gpurun_out/r4_probe.txt | cut -d. -f1)
STEPS=4000; [ "${MS:-999}" -lt 45 ] && STEPS=10000
echo "training $STEPS steps"
cd nerf-lidar_amd
python3 -m nerflidar_hip.train_scene --workload C2 --log2-hashmap 13 --steps $STEPS --rays 8192 --depth-lam 4.0 --eval-every 2000 --log-every 1000 --out ../gpurun_out/trained_c2s > ../gpurun_out/train_c2s.log 2>&1 || { tail -5 ../gpurun_out/train_c2s.log; exit 1; }
cd ..
grep held_out gpurun_out/train_c2s.log | cut -c1-400
python3 bench.py --ckpt gpurun_out/trained_c2s --inflate-log2 21 > gpurun_out/r4_bench_c2s_inflated.json 2> gpurun_out/r4_bench_c2s_inflated.err
python3 bench.py --ckpt gpurun_out/trained_c2s --no-cpu-baseline > gpurun_out/r4_bench_c2s_small.json 2>> gpurun_out/r4_bench_c2s_inflated.err
python3 bench.py --ckpt tests/golden/ckpt_trained --inflate-log2 21 --no-cpu-baseline > gpurun_out/r4_bench_refi_inflated.json 2>> gpurun_out/r4_bench_c2s_inflated.err
python3 bench.py --ckpt gpurun_out/trained_c2s --inflate-log2 21 --table-dtype f16 --no-cpu-baseline > gpurun_out/r4_bench_c2s_inflated_f16.json 2>> gpurun_out/r4_bench_c2s_inflated.err
for f in c2s_inflated c2s_small refi_inflated c2s_inflated_f16; do python3 -c "
import json; d=json.load(open('gpurun_out/r4_bench_$f.json')); print('$f', int(d['value']), round(d['ms_per_step'],3), d['kernel_ms'], (d.get('accuracy') or {}).get('depth_max'))"; done
