#!/bin/bash
# socket power + shader clock sampled at 20 Hz by a SEPARATE process (scripts/power_sampler.py, sysfs only) while
#  (1) the default bench soaks (white-noise weights), (2) the same instruction stream runs on all-zero weights and tables, (3) the trained
#  checkpoint (inflated maps) renders, (4) scripts/micro/mfma_shape4 sustains its LDS-fed MFMA loop on random and (5) on zero operands.
# Writes gpurun_out/power_trace/{samples.csv, phases.txt, *.json|txt, summary.txt}.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/power_trace
mkdir -p $OUT
cd $R
python3 scripts/power_sampler.py $OUT/samples.csv 20 &
SP=$!
sleep 3
ph() { echo "$1 $(date +%s.%N)" >> $OUT/phases.txt; }
ph idle_end
ph soak_noise_start;   python3 bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-trained-leg > $OUT/soak_noise.json 2> $OUT/soak_noise.err;   ph soak_noise_end
sleep 3
ph soak_zero_start;    python3 bench.py --steps 1500 --warmup 20 --no-cpu-baseline --weight-scale 0 > $OUT/soak_zero.json 2> $OUT/soak_zero.err;    ph soak_zero_end
sleep 3
ph soak_trained_start; python3 bench.py --steps 800 --warmup 20 --no-cpu-baseline --ckpt tests/golden/ckpt_trained_c2 --inflate-log2 21 > $OUT/soak_trained.json 2> $OUT/soak_trained.err; ph soak_trained_end
sleep 3
ph micro_random_start; timeout -k 5 60 nerf-lidar_amd/build/mfma_shape4 10 > $OUT/micro_random.txt 2>&1;      ph micro_random_end
sleep 3
ph micro_zero_start;   timeout -k 5 60 nerf-lidar_amd/build/mfma_shape4 10 zero > $OUT/micro_zero.txt 2>&1;   ph micro_zero_end
sleep 2
kill $SP; wait $SP 2>/dev/null
python3 scripts/power_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
