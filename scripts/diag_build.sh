#!/bin/bash
# Diagnostic only: build lib_d<mask>.so variants of the FAST 8_4_2 MLP instance with s_memtime stamps (+ ablations).
# usage: [EXTRA="-DNLR_POLL_F=20" TAG=p20] scripts/diag_build.sh 0 1 8 ...   (needs a finished `make` in nerf-lidar_amd/; output in nerf-lidar_amd/build/var)
set -e
cd "$(dirname "$0")/../nerf-lidar_amd"
python ../scripts/make_diag_kernel.py build/var
FL="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Icsrc -DNLR_INST_WT=8 -DNLR_INST_HT=4 -DNLR_INST_PREC=2"
OBJS=$(ls build/*.o | grep -v inst_8_4_2)
for d in "$@"; do
  ( hipcc $FL $EXTRA -DNLR_DIAG=$d -c build/var/diag_inst.hip -o build/var/inst_d$d$TAG.o && hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/lib_d$d$TAG.so $OBJS build/var/inst_d$d$TAG.o ) &
done
wait
ls -la build/var/*.so
