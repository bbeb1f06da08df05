#!/bin/bash
# Diagnostic only: build nerf-lidar_amd/build/var/lib_<TAG>.so = the current library with the FAST 8x256 sem+intensity MLP instance
# recompiled with extra defines (e.g. -DNLR_STAMPS for phase stamps, -DNLR_PF=4, -DNLR_POLL_F=20, ablation switches).
# usage: TAG=stamps EXTRA="-DNLR_STAMPS" [COMP=0|1] scripts/diag_build.sh      (needs a finished `make` in nerf-lidar_amd/)
set -e
cd "$(dirname "$0")/../nerf-lidar_amd"
mkdir -p build/var
FL="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Icsrc"
hipcc $FL $EXTRA -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize -DNLR_INST_WT=8 -DNLR_INST_HT=4 -DNLR_INST_PREC=2 -DNLR_INST_COMP=${COMP:-1} -c csrc/nlr_mlp_inst.hip -o build/var/inst_$TAG.o
OBJS=$(ls build/*.o | grep -v inst_8_4_2_${COMP:-1})
hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/lib_$TAG.so $OBJS build/var/inst_$TAG.o
ls -la build/var/lib_$TAG.so
