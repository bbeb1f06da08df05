#!/bin/bash
# Diagnostic only: build nerf-lidar_amd/build/var/lib_<TAG>.so = the current library with csrc/nlr_encode.hip recompiled with extra defines
# (-DNLR_DBG_ENV: level range / XCD order / scalar path switchable from the environment, see nlr_encode.hip).
# usage: TAG=encdbg EXTRA="-DNLR_DBG_ENV" scripts/diag_encode.sh      (needs a finished `make` in nerf-lidar_amd/)
set -e
cd "$(dirname "$0")/../nerf-lidar_amd"
mkdir -p build/var
FL="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Icsrc"
hipcc $FL $EXTRA -c csrc/nlr_encode.hip -o build/var/encode_$TAG.o
OBJS=$(ls build/*.o | grep -v "build/nlr_encode.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/lib_$TAG.so $OBJS build/var/encode_$TAG.o
ls -la build/var/lib_$TAG.so
