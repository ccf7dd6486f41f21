#!/bin/bash
# Ablation builds of ONE source file: pipe_amd/lib/libpipe_hip_<tag><N>.so for every N given, the
# file compiled with -D<MACRO>=N, everything else from the normal build.  Use with PIPE_HIP_LIB=...
#   scripts/build_ablate_lib.sh resampler PH_RS_ABLATE rs 1 2 3 4
#   EXTRA_FLAGS='-mllvm -amdgpu-sched-strategy=max-memory-clause' scripts/build_ablate_lib.sh fir_ols32 PH_OLS_ABLATE ols 1 2 3
set -e
EXTRA_FLAGS=${EXTRA_FLAGS:-}
cd "$(dirname "$0")/../pipe_amd/csrc"
SRC=$1; MACRO=$2; TAG=$3; shift 3
mkdir -p build_prof
for N in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I. -D$MACRO=$N $EXTRA_FLAGS -c $SRC.hip -o build_prof/${SRC}_$N.o &
done
wait
for N in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/libpipe_hip_$TAG$N.so $(ls build/*.o | grep -v "build/$SRC.o" | grep -v host_) build_prof/${SRC}_$N.o
  echo built ../lib/libpipe_hip_$TAG$N.so
done
