#!/bin/bash
# Debug build with in-kernel phase timing of the resampler kernels (-DPH_RS_PROF) and the A/B switches
# (-DPIPE_HIP_AB): pipe_amd/lib/libpipe_hip_prof.so; use with PIPE_HIP_LIB=...
set -e
cd "$(dirname "$0")/../pipe_amd/csrc"
mkdir -p build_prof
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I. -DPH_RS_PROF=1 -DPIPE_HIP_AB=1 -c resampler.hip -o build_prof/resampler.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/libpipe_hip_prof.so $(ls build/*.o | grep -v "/resampler.o" | grep -v host_) build_prof/resampler.o
echo built ../lib/libpipe_hip_prof.so
