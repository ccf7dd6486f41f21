#!/bin/bash
# Debug build of the library with in-kernel phase timing of the fused chain kernel
# (-DPH_FUSE_PROF): pipe_amd/lib/libpipe_hip_prof.so; use with PIPE_HIP_LIB=...
# TIMELINE=1: also -DPH_FUSE_TIMELINE (event timeline of two workgroups; distorts the phase sums)
set -e
cd "$(dirname "$0")/../pipe_amd/csrc"
mkdir -p build_prof
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I. -DPH_FUSE_PROF=1 ${TIMELINE:+-DPH_FUSE_TIMELINE=1} -c chain_fused.hip -o build_prof/chain_fused.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/libpipe_hip_prof.so $(ls build/*.o | grep -v chain_fused.o) build_prof/chain_fused.o
echo built ../lib/libpipe_hip_prof.so
