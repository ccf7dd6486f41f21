#!/bin/bash
# A/B of the fused chain's environment knobs at the configs[3] shape, interleaved, 3 rounds; then the
# event timeline of the prof build for each:   scripts/gpu_chain_ab.sh "ENV1" "ENV2" ...   ("-" = no knob)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
for e in "$@"; do
  [ "$e" = "-" ] && e="X_=1"
  echo -n "[$e] "; env $e python scripts/chain_probe.py 400 2>/dev/null | tail -1
done; done
if [ -f pipe_amd/lib/libpipe_hip_prof.so ]; then
for e in "$@"; do
  [ "$e" = "-" ] && e="X_=1"
  echo "=== timeline [$e]"; env $e PIPE_HIP_LIB=$PWD/pipe_amd/lib/libpipe_hip_prof.so python scripts/chain_probe.py 40 2>&1 | grep -E "fused prof|avg kernel"
done; fi
