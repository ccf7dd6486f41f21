#!/bin/bash
# A/B of ablation builds of the fused chain (scripts/build_ablate_lib.sh chain_fused PH_FUSE_ABLATE fab 0 1 2 3 ...) at the
# configs[3] shape, one and sixteen buffers per Line, interleaved, 3 rounds:   scripts/gpu_chain_fold_ab.sh 0 1 2 3
set -u
export TMPDIR=/tmp
for rep in 1 2 3; do
for k in 1 16; do
for n in "$@"; do
  PROBE_BUFFERS=$k PIPE_HIP_LIB=$PWD/pipe_amd/lib/libpipe_hip_fab$n.so python scripts/chain_probe.py 400 2>/dev/null | tail -1
done; done; done
