#!/bin/bash
# A/B of the 16-byte window requests / stores of the 32 x 32 kernel on streams of 4 k channels (Args32::quad;
# libpipe_hip_quad0.so = chain_fused.hip built with -DPH_OLS_QUAD=0, scripts/build_ablate_lib.sh chain_fused PH_OLS_QUAD quad 0):
# the fused chain at one and sixteen buffers per Line, interleaved, 3 rounds.
set -u
export TMPDIR=/tmp
for rep in 1 2 3; do
for k in 1 16; do
  PROBE_BUFFERS=$k PIPE_HIP_LIB=$PWD/pipe_amd/lib/libpipe_hip_quad0.so python scripts/chain_probe.py 400 2>/dev/null | tail -1
  PROBE_BUFFERS=$k PIPE_HIP_LIB=$PWD/pipe_amd/lib/libpipe_hip.so python scripts/chain_probe.py 400 2>/dev/null | tail -1
done; done
