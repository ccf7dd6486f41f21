#!/bin/bash
# Same-box A/B of two builds of libpipe_hip.so at steady state (scripts/gpu_steady.sh):
#   cp pipe_amd/lib/libpipe_hip.so pipe_amd/lib/libpipe_hip_old.so   (before the change)
#   cp pipe_amd/lib/libpipe_hip.so pipe_amd/lib/libpipe_hip_new.so   (after it)
#   gpurun -- 'bash scripts/gpu_ab.sh'
cp pipe_amd/lib/libpipe_hip.so /tmp/orig.so
for v in old new old new; do cp pipe_amd/lib/libpipe_hip_$v.so pipe_amd/lib/libpipe_hip.so; TAG=$v bash scripts/gpu_steady.sh | tail -1; done
cp /tmp/orig.so pipe_amd/lib/libpipe_hip.so
