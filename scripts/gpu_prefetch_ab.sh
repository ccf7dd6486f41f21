#!/bin/bash
# A/B of the window requested ahead of the stores (ols32_kernel.hpp, kPrefetch; ablation 9 = off): the fused chain at
# one and sixteen buffers per Line, the headline launch, interleaved, 3 rounds.  Libraries: scripts/build_ablate_lib.sh
#   chain_fused PH_FUSE_ABLATE fab 0 9   and   fir_ols32 PH_FUSE_ABLATE ols 0 9
set -u
export TMPDIR=/tmp
for rep in 1 2 3; do
for k in 1 16; do for n in 0 9; do
  PROBE_BUFFERS=$k PIPE_HIP_LIB=$PWD/pipe_amd/lib/libpipe_hip_fab$n.so python scripts/chain_probe.py 400 2>/dev/null | tail -1
done; done
for n in 0 9; do
  echo -n "headline ols$n: "; PIPE_HIP_LIB=$PWD/pipe_amd/lib/libpipe_hip_ols$n.so python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-live-pmc --no-power 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['roofline']['kernel'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'])"
done; done
