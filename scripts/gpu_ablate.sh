#!/bin/bash
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('$1 kernel_ms=%.4f value=%.0f' % (rf['avg_kernel_ms'], r['value']))"; }
for R in 16 8; do
export PIPE_HIP_FIR_R=$R
PIPE_HIP_FIR_ABLATE=0 run "R=$R full          "
PIPE_HIP_FIR_ABLATE=1 run "R=$R no-stage-loads"
PIPE_HIP_FIR_ABLATE=2 run "R=$R no-taps       "
PIPE_HIP_FIR_ABLATE=3 run "R=$R neither       "
done
