#!/bin/bash
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('$1 kernel=%s kernel_ms=%.4f ms_per_step=%.4f value=%.0f' % (rf['kernel'], rf['avg_kernel_ms'], r['ms_per_step'], r['value']))"; }
export PIPE_HIP_FIR_OLS_MIN_ITEMS=1
for B in 8 64 512 1024 2048 4096 8192; do run "buffers=$B" "--buffers $B"; done
