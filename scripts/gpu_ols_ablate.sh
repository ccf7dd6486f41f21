#!/bin/bash
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('$1 kernel_ms=%.4f' % (rf['avg_kernel_ms']))"; }
export PIPE_HIP_OLS_WAVES=${W:-8}
for A in 0 1 2 3 4 8 12 7 15; do PIPE_HIP_OLS_ABLATE=$A run "ablate=$A"; done
