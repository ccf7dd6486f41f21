#!/bin/bash
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('$1 kernel_ms=%.4f value=%.0f frac=%.4f' % (rf['avg_kernel_ms'], r['value'], rf['frac']))"; }
for w in 8 12 16; do PIPE_HIP_OLS_WAVES=$w run "waves=$w"; done
PIPE_HIP_OLS_WAVES=${BEST:-16} timeout 900 python -m pytest tests/test_gpu_fir_ols.py -x -q 2>&1 | tail -3
