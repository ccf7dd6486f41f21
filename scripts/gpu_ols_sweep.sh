#!/bin/bash
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('$1 kernel=%s kernel_ms=%.4f value=%.0f hbm_frac=%.4f' % (rf['kernel'], rf['avg_kernel_ms'], r['value'], rf['frac']))"; }
for W in ${WS:-8 12}; do PIPE_HIP_OLS_WAVES=$W run "waves=$W"; PIPE_HIP_OLS_WAVES=$W run "waves=$W"; done
timeout 600 python -m pytest tests/test_gpu_fir_ols.py -x -q 2>&1 | tail -3
