#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fir_ols.py -x -q > gpurun_out/pytest_ols.log 2>&1; echo "pytest ols rc=$?"; tail -25 gpurun_out/pytest_ols.log
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('$1 kernel=%s kernel_ms=%.4f value=%.0f hbm_frac=%.4f' % (rf['kernel'], rf['avg_kernel_ms'], r['value'], rf['frac']))"; }
run "default"
PIPE_HIP_FIR_EXACT=1 run "exact  "
