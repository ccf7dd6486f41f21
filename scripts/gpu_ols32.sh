#!/bin/bash
# A/B of the two overlap-save decompositions on the GPU box
set -u
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
( PIPE_HIP_OLS_VARIANT=32 timeout 1200 python -m pytest tests/test_gpu_fir_ols.py -x -q 2>&1 | tail -15 ) > gpurun_out/r2/pytest_ols32.log
for v in 16 32 16 32; do
  PIPE_HIP_OLS_VARIANT=$v python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('variant $v', r['kernel'], 'ms', r['avg_kernel_ms'], 'frac', r['frac'], 'value', d['value'])"
done > gpurun_out/r2/ab_ols32.txt 2>&1
for v in 16 32; do
  PIPE_HIP_OLS_VARIANT=$v python scripts/bench_configs.py 2>/dev/null | grep '"config": [23]' | grep -i "fir\|chain" | cut -c1-330
done > gpurun_out/r2/ab_ols32_configs.txt 2>&1
tail -5 gpurun_out/r2/pytest_ols32.log; cat gpurun_out/r2/ab_ols32.txt gpurun_out/r2/ab_ols32_configs.txt
