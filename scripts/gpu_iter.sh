#!/bin/bash
# quick GPU iteration: parity tests, then the bit-exact direct form at a few register blockings
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
for R in ${RS:-16 8 4}; do
  PIPE_HIP_FIR_EXACT=1 PIPE_HIP_FIR_R=$R python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('R=$R value=%.0f Ms/s ms/step=%.4f kernel_ms=%.4f hbm_frac=%.4f valu_frac=%.4f' % (r['value'], r['ms_per_step'], rf['avg_kernel_ms'], rf['frac'], rf['valu_f64']['frac']))"
done
