#!/bin/bash
# sweep register blocking x workgroups per CU of the bit-exact direct form on the bench workload
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('$1 kernel_ms=%.4f value=%.0f valu=%.3f' % (rf['avg_kernel_ms'], r['value'], rf['valu_f64']['frac']))"; }
DEFAULT_CFGS="16:2 16:0 8:2 8:3 8:4 8:0 4:4 4:0"
for cfg in ${CFGS:-$DEFAULT_CFGS}; do
  R=${cfg%%:*}; W=${cfg##*:}
  PIPE_HIP_FIR_EXACT=1 PIPE_HIP_FIR_R=$R PIPE_HIP_FIR_WGS_PER_CU=$W run "R=$R wgs/cu=$W"
done
