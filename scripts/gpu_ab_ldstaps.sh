#!/bin/bash
# A/B: taps through SGPR (s_load) vs through LDS (broadcast ds_read); needs hipcc on the box
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('$1 kernel_ms=%.4f value=%.0f valu=%.3f' % (rf['avg_kernel_ms'], r['value'], rf['valu_f64']['frac']))"; }
for V in 0 1; do
  touch pipe_amd/csrc/fir.hip
  make -C pipe_amd/csrc -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -DPIPE_HIP_FIR_LDS_TAPS=$V --offload-arch=gfx950 -I../../include -I." 2>&1 | grep -E " error" 
  run "ldstaps=$V"
  run "ldstaps=$V"
done
