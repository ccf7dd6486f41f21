#!/bin/bash
# headline kernel at steady state on a short launch (4096 buffers): for A/B comparisons of kernel variants
for i in 1 2; do python bench.py --buffers ${K:-4096} --steps 800 --warmup 300 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('${TAG:-head}', r['roofline']['kernel'], r['roofline']['avg_kernel_ms'], r['roofline']['frac'])"; done
