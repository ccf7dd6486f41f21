#!/bin/bash
# A/B bench of environment-selected kernel variants, interleaved, 3 rounds:
#   scripts/gpu_ab.sh "ENV1" "ENV2" ...
set -u
export TMPDIR=/tmp
for rep in 1 2 3; do
for e in "$@"; do
  env $e python bench.py --no-cpu-baseline --no-live-pmc --steps ${AB_STEPS:-150} --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$e]', r['kernel'], 'ms', r['avg_kernel_ms'], 'frac', r['frac'])"
done; done
