#!/bin/bash
# the bench lines of every --config plus smoke(), on the GPU box
set -u
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for c in 1 2 3; do
  python bench.py --config $c $( [ $c = 1 ] || echo --no-cpu-baseline ) > gpurun_out/r2/bench_c$c.json 2> gpurun_out/r2/bench_c$c.err
  tail -c 3000 gpurun_out/r2/bench_c$c.json; echo
done
