#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_bq -o bq -- python $REPO/scripts/bench_biquad.py > $REPO/gpurun_out/bq.log 2>&1
cd $REPO; cat gpurun_out/bq.log | tail -3
python scripts/summarize_prof.py gpurun_out/prof_bq 2>&1 | head -30
