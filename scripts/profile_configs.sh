#!/bin/bash
# rocprofv3 kernel trace + HBM counters over scripts/bench_configs.py (every BASELINE config: FIR,
# chain, exact / segmented biquad, resampler, mix, gain) -> gpurun_out/prof/TAG/summary.txt
set -u
TAG=${1:-r02_configs}
OUT=$PWD/gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/scripts/bench_configs.py > $OUT/trace.jsonl 2> $OUT/trace.err
timeout -k 5 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python $REPO/scripts/bench_configs.py > $OUT/pmc_fetch.jsonl 2> $OUT/pmc_fetch.err
timeout -k 5 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python $REPO/scripts/bench_configs.py > $OUT/pmc_write.jsonl 2> $OUT/pmc_write.err
cd $REPO
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-260
