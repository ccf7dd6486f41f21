#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats, then separate PMC passes (never combined
# with other trace domains), of the same bench.py command.
#   scripts/profile.sh TAG [bench.py args...]        EXTRA_ENV="VAR=..." for environment switches
# Output under gpurun_out/prof/TAG/; summary printed and written to summary.txt; the HBM traffic
# of the bench kernel goes to gpurun_out/prof/TAG/pmc_entry.json (scripts/make_pmc_json.py).
set -u
TAG=${1:-r02}
shift || true
BENCH_ARGS="$*"
EXTRA_ENV=${EXTRA_ENV:-}
OUT=$PWD/gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
env $EXTRA_ENV timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py $BENCH_ARGS --no-cpu-baseline --no-secondary --no-live-pmc --no-power --no-per-call > $OUT/trace.json 2> $OUT/trace.err
# PMC_SECONDARY=1: the FETCH / WRITE passes also run the default line's configs[3] / configs[4] objects
# (c4_chain, c5_resampler_mix), so that their HBM traffic gets an entry too
pmc() { # name counters...
  local name=$1; shift
  local sec=--no-secondary
  if [ "${PMC_SECONDARY:-0}" = 1 ] && { [ $name = fetch ] || [ $name = write ]; }; then sec=; fi
  # (--no-per-call: the per_call row parks doorbell waits that a profiler serialising dispatches turns into watchdog
  # timeouts, 250 ms a call -- round 6 lost 45 GPU-minutes to it; every pass under its own limit)
  env $EXTRA_ENV timeout -k 5 400 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o pmc -- python $REPO/bench.py $BENCH_ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-power --no-scale-projection --no-per-call $sec > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
}
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $REPO
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
python scripts/make_pmc_json.py $OUT > $OUT/pmc_entry.json 2>> $OUT/summary.txt
cat $OUT/summary.txt
cat $OUT/pmc_entry.json
