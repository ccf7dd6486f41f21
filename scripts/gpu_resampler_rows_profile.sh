#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of scripts/bench_resampler.py (its 8-channel shape takes the row form),
# then separate --pmc passes: HBM bytes (FETCH_SIZE, WRITE_SIZE) and the SQ block's view of the row kernel.
#   scripts/gpu_resampler_rows_profile.sh [outdir]
OUT=${1:-gpurun_out/r05/rows_profile}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/t -o t -- env PYTHONPATH=$REPO python $REPO/scripts/bench_resampler.py > $REPO/$OUT/t.log 2>&1
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/t/**/*kernel_stats.csv", recursive=True):
    print("== rocprofv3 --kernel-trace --stats (durations in ns)")
    for r in csv.DictReader(open(f)):
        if "resample" in r["Name"]:
            print(f'{r["Name"].split("(")[0][:90]:92s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]):10.1f} min {r["MinNs"]:>8s} max {r["MaxNs"]:>8s}')
PY
if [ -z "${ROWS_SQ_ONLY:-}" ]; then  # (FETCH_SIZE and WRITE_SIZE do not schedule together: one pass each)
bash scripts/pmc_quick.sh rows_fetch "FETCH_SIZE" -- env PYTHONPATH=$REPO python $REPO/scripts/bench_resampler.py | grep -A2 "resample_rows"
bash scripts/pmc_quick.sh rows_write "WRITE_SIZE" -- env PYTHONPATH=$REPO python $REPO/scripts/bench_resampler.py | grep -A2 "resample_rows"
fi
[ -n "${ROWS_HBM_ONLY:-}" ] && exit 0
bash scripts/pmc_quick.sh rows_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" -- env PYTHONPATH=$REPO python $REPO/scripts/bench_resampler.py | grep -A9 "resample_rows"
bash scripts/pmc_quick.sh rows_sq2 "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" -- env PYTHONPATH=$REPO python $REPO/scripts/bench_resampler.py | grep -A9 "resample_rows"
find $OUT -name '*.db' -delete; rm -rf $OUT/t
