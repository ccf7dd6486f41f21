#!/bin/bash
# Where the resampler's wave-cycles go (rocprofv3 PMC, SQ block, one pass of 8 counters): parked on a wait, stalled
# at issue, issuing -- and the dynamic instruction mix per launch.  PIPE_HIP_RESAMPLE_NO_WAVE=1 with the A/B library
# profiles the workgroup-tiled pair kernel instead.
OUT=${1:-gpurun_out/r05/resampler_pmc}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $REPO/$OUT/a -o a -f csv -- env PYTHONPATH=$REPO python $REPO/scripts/bench_resampler.py > $REPO/$OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $REPO/$OUT/b -o b -f csv -- env PYTHONPATH=$REPO python $REPO/scripts/bench_resampler.py > $REPO/$OUT/b.log 2>&1
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in "ab":
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            if "resample" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print(f"   {c:24s} {v / max(n[(k, c)], 1):16.1f} per launch")
PY
