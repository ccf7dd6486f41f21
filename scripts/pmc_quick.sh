#!/bin/bash
# quick PMC pass over an arbitrary command on the GPU box:  scripts/pmc_quick.sh OUTTAG "COUNTERS..." -- cmd...
set -u
TAG=$1; CNT=$2; shift 3
OUT=$PWD/gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout -k 5 ${PMC_TIMEOUT:-180} rocprofv3 --kernel-trace --pmc $CNT -d $OUT/pmc_q -o pmc -- "$@" > $OUT/cmd.out 2> $OUT/cmd.err  # (a counter set rocprofv3 cannot schedule aborts and then hangs in its finalizer)
cd $REPO
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(out, "pmc_q", "**", "*.db"), recursive=True):
    db = sqlite3.connect(p)
    per = collections.defaultdict(float)
    for name, cname, val, disp in db.execute("select name, counter_name, counter_value, dispatch_id from pmc_events"):
        per[(name, cname, disp)] += val
    for (name, cname, disp), v in per.items():
        acc[name][cname].append(v)
for name, cs in acc.items():
    if "pipehip" not in name: continue
    print(name[:110])
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} mean={sum(v)/len(v):14.1f} n={len(v)}")
PY
find $OUT -name '*.db' -delete
