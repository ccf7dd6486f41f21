#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
python $REPO/scripts/bench_percall.py 2>/dev/null
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_pc -o pc -- python $REPO/scripts/bench_percall.py > /dev/null 2>&1
cd $REPO
python - <<'PY'
import sqlite3,glob
p=glob.glob('gpurun_out/prof_pc/**/*.db',recursive=True)[0]
db=sqlite3.connect(p)
for r in db.execute("select name,total_calls,average from top_kernels limit 12"): print(r[0][:80], r[1], round(r[2]/1000,2),'us')
PY
