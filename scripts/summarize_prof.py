#!/usr/bin/env python3
"""Condense rocprofv3 output (rocpd sqlite .db: kernel stats + PMC passes) into text."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]
KEEP = ("fir_", "ols", "gain_kernel", "biquad", "resample", "mix_kernel", "chain")

print(f"# rocprofv3 summary for {os.path.basename(out.rstrip('/'))}")
for p in sorted(glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True)):
    db = sqlite3.connect(p)
    print("## kernel stats (rocprofv3 --kernel-trace --stats); top_kernels reports MICROSECONDS")
    for name, calls, total, avg, pct in db.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        print(f"  {name[:110]:110s} calls={calls} total_us={total:.1f} avg_us={avg:.2f} pct={pct:.2f}")
    r = db.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
                   "grid_x, grid_y, grid_z, workgroup_x from kernels where name like ? limit 1", (os.environ.get('PROF_DISPATCH_LIKE', '%fir_ols%'),)).fetchone()
    if r:
        print(f"  dispatch: {r[0][:80]} vgpr={r[1]} agpr={r[2]} sgpr={r[3]} lds={r[4]} scratch={r[5]} "
              f"grid=({r[6]},{r[7]},{r[8]}) wg={r[9]}")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(p)
        agg = defaultdict(lambda: defaultdict(list))
        for name, cname, val in db.execute("select name, counter_name, counter_value from pmc_events"):
            agg[name][cname].append(val)
        print(f"## counters ({os.path.basename(d)}) -- mean per dispatch")
        for k, cs in agg.items():
            if not any(s in k for s in KEEP):
                continue
            print(f"  kernel {k[:110]}")
            for c, v in sorted(cs.items()):
                print(f"    {c:28s} mean={sum(v)/len(v):.6g}  n={len(v)}")
