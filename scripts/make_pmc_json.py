#!/usr/bin/env python3
"""One entry of profiles/pmc_latest.json from a scripts/profile.sh output directory: the HBM bytes
per launch of the kernel bench.py timed, from the FETCH_SIZE / WRITE_SIZE passes.

  FETCH_SIZE, WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
  coalesced stream, so it is doubled (MI355X_MICROARCH.md, "HBM").  The entry carries the hash of
  the kernel sources it was measured on: bench.py quotes it only for that build."""
import glob
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1]
bench = json.loads(open(os.path.join(out, "trace.json")).read().strip().splitlines()[-1])
kname = bench["roofline"]["kernel"]
# bench kernel label -> device kernel name
if kname.startswith("chain_fused_kernel"):
    pat = r"fir_ols32_kernel<float, float, [12]"
elif "32x32" in kname:
    pat = r"fir_ols32_kernel<float, float, 0"
elif kname.startswith("fir_ols_kernel"):
    pat = r"fir_ols_kernel<"
else:
    pat = re.escape(kname.split("<")[0])


def mean_counter(sub, counter):
    vals = []
    for p in glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True):
        db = sqlite3.connect(p)
        per = {}
        for name, cname, val, disp in db.execute(
                "select name, counter_name, counter_value, dispatch_id from pmc_events"):
            if cname == counter and re.search(pat, name):
                per[disp] = per.get(disp, 0.0) + val   # one row per shader engine / XCC: sum them
        vals += list(per.values())
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fetch_kb, nf = mean_counter("pmc_fetch", "FETCH_SIZE")
write_kb, nw = mean_counter("pmc_write", "WRITE_SIZE")
import bench as B  # noqa: E402  (csrc_sha16)
entry = {
    "bench_kernel": kname,
    "device_kernel_regex": pat,
    "workload": bench["config"]["workload"],
    "csrc_sha16": B.csrc_sha16(),
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "dispatches": [nf, nw],
    "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM): 64 B counted per 128 B request",
    "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024) if fetch_kb and write_kb else None,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "source": f"scripts/profile.sh {os.path.basename(out.rstrip('/'))} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
}
if entry["hbm_bytes_per_launch"]:
    entry["traffic_over_algorithmic"] = round(entry["hbm_bytes_per_launch"] / entry["algorithmic_bytes_per_launch"], 4)
print(json.dumps(entry, indent=1))
