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


global_pat = [pat]


def mean_counter(sub, counter, largest_only=False, expect_kib=None):
    vals = []
    for p in glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True):
        db = sqlite3.connect(p)
        per = {}
        for name, cname, val, disp in db.execute(
                "select name, counter_name, counter_value, dispatch_id from pmc_events"):
            if cname == counter and re.search(global_pat[0], name):
                per[disp] = per.get(disp, 0.0) + val   # one row per shader engine / XCC: sum them
        vals += list(per.values())
    if largest_only and vals:  # the bench kernel's own launches: the same kernel also runs the smaller c3_shape
        top = max(vals)
        vals = [v for v in vals if v >= 0.5 * top]
    elif expect_kib and vals:  # the same kernel runs other objects of the line (the 64-Line resampler): this object's size only
        vals = [v for v in vals if 0.4 < v / expect_kib < 2.5]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def entry_for(kname, pat, workload, alg_bytes, largest_only=False):
    """HBM bytes per launch of the dispatches whose device kernel name matches `pat`."""
    global_pat[0] = pat
    # (bytes in ~ bytes out ~ alg / 2; FETCH_SIZE counts half of the bytes read: bench.py live_pmc's filter)
    fetch_kb, nf = mean_counter("pmc_fetch", "FETCH_SIZE", largest_only, alg_bytes / 4.0 / 1024.0)
    write_kb, nw = mean_counter("pmc_write", "WRITE_SIZE", largest_only, alg_bytes / 2.0 / 1024.0)
    e = {
        "bench_kernel": kname,
        "device_kernel_regex": pat,
        "workload": workload,
        "csrc_sha16": B.csrc_sha16(),
        "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "dispatches": [nf, nw],
        "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM; calibrated for 16 / 8 / 4 B per lane loads in "
                      "profiles/r03_fetch_calibration.txt: 0.500 - 0.508 of the bytes moved), WRITE_SIZE x1 (1.000 there)",
        "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024) if fetch_kb and write_kb else None,
        "algorithmic_bytes_per_launch": alg_bytes,
        "source": f"scripts/profile.sh {os.path.basename(out.rstrip('/'))} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
    }
    if e["hbm_bytes_per_launch"]:
        e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / alg_bytes, 4)
    return e


import bench as B  # noqa: E402  (csrc_sha16)
entries = [entry_for(kname, pat, bench["config"]["workload"], bench["roofline"]["algorithmic_bytes_per_launch"], largest_only=True)]
# the BASELINE configs[3] / configs[4] objects of the default line (when the PMC passes ran them too)
try:
    full = json.loads(open(os.path.join(out, "pmc_fetch.json")).read().strip().splitlines()[-1])
except (OSError, ValueError, IndexError):
    full = {}
c4 = full.get("c4_chain")
if c4:
    entries.append(entry_for(c4["kernel"], r"fir_ols32_kernel<float, float, [12]", c4["workload"], c4["algorithmic_bytes_per_launch"]))
c5 = (full.get("c5_resampler_mix") or {}).get("resampler")
if c5:
    entries.append(entry_for(c5["kernel"], r"resample_(wave|pair|tiled)_kernel<", full["c5_resampler_mix"]["workload"],
                             c5["algorithmic_bytes_per_launch"]))
print(json.dumps(entries if len(entries) > 1 else entries[0], indent=1))
