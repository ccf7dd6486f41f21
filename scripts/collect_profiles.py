#!/usr/bin/env python3
"""Copy what scripts/gpu_profile_all.sh TAG left under gpurun_out/prof/ into profiles/ (tracked):
   profiles/TAG_c{1,2,3}_summary.txt   kernel stats + counters of bench.py --config N
   profiles/TAG_c{1,2,3}_bench.json    the bench line of the traced run
   profiles/TAG_configs_summary.txt    the same for scripts/bench_configs.py (every BASELINE config)
   profiles/TAG_configs.jsonl          its result lines
   profiles/pmc_latest.json            {"kernels": [entry per config]} -- what bench.py quotes as
                                       roofline.traffic when kernel, size and source hash match
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
entries = []
for c in (1, 2, 3):
    d = os.path.join(src, f"{tag}_c{c}")
    if not os.path.isdir(d):
        continue
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(dst, f"{tag}_c{c}_summary.txt"))
    line = open(os.path.join(d, "trace.json")).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(dst, f"{tag}_c{c}_bench.json"), "w").write(line + "\n")
    e = json.load(open(os.path.join(d, "pmc_entry.json")))
    for one in (e if isinstance(e, list) else [e]):
        one["config"] = c if one is (e[0] if isinstance(e, list) else e) else ("3 (c4_chain of the default line)" if "chain" in one["bench_kernel"] else "4 (c5 of the default line)")
        entries.append(one)
d = os.path.join(src, f"{tag}_configs")
if os.path.isdir(d):
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(dst, f"{tag}_configs_summary.txt"))
    lines = [l for l in open(os.path.join(d, "trace.jsonl")).read().splitlines() if l.startswith("{")]
    open(os.path.join(dst, f"{tag}_configs.jsonl"), "w").write("\n".join(lines) + "\n")
if entries:
    json.dump({"round": int(tag[1:3]) if tag[1:3].isdigit() else None, "kernels": entries},
              open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
    for e in entries:
        print(e["config"], e["bench_kernel"], e.get("traffic_over_algorithmic"), e["csrc_sha16"])
