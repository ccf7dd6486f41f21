#!/bin/bash
# The fused chain of configs[3] with K buffers per Line and launch, under the power / clock sampler: where the kernel
# lands once a launch's edges are amortised, and at what power.   scripts/gpu_chain_k_power.sh [K...]
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for k in ${@:-1 4 16}; do
  python scripts/clock_log.py --period 0.01 --out gpurun_out/clock_c3_k$k.json -- \
    python bench.py --config 3 --buffers $k --no-cpu-baseline --no-live-pmc --steps $((6000 / k)) --warmup 300 2>/dev/null > gpurun_out/c3_k$k.out
  python - "$k" <<'PY'
import json, sys
k = sys.argv[1]
for l in open(f"gpurun_out/c3_k{k}.out"):
    if l.startswith("{") and '"roofline"' in l:
        d = json.loads(l); r = d["roofline"]
        line = {"buffers_per_line": int(k), "kernel": r["kernel"], "avg_kernel_ms": r["avg_kernel_ms"], "frac": r["frac"], "ms_per_step": d["ms_per_step"]}
c = json.load(open(f"gpurun_out/clock_c3_k{k}.json"))
line["power_w"] = c.get("power_w_busy") or c.get("power_w_all")
line["sclk_mhz"] = c.get("sclk_mhz_busy") or c.get("sclk_mhz_all")
print(json.dumps(line))
PY
done
