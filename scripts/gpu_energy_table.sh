#!/bin/bash
# On the GPU box: shader clock, socket power and kernel time at steady state for the headline kernel, its
# ablation builds (scripts/build_ablate_lib.sh fir_ols32 PH_OLS_ABLATE ols 1 2 3), the direct form on the VALU and on the matrix pipe, the gain
# kernel, the fused chain and the resampler.  Output: gpurun_out/energy/*.json + table.txt
set -u
OUT=$PWD/gpurun_out/energy
mkdir -p $OUT
SEC=${SEC:-3}
run() { # tag kind [lib]
  local tag=$1 kind=$2 lib=${3:-}
  if [ -n "$lib" ]; then export PIPE_HIP_LIB=$lib; else unset PIPE_HIP_LIB; fi
  python scripts/clock_log.py --period 0.01 --out $OUT/clock_$tag.json -- python scripts/power_probe.py $kind $SEC > $OUT/run_$tag.log 2>&1
  unset PIPE_HIP_LIB
}
run idle idle
run ols ols
for n in 1 2 3; do [ -f pipe_amd/lib/libpipe_hip_ols$n.so ] && run ols_ablate$n ols $PWD/pipe_amd/lib/libpipe_hip_ols$n.so; done
[ -f pipe_amd/lib/libpipe_hip_ab.so ] && PIPE_HIP_FIR_NO_MFMA=1 run direct direct $PWD/pipe_amd/lib/libpipe_hip_ab.so   # the ordered-fma form on the VALU (an A/B switch: the AB build)
run mfma direct                            # ... and on the float64 matrix pipe (what large calls take)
run gain gain
run chain chain
run resampler resampler
run ols_again ols
python - "$OUT" <<'PY' | tee $OUT/table.txt
import glob, json, os, sys
out = sys.argv[1]
rows = []
idle_w = None
for tag in ["idle", "ols", "ols_ablate1", "ols_ablate2", "ols_ablate3", "direct", "mfma", "gain", "chain", "resampler", "ols_again"]:
    cp, rp = os.path.join(out, f"clock_{tag}.json"), os.path.join(out, f"run_{tag}.log")
    if not os.path.exists(cp):
        continue
    c = json.load(open(cp))
    probe = {}
    for line in open(rp):
        if line.startswith("{") and '"kind"' in line:
            probe = json.loads(line)
    pw, sc = c.get("power_w_busy") or c.get("power_w_all"), c.get("sclk_mhz_busy") or c.get("sclk_mhz_all")
    if tag == "idle":
        pw, sc = c.get("power_w_all"), c.get("sclk_mhz_all")
        idle_w = pw["median"]
    rows.append((tag, probe, pw, sc))
print("steady-state power / clock / kernel time (scripts/gpu_energy_table.sh; power = socket, hwmon; medians of 10 ms samples while busy)")
print(f"{'run':14s} {'kernel ms':>10s} {'Gsamples/s':>11s} {'TF/s f64':>9s} {'W med':>7s} {'W p90':>7s} {'sclk med':>9s} {'nJ/sample':>10s} {'nJ/sample above idle':>21s} {'pJ/flop above idle':>19s}")
for tag, p, pw, sc in rows:
    if not p or "avg_kernel_ms" not in p:
        print(f"{tag:14s} {'-':>10s} {'-':>11s} {'-':>9s} {pw['median']:7.0f} {pw['p90']:7.0f} {sc['median']:9.0f}")
        continue
    busy = p.get("busy_fraction", 1.0)
    sps = p["gsamples_per_s"] * 1e9
    nj = pw["median"] / sps * 1e9
    nj_dyn = (pw["median"] - (idle_w or 0)) / sps * 1e9
    fl = p["f64_tflops_as_issued"] * 1e12
    print(f"{tag:14s} {p['avg_kernel_ms']:10.4f} {p['gsamples_per_s']:11.1f} {p['f64_tflops_as_issued']:9.2f} {pw['median']:7.0f} {pw['p90']:7.0f} {sc['median']:9.0f} {nj:10.3f} {nj_dyn:21.3f} {(pw['median'] - (idle_w or 0)) / fl * 1e12:19.1f}   busy {busy}")
PY
