#!/bin/bash
# The last gpurun of a round: the GPU suite, smoke() and the default bench line of ONE tree -- the committed HEAD
# (scripts/final_gate.sh checks that the tree is clean and passes the sha) -- with the sha256 of the libraries that ran.
#   usage (on the GPU box, from the repo root):  scripts/gpu_final_gate.sh <head sha> [round tag]
HEAD_SHA=${1:-unknown}; TAG=${2:-r06}
OUT=gpurun_out/${TAG}_gate; mkdir -p $OUT
R=$OUT/gate.txt
{
  echo "gate of $TAG: HEAD $HEAD_SHA"
  echo "date (GPU box): $(date -u +%Y-%m-%dT%H:%M:%SZ)"
  sha256sum pipe_amd/lib/libpipe_hip.so pipe_amd/lib/libpipe_host.so oracle/liboracle_pipe.so 2>/dev/null
  sha256sum bench.py | sed 's/$/  (first 16 hex = bench_py_sha16)/'
} > $R
t0=$(date +%s)
timeout 1200 python -m pytest tests -q -m gpu --durations=15 > $OUT/pytest.txt 2>&1; rc_t=$?
t1=$(date +%s)
echo "pytest -m gpu: rc $rc_t in $((t1 - t0)) s: $(tail -1 $OUT/pytest.txt)" >> $R
grep -E "^(FAILED|ERROR)" $OUT/pytest.txt | head -20 >> $R
# (which tests are nearest the 120 s per-test limit on THIS box: the 15 slowest)
sed -n '/slowest 15 durations/,/^=/p' $OUT/pytest.txt | head -17 >> $R
# every soak script once, on the library that is being gated (VERDICT r5 "weak" 8: the soak was of an older library)
rc_k=0
for sc in stress_fused:12 stress_biquad_seg:6 stress_fir_mfma:6 stress_long_fir:4 stress_resampler:12 stress_resampler_rows:12 stress_percall:30 stress_hostcall:15; do
  name=${sc%%:*}; arg=${sc##*:}
  ts=$(date +%s)
  timeout 300 python scripts/$name.py $arg > $OUT/$name.txt 2>&1; rc=$?
  echo "soak $name $arg: rc $rc in $(( $(date +%s) - ts )) s: $(tail -1 $OUT/$name.txt | cut -c1-160)" >> $R
  [ $rc -ne 0 ] && rc_k=1
done
# the size-dependent dispatch rules against every form a knob can force (round 6: five rules were right only at the
# shape their round had measured); exit 1 = a default more than 25 % behind an alternative
ts=$(date +%s)
timeout 400 python scripts/dispatch_audit.py > $OUT/dispatch_audit.txt 2>&1; rc_a=$?
echo "dispatch audit: rc $rc_a in $(( $(date +%s) - ts )) s: $(tail -1 $OUT/dispatch_audit.txt | cut -c1-200)" >> $R
[ $rc_a -ne 0 ] && rc_k=1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; rc_s=$?
echo "smoke(): rc $rc_s: $(grep -E '^smoke' $OUT/smoke.txt | tail -1)" >> $R
timeout 600 python bench.py --gpus 1 > $OUT/bench.txt 2> $OUT/bench.err; rc_b=$?
echo "bench.py --gpus 1: rc $rc_b" >> $R
python - "$OUT/bench.txt" >> $R <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f'  value {d["value"]} {d["unit"]}, ms_per_step {d["ms_per_step"]}, roofline.frac {r["frac"]}, kernel {r["kernel"]} '
          f'avg {r["avg_kernel_ms"]} ms, traffic {r["traffic"]}')
    for k in ("headline_f64_buffers", "c4_chain", "c5_resampler_mix", "parity_stats"):
        if k in d:
            print(f"  {k}: " + json.dumps(d[k])[:600])
    print(f'  cpu_baseline: {json.dumps(d.get("cpu_baseline"))[:300]}')
except Exception as e:  # noqa: BLE001
    print("  no bench line:", e)
PY
echo "verdict: $([ $rc_t -eq 0 ] && [ $rc_s -eq 0 ] && [ $rc_b -eq 0 ] && [ $rc_k -eq 0 ] && echo GREEN || echo RED)" >> $R
cat $R
