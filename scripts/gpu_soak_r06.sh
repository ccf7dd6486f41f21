#!/bin/bash
# Round 6's long soak, on the library that is in the tree (its sha256 first): every scripts/stress_*.py at a hundred
# times the gate's counts, the async host loop with every stage in the SHARED doorbell queue (the hang of this round:
# profiles/r06_shared_queue_async_hang.txt) and with every stage asking for the exclusive doorbell, 40 child runs each
# with the hardware-queue pool at its default and at 1 / 2 / 8.
#   scripts/gpu_soak_r06.sh [out dir]
OUT=${1:-gpurun_out/r06_soak}; mkdir -p $OUT
R=$OUT/soak.txt
{ echo "soak of $(sha256sum pipe_amd/lib/libpipe_hip.so)"; date -u +%Y-%m-%dT%H:%M:%SZ; } > $R
export TMPDIR=/tmp PIPE_HIP_STALL_DUMP_MS=8000
for sc in stress_fused:1000 stress_biquad_seg:300 stress_fir_mfma:150 stress_long_fir:60 stress_resampler:400 stress_resampler_rows:300 stress_percall:600 stress_hostcall:90; do
  name=${sc%%:*}; arg=${sc##*:}
  ts=$(date +%s)
  timeout 900 python scripts/$name.py $arg > $OUT/$name.txt 2>&1; rc=$?
  echo "soak $name $arg: rc $rc in $(( $(date +%s) - ts )) s: $(tail -1 $OUT/$name.txt | cut -c1-220)" >> $R
done
for mode in SHARED EXCLUSIVE; do
for q in default 1 2 8; do
PYTHONPATH=$PWD python - "$mode" "$q" >> $R 2>&1 <<'PY'
import sys, time
mode, q = sys.argv[1], sys.argv[2]
from tests._child import run_child
env = {"PIPE_HOST_RESIDENT_SHARED" if mode == "SHARED" else "PIPE_HOST_RESIDENT": "1"}
if q != "default":
    env["GPU_MAX_HW_QUEUES"] = q
ok, worst, bad = 0, 0.0, []
for i in range(10 if q != "default" else 40):
    t0 = time.perf_counter()
    try:
        run_child("""
            import tests.test_host_pipe as T
            T.test_hip_copy_in_the_loop_config1(1)
            for _ in range(6):
                T.test_hip_fir_biquad_gain_lines_equal_oracle_loop(1)
            T.test_hip_fused_chain_equals_separate_stages_and_oracle()
            T.test_mutation_reaches_hip_handle_through_the_message()
            T.test_hip_processor_error_surfaces_as_run_error()
        """, timeout_s=60, env=env)
        ok += 1
    except AssertionError as e:
        bad.append((i, str(e)[-1500:]))
    worst = max(worst, time.perf_counter() - t0)
print(f"async host loop, every stage {mode}, hardware queues {q}: {ok} ok, {len(bad)} failed, slowest run {worst:.1f} s")
for i, msg in bad[:3]:
    print(f"  run {i}: {msg}")
PY
done; done
cat $R
