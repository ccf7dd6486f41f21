#!/bin/bash
# the test that never returned in round 4 -- the async host loop (a thread per component) with
# every HIP stage asking for PIPE_HIP_PARAM_RESIDENT -- again and again, each run in a child with a 60 s limit, with
# the runtime's hardware-queue pool at its default (4) and at 1, 2 and 8.  Also the 8-Line stress.
OUT=${1:-gpurun_out/r05}; mkdir -p $OUT
R=$OUT/resident_soak.txt; : > $R
run() {  # $1 = GPU_MAX_HW_QUEUES or "default", $2 = runs
  PYTHONPATH=$PWD GPU_MAX_HW_QUEUES_ARG=$1 python - "$1" "$2" >> $R 2>&1 <<'PY'
import os, sys, time
q, n = sys.argv[1], int(sys.argv[2])
from tests._child import run_child
env = {"PIPE_HOST_RESIDENT": "1"}
if q != "default":
    env["GPU_MAX_HW_QUEUES"] = q
ok, worst, bad = 0, 0.0, []
for i in range(n):
    t0 = time.perf_counter()
    try:
        run_child("""
            import tests.test_host_pipe as T
            T.test_hip_copy_in_the_loop_config1(1)
            T.test_hip_fir_biquad_gain_lines_equal_oracle_loop(1)
            T.test_hip_fused_chain_equals_separate_stages_and_oracle()
            T.test_mutation_reaches_hip_handle_through_the_message()
            T.test_hip_processor_error_surfaces_as_run_error()
        """, timeout_s=60, env=env)
        ok += 1
    except AssertionError as e:
        bad.append((i, str(e)[-400:]))
    worst = max(worst, time.perf_counter() - t0)
print(f"GPU_MAX_HW_QUEUES={q}: {ok} of {n} runs of the async host loop through the resident path passed; slowest {worst:.1f} s (limit 60)")
for i, e in bad[:3]:
    print(f"  run {i}: {e}")
PY
}
run default 50
run 1 10
run 2 10
run 8 10
( time timeout 400 python -m pytest "tests/test_host_pipe.py::test_many_lines_of_resident_stages_async_stress" -q -m gpu ) >> $R 2>&1
cat $R
