#!/bin/bash
# per-call latency through the C ABI: plain path completing by a word (default) / by an event
# (PIPE_HIP_COMPLETION_EVENT=1: round 4's plain path) / with the device's doorbell (PIPE_HIP_PARAM_RESIDENT)
OUT=${1:-gpurun_out/r05}; mkdir -p $OUT
gcc -std=c99 -O2 -Iinclude examples/percall_latency.c -Lpipe_amd/lib -lpipe_hip -lm -Wl,-rpath,$PWD/pipe_amd/lib -o /tmp/pl || exit 1
timeout 150 /tmp/pl 3000 > $OUT/percall_latency.jsonl 2>&1
PIPE_HIP_COMPLETION_EVENT=1 timeout 150 /tmp/pl 3000 > $OUT/percall_latency_event.jsonl 2>&1
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
rows = lambda f: [json.loads(l) for l in open(f) if l.startswith("{")]
w, e = rows(out + "/percall_latency.jsonl"), rows(out + "/percall_latency_event.jsonl")
print("stage io | plain by event | plain by word | doorbell | identical")
for a, b in zip(w, e):
    print(f'{a["stage"]} {a["io"]} | {b["plain_us"]["median"]:.1f} | {a["plain_us"]["median"]:.1f} | {a["resident_us"]["median"]:.1f} | {a["outputs_identical"]}')
PY
