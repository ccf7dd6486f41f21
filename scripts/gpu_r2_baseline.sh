#!/bin/bash
# round-2 baseline on the GPU box: full GPU test suite, bench with a clock log, tool probes
set -u
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r2/pytest_gpu.log
python scripts/clock_log.py --out gpurun_out/r2/clock_bench.json -- python bench.py --no-cpu-baseline > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err
python scripts/bench_configs.py > gpurun_out/r2/configs.jsonl 2> gpurun_out/r2/configs.err
( rocprofv3 --help 2>&1 | grep -i -A3 "att\|thread-trace\|advanced" | head -60 ) > gpurun_out/r2/rocprof_help.txt
ls /opt/rocm/lib | grep -i "att\|trace-decoder\|rocprof" >> gpurun_out/r2/rocprof_help.txt
ls /sys/class/drm/ >> gpurun_out/r2/rocprof_help.txt
tail -3 gpurun_out/r2/pytest_gpu.log; cat gpurun_out/r2/bench.json | head -c 1500
