#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_host_pipe.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/batched_tests.log
timeout 600 python scripts/bench_host_batched.py > gpurun_out/batched_bench.log 2>&1
cat gpurun_out/batched_tests.log gpurun_out/batched_bench.log
