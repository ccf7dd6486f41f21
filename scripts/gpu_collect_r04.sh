#!/bin/bash
# On the GPU box: what profiles/ holds for round 4, in one call (a trimmed scripts/gpu_collect_round.sh: the
# kernels' rocprofv3 summaries and counters, the bench lines, the host-path probes of this round).
set -u
TAG=r04
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
AB=$PWD/pipe_amd/lib/libpipe_hip_ab.so
bash scripts/gpu_profile_all.sh $TAG > $OUT/profile_all.log 2>&1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in 2 3; do python bench.py --config $c --no-cpu-baseline > $OUT/bench_c$c.json 2>/dev/null; done
python scripts/bench_hostcall.py > $OUT/hostcall.jsonl 2>/dev/null
python scripts/bench_hostcall_pinned.py > $OUT/hostcall_pinned.jsonl 2>/dev/null
gcc -std=c99 -O2 -Iinclude examples/percall_latency.c -Lpipe_amd/lib -lpipe_hip -lm -Wl,-rpath,$PWD/pipe_amd/lib -o /tmp/pl && timeout 120 /tmp/pl 3000 > $OUT/percall_latency.jsonl 2>&1
python scripts/percall_probe.py auto gain biquad biquad2 chain > $OUT/percall_probe.jsonl 2>/dev/null
timeout 60 scripts/micro/stream_wait_latency 2000 > $OUT/stream_wait_latency.txt 2>&1
timeout 60 scripts/micro/pcie_rates > $OUT/pcie_rates.txt 2>&1
python scripts/chain_probe.py > $OUT/chain_probe.txt 2>&1
PROBE_SECTIONS=2 python scripts/chain_probe.py >> $OUT/chain_probe.txt 2>&1
PROBE_LINES=64 PIPE_HIP_LIB=$AB python scripts/chain_probe.py >> $OUT/chain_probe.txt 2>&1
python scripts/resampler_lines_probe.py > $OUT/resampler_lines.txt 2>&1
SEC=3 bash scripts/gpu_energy_table.sh > $OUT/energy.log 2>&1
cp gpurun_out/energy/table.txt $OUT/energy_table.txt
grep -v amdgpu $OUT/chain_probe.txt; tail -3 $OUT/hostcall_pinned.jsonl; tail -12 $OUT/energy_table.txt
