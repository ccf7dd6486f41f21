#!/bin/bash
# On the GPU box: everything profiles/ holds for a round, in one call.
#   scripts/gpu_collect_round.sh r03
# Instrumented / ablation libraries are used when present (never shipped); build them first with
#   scripts/build_prof_lib.sh
#   scripts/build_ablate_lib.sh fir_ols32 PH_OLS_ABLATE ols 1 2 3
#   scripts/build_ablate_lib.sh resampler PH_RS_PROF rsprof_ 1 && mv pipe_amd/lib/libpipe_hip_rsprof_1.so pipe_amd/lib/libpipe_hip_rsprof.so
#   scripts/build_ablate_lib.sh fir_ols32p PH_OLSD_PROF dprof_ 1 && mv pipe_amd/lib/libpipe_hip_dprof_1.so pipe_amd/lib/libpipe_hip_dprof.so
set -u
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash scripts/gpu_profile_all.sh $TAG > $OUT/profile_all.log 2>&1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in 2 3; do python bench.py --config $c --no-cpu-baseline > $OUT/bench_c$c.json 2>/dev/null; done
python scripts/bench_hostcall.py > $OUT/hostcall.jsonl 2>/dev/null
python scripts/bench_configs.py > $OUT/configs_untraced.jsonl 2>/dev/null
bash scripts/gpu_fetch_calib.sh > $OUT/calib.log 2>&1
cp gpurun_out/prof/fetch_calib.txt $OUT/fetch_calibration.txt
# in-kernel phase profiles (instrumented builds)
if [ -f pipe_amd/lib/libpipe_hip_prof.so ]; then
  PIPE_HIP_LIB=$PWD/pipe_amd/lib/libpipe_hip_prof.so python scripts/chain_probe.py 50 > $OUT/fused_chain_phase_profile.txt 2>&1
  PROBE_SECTIONS=2 PIPE_HIP_LIB=$PWD/pipe_amd/lib/libpipe_hip_prof.so python scripts/chain_probe.py 50 >> $OUT/fused_chain_phase_profile.txt 2>&1
fi
if [ -f pipe_amd/lib/libpipe_hip_rsprof.so ]; then
  PIPE_HIP_LIB=$PWD/pipe_amd/lib/libpipe_hip_rsprof.so python scripts/resampler_probe.py 40 > $OUT/resampler_phase_profile.txt 2>&1
fi
if [ -f pipe_amd/lib/libpipe_hip_dprof.so ]; then
  PROF_LIB=$PWD/pipe_amd/lib/libpipe_hip_dprof.so bash scripts/gpu_long_fir.sh > $OUT/long_fir_phase_profile.txt 2>&1
fi
python scripts/resampler_lines_probe.py > $OUT/resampler_lines.txt 2>&1
python scripts/resampler_probe.py 3000 > $OUT/resampler_probe.txt 2>&1
PIPE_HIP_RESAMPLE_NO_PAIR=1 python scripts/resampler_probe.py 3000 >> $OUT/resampler_probe.txt 2>&1
python scripts/chain_probe.py > $OUT/chain_probe.txt 2>&1
PROBE_SECTIONS=2 python scripts/chain_probe.py >> $OUT/chain_probe.txt 2>&1
PROBE_SECTIONS=2 PIPE_HIP_CHAIN_ONE_SECTION=1 python scripts/chain_probe.py >> $OUT/chain_probe.txt 2>&1
rm -f $OUT/long_fir.jsonl $OUT/long_fir_short_lines.jsonl
for n in 1024 2048 4096; do  # the same long FIRs over 512 Lines of one 4096-frame buffer
  python bench.py --config 2 --lines 512 --buffers 1 --taps $n --no-secondary --no-cpu-baseline --no-live-pmc --steps 50 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({'taps': $n, 'lines': 512, 'frames_per_line': 4096, 'kernel': r['kernel'], 'avg_kernel_ms': r['avg_kernel_ms'], 'gsamples_per_s': round(d['value']/1e3,1)}))" >> $OUT/long_fir_short_lines.jsonl
done
for n in 512 1024 2048 4096; do
  python bench.py --taps $n --no-secondary --no-cpu-baseline --no-live-pmc --steps 20 --warmup 3 --buffers 32768 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({'taps': $n, 'kernel': r['kernel'], 'avg_kernel_ms': r['avg_kernel_ms'], 'gsamples_per_s': round(d['value']/1e3,1), 'bit_exact_gsamples_per_s': round(d['bit_exact_form']['msamples_per_s']/1e3,1)}))" >> $OUT/long_fir.jsonl
done
rm -f $OUT/channels.jsonl
for c in 1 2 3; do  # the headline workload with one, two, three channels
  python bench.py --channels $c --buffers 32768 --no-secondary --no-cpu-baseline --no-live-pmc --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({'channels': $c, 'kernel': r['kernel'], 'avg_kernel_ms': r['avg_kernel_ms'], 'hbm_frac': r['frac'], 'gsamples_per_s': round(d['value']/1e3,1)}))" >> $OUT/channels.jsonl
done
python scripts/fir_exact_sweep.py > $OUT/fir_exact_sweep.txt 2>&1
# the segmented biquad over shapes: one-pass tile form, 1 and 2 sections; two-pass tiles and the lane walk beside it
( for sct in 1 2 3 4; do echo "== sections $sct"; PROBE_SECTIONS=$sct python scripts/biquad_shapes_probe.py 2>&1 | grep -v amdgpu; done
  echo "== two passes + scan kernel (PIPE_HIP_BIQUAD_TWO_PASS)"; PIPE_HIP_BIQUAD_TWO_PASS=1 python scripts/biquad_shapes_probe.py 2>&1 | grep -v amdgpu
  echo "== lane walk (PIPE_HIP_BIQUAD_NO_TILE)"; PIPE_HIP_BIQUAD_NO_TILE=1 python scripts/biquad_shapes_probe.py 2>&1 | grep -v amdgpu ) > $OUT/biquad_shapes.txt
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && PROBE_ONLY_C=2,8 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bqprof -o bq -- python scripts/biquad_shapes_probe.py ) < /dev/null > /dev/null 2>&1
cp $OUT/bqprof/bq_kernel_stats.csv $OUT/biquad_tile_kernel_stats.csv 2>/dev/null; rm -rf $OUT/bqprof
SEC=3 bash scripts/gpu_energy_table.sh > $OUT/energy.log 2>&1
cp gpurun_out/energy/table.txt $OUT/energy_table.txt
grep -v amdgpu $OUT/chain_probe.txt; cat $OUT/long_fir.jsonl; tail -4 $OUT/hostcall.jsonl; tail -12 $OUT/energy_table.txt
