#!/bin/bash
# The resampler at BASELINE configs[4]'s launch (1 Line, 1024 buffers of 4096 x 2 float32, 160/147, 24 taps a phase):
# phase profile (s_memtime stamps, scripts/build_prof_resampler.sh) of the wave kernel and of the workgroup-tiled pair
# kernel it replaces, waves per CU, and the SQ counters of the shipped library.
OUT=${1:-gpurun_out/r05}; mkdir -p $OUT
R=$OUT/resampler_wave_profile.txt; : > $R
PROF=$PWD/pipe_amd/lib/libpipe_hip_prof.so
for w in 12 8 16; do
  echo "== resample_wave_kernel, $w waves per CU (phase profile build)" >> $R
  PIPE_HIP_LIB=$PROF PIPE_HIP_RESAMPLE_WAVES_PER_CU=$w PYTHONPATH=$PWD timeout 100 python scripts/bench_resampler.py 2>&1 | grep -E "160/147 C=2|wave prof" >> $R
done
echo "== resample_pair_kernel (PIPE_HIP_RESAMPLE_NO_WAVE=1: round 4's kernel, same build, same box)" >> $R
PIPE_HIP_LIB=$PROF PIPE_HIP_RESAMPLE_NO_WAVE=1 PYTHONPATH=$PWD timeout 100 python scripts/bench_resampler.py 2>&1 | grep -E "160/147 C=2|resampler prof" >> $R
echo "== the shipped library, every shape of scripts/bench_resampler.py" >> $R
PYTHONPATH=$PWD timeout 100 python scripts/bench_resampler.py 2>&1 | grep -v amdgpu.ids >> $R
./scripts/gpu_resampler_pmc.sh $OUT/resampler_pmc > /dev/null 2>&1
python - "$OUT" >> $R <<'PY'
import csv, collections, sys
out = sys.argv[1]
print("== SQ counters of resample_wave_kernel<f32,f32,24 taps> per launch and per wave-step (35 670 wave-steps a launch; rocprofv3 --pmc, two passes)")
for tag in "ab":
    f = f"{out}/resampler_pmc/{tag}/{tag}_counter_collection.csv"
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "resample_wave_kernel<float, float, 24, 0>" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c, v in sorted(acc.items()):
        print(f"   {c:24s} {v / max(n[c], 1):16.1f} per launch  ({v / max(n[c], 1) / 35670:8.1f} per wave-step)")
PY
cat $R
