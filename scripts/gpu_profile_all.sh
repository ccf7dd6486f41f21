#!/bin/bash
# On the GPU box: rocprofv3 kernel trace + PMC passes for every bench.py --config and for
# scripts/bench_configs.py (every BASELINE config's kernels).  Raw traces are dropped after they
# are summarized (gpurun_out/ is capped at 64 MiB); copy gpurun_out/prof/TAG_*/summary.txt and
# pmc_entry.json into profiles/.
#   scripts/gpu_profile_all.sh TAG
set -u
TAG=${1:-r02}
for c in 1 2 3; do
  PMC_SECONDARY=$([ $c = 1 ] && echo 1 || echo 0) bash scripts/profile.sh ${TAG}_c$c --config $c > /dev/null 2>&1
  find gpurun_out/prof/${TAG}_c$c -name '*.db' -delete
  find gpurun_out/prof/${TAG}_c$c -type d -empty -delete
  tail -5 gpurun_out/prof/${TAG}_c$c/pmc_entry.json
done
bash scripts/profile_configs.sh ${TAG}_configs > /dev/null 2>&1
find gpurun_out/prof/${TAG}_configs -name '*.db' -delete
find gpurun_out/prof/${TAG}_configs -type d -empty -delete
head -40 gpurun_out/prof/${TAG}_configs/summary.txt | cut -c1-200
