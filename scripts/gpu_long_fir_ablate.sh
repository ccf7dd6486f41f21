# delay-line kernel with one kind of memory access removed at a time (scripts/build_ablate_lib.sh
# fir_ols32p PH_OLSD_ABLATE olsd 1 2 4 8 16 31): what each one costs the launch
for n in ${TAPS:-1024 2048}; do
for v in 0 1 2 4 8 16 31; do
  lib=$PWD/pipe_amd/lib/libpipe_hip_olsd$v.so; [ $v = 0 ] && lib=$PWD/pipe_amd/lib/libpipe_hip.so
  PIPE_HIP_LIB=$lib python bench.py --taps $n --no-secondary --no-cpu-baseline --no-live-pmc --steps 10 --warmup 3 --buffers 32768 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({'ablate': $v, 'taps': $n, 'avg_kernel_ms': r['avg_kernel_ms']}))"
done
done
