# long FIRs (partitioned overlap-save): kernel time per tap count; with a -DPH_OLSD_PROF library
# (PROF_LIB=path) also the delay-line kernel's phase profile
for n in ${TAPS:-1024 2048 4096}; do
  python bench.py --taps $n --no-secondary --no-cpu-baseline --no-live-pmc --steps 20 --warmup 3 --buffers 32768 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({'taps': $n, 'kernel': r['kernel'], 'avg_kernel_ms': r['avg_kernel_ms'], 'gsamples_per_s': round(d['value']/1e3,1)}))"
  if [ -n "$PROF_LIB" ]; then
    PIPE_HIP_LIB=$PROF_LIB python bench.py --taps $n --no-secondary --no-cpu-baseline --no-live-pmc --steps 6 --warmup 2 --buffers 32768 2>&1 | grep -E "prof" | cut -c1-200
  fi
done
