#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE per known byte for the library's access patterns
# (scripts/micro/fetch_calib.hip).  Output: gpurun_out/prof/fetch_calib.txt (copy into profiles/).
set -u
OUT=$PWD/gpurun_out/prof/fetch_calib
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/micro/fetch_calib.hip -o /tmp/fetch_calib 2> $OUT/build.err || { cat $OUT/build.err; exit 1; }
cd /tmp
/tmp/fetch_calib > $OUT/bytes.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o pmc -- /tmp/fetch_calib > /dev/null 2> $OUT/pmc_$c.err
done
cd $REPO
python - "$OUT" <<'PY' | tee $OUT/../fetch_calib.txt
import collections, glob, json, os, sqlite3, sys
out = sys.argv[1]
meta = json.load(open(os.path.join(out, "bytes.json")))
print("FETCH_SIZE / WRITE_SIZE (KiB x 1024) per byte actually moved, gfx950, rocprofv3 of this image")
print("(each kernel moves a known byte count exactly once; %d dispatches each)" % meta["reps"])
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(float)
    names = {}
    for p in glob.glob(os.path.join(out, "pmc_" + counter, "**", "*.db"), recursive=True):
        db = sqlite3.connect(p)
        for name, cname, val, disp in db.execute("select name, counter_name, counter_value, dispatch_id from pmc_events"):
            if cname == counter and "pipehip_calib" in name:
                per[disp] += val
                names[disp] = name
    disps = sorted(per)
    k = len(meta["order"])
    print(f"\n{counter}:")
    for i, label in enumerate(meta["order"]):
        vals = [per[d] * 1024.0 for j, d in enumerate(disps) if j % k == i]
        if not vals:
            continue
        b = meta["bytes"][i]
        mean = sum(vals) / len(vals)
        extra = ""
        if "overlap" in label:
            extra = f"   (unique bytes {meta['unique_bytes_overlap']}: counter / unique = {mean / meta['unique_bytes_overlap']:.4f})"
        print(f"  {label:28s} bytes {b:>12d}  counter {mean:>14.0f}  counter/bytes {mean / b:.4f}  n={len(vals)}{extra}")
PY
find $OUT -name '*.db' -delete
