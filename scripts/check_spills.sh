#!/bin/bash
# Every kernel of the library must be free of VGPR spills: hipcc (ROCm 7.2) may place a spill
# store inside an exec-masked region, and the lanes that were masked off then reload garbage
# (seen in a two-section form of the fused chain kernel).  Reads the resource-usage
# remarks the build leaves next to every object (pipe_amd/csrc/build/*.remarks); fails if any
# kernel spills or if an object has no remarks.
cd "$(dirname "$0")/../pipe_amd/csrc" || exit 2
make -s -j8 || exit 2
bad=0
for f in *.hip; do
  r=build/${f%.hip}.remarks
  if [ ! -e "$r" ]; then echo "$f: no resource-usage remarks (rebuild: make clean all)"; bad=1; continue; fi
  out=$(grep -E "Function Name|VGPRs Spill" "$r" | sed 's/.*remark: *//;s/ \[.*//' | paste - - | awk '$NF != 0' | cut -c1-240)
  if [ -n "$out" ]; then echo "$f:"; echo "$out"; bad=1; fi
done
[ $bad = 0 ] && echo "no kernel spills VGPRs"
exit $bad
