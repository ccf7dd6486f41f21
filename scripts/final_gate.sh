#!/bin/bash
# Run from the repo root in the build container as the LAST GPU call of a round:
#   - refuses a dirty tree (the gate must describe the committed HEAD),
#   - rebuilds nothing (the in-tree libraries are what travels; `make -q` says whether they are current),
#   - runs scripts/gpu_final_gate.sh on a fresh MI355X, copies its report to profiles/<tag>_gate.txt and commits it.
# No code commit may follow it (only this report).
TAG=${1:-r06}
if [ -n "$(git status --porcelain)" ]; then echo "tree is dirty: commit first"; git status --short | head; exit 1; fi
( cd pipe_amd/csrc && make -q ) || { echo "libpipe_hip.so is older than its sources: make first"; exit 1; }
SHA=$(git rev-parse HEAD)
for try in 1 2 3 4 5 6 7 8; do /usr/local/graft/bin/gpurun --timeout 1800 -- "scripts/gpu_final_gate.sh $SHA $TAG"; rc=$?; [ $rc -ne 3 ] && break; sleep 60; done; [ $rc -eq 0 ] || exit 1
cp gpurun_out/${TAG}_gate/gate.txt profiles/${TAG}_gate.txt
cp gpurun_out/${TAG}_gate/pytest.txt profiles/${TAG}_gate_pytest.txt
grep '^{' gpurun_out/${TAG}_gate/bench.txt | tail -1 > profiles/${TAG}_bench_default.json  # (the gate's own default bench line)
git add profiles/${TAG}_gate.txt profiles/${TAG}_gate_pytest.txt profiles/${TAG}_bench_default.json && git commit -q -m "$TAG gate: suite + smoke + bench of $SHA on a fresh MI355X" && echo committed
