#!/bin/bash
# resampler_rows.hip loads an output's taps into SCALAR registers with hand-written `s_load_dwordx8` statements whose
# results are valid only behind a later hand-written `s_waitcnt lgkmcnt(0)` -- the compiler does not know the asm
# statements are loads (a seen failure: it handed the registers to address arithmetic, the late data landed in a
# pointer -- "write access to a read-only page").  Nothing at build time used to check the order it leaves them in.
# This script compiles the file to gfx950 assembly and, for every kernel `resample_rows_kernel<...>`:
#   1. from every asm-statement `s_load_dwordx8 s[a:b]` it walks the control-flow graph forwards and asserts that no
#      instruction reads or writes any of s[a:b] before an `s_waitcnt` with `lgkmcnt(0)` is passed on that path;
#   2. from the first tap load on, no tap register is spilled (`v_writelane_b32 v, s[a..b]`): the 2 T scalar registers
#      of taps stay put;
#   3. no VGPR spills at all, and the SGPR count stays within the 102 + VCC the part has.
# Exit 0 = holds; prints what does not.  tests/test_abi_surface.py runs it (a ROCm bump that reorders the loads fails a
# CPU test instead of faulting on a GPU).
set -u
cd "$(dirname "$0")/../pipe_amd/csrc" || exit 2
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
if [ -n "${1:-}" ]; then
  S=$1  # (an assembly file made earlier: the checker's own test feeds it a broken one)
else
  S=$(mktemp /tmp/rows_asm.XXXXXX.s)
  trap 'rm -f "$S"' EXIT
  # the flags of the Makefile's object rule (no per-file scheduling strategy is set for this file)
  $HIPCC -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I. --offload-device-only -S \
      resampler_rows.hip -o "$S" 2> /dev/null || { echo "check_rows_asm: compile failed"; exit 2; }
  [ -n "${ROWS_ASM_KEEP:-}" ] && cp "$S" "$ROWS_ASM_KEEP"
fi
python3 - "$S" <<'PY'
import re
import sys

text = open(sys.argv[1]).read().split("\n")
bad = []
kernels = 0

SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs(operands):
    out = set()
    for m in SREG.finditer(operands):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


i = 0
while i < len(text):
    m = re.match(r"^(_Z\S*resample_rows_kernel\S*):\s*(;.*)?$", text[i])
    if not m:
        i += 1
        continue
    name = m.group(1)
    kernels += 1
    # the function's instructions, labels resolved to instruction indices
    ins, labels, in_asm, marked = [], {}, False, set()
    j = i + 1
    while j < len(text) and not text[j].startswith(".Lfunc_end"):
        line = text[j].split(";", 1)[0].rstrip() if not text[j].lstrip().startswith(";;#ASM") else text[j].strip()
        if line == ";;#ASMSTART":
            in_asm = True
        elif line == ";;#ASMEND":
            in_asm = False
        else:
            lm = re.match(r"^(\.LBB\S+|\.L\S+):", line)
            if lm:
                labels[lm.group(1)] = len(ins)
            elif line.startswith("\t") and line.strip() and not line.strip().startswith("."):
                if in_asm and line.strip().startswith("s_load_dwordx8"):
                    marked.add(len(ins))
                ins.append(line.strip())
        j += 1
    loads = sorted(marked)
    if len(loads) < 8:
        bad.append(f"{name}: only {len(loads)} hand-written s_load_dwordx8 found (the kernel changed shape: update this script)")
    tap_regs = set()
    for k in loads:
        mm = re.match(r"s_load_dwordx8\s+s\[(\d+):(\d+)\]\s*,\s*(.*)", ins[k])
        if not mm:
            bad.append(f"{name}: cannot parse `{ins[k]}`")
            continue
        dst = set(range(int(mm.group(1)), int(mm.group(2)) + 1))
        tap_regs |= dst
        # forward walk over the control-flow graph until an lgkmcnt(0) wait is passed on every path
        seen, work = set(), [k + 1]
        while work:
            p = work.pop()
            while p < len(ins) and p not in seen:
                seen.add(p)
                op, _, rest = ins[p].partition(" ")
                if op == "s_waitcnt" and "lgkmcnt(0)" in rest:
                    break
                if op == "s_endpgm":
                    break
                touched = sregs(rest) & dst
                if touched and not (p in marked and False):
                    bad.append(f"{name}: `{ins[p]}` touches s{sorted(touched)} between `{ins[k]}` and its s_waitcnt lgkmcnt(0)")
                    break
                if op == "s_branch":
                    p = labels.get(rest.strip(), len(ins))
                    continue
                if op.startswith("s_cbranch"):
                    t = labels.get(rest.strip())
                    if t is not None:
                        work.append(t)
                p += 1
    # (the same physical registers hold other values in the kernel's prologue, and those may be spilled: a tap is in
    # them from the first hand-written load on)
    for p, line in enumerate(ins):
        if loads and p > loads[0] and line.startswith("v_writelane_b32"):
            ops = line.split(",")
            if len(ops) >= 2 and sregs(ops[1]) & tap_regs:
                bad.append(f"{name}: a tap register is spilled: `{line}`")
    i = j

# resource usage from the metadata notes (one block per kernel)
for m in re.finditer(r"\.name:\s+(\S*resample_rows_kernel\S*)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", "\n".join(text)):
    if int(m.group(3)) != 0:
        bad.append(f"{m.group(1)}: {m.group(3)} VGPRs spilled")
    if int(m.group(2)) > 106:
        bad.append(f"{m.group(1)}: sgpr_count {m.group(2)} > 106")

if kernels < 4:
    bad.append(f"only {kernels} resample_rows_kernel instantiations found (expected T = 8, 12, 16, 24)")
if bad:
    print("\n".join(bad[:40]))
    sys.exit(1)
print(f"resample_rows_kernel: {kernels} instantiations; every hand-written tap load is waited for before its registers are touched; no tap register spilled")
PY
