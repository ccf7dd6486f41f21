"""Soak test of the overlapped large-call path (abi.hip process_overlapped: chunks of Lines, three streams,
a pool of copy threads, uploads through the BAR): random shapes, stage kinds and pass sequences, every call
compared bit for bit with the serial path on a second handle fed the same stream; two handles driven from
two threads at once.  scripts/stress_hostcall.py [iterations] [seed]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# bit-for-bit comparison of two call sizes: pin the ordered forms (a large serial call may take the
# time-segmented biquad or the overlap-save FIR where its chunks do not: both within the contract, not equal)
os.environ["PIPE_HIP_FIR_EXACT"] = "1"
os.environ["PIPE_HIP_BIQUAD_EXACT"] = "1"
from pipe_amd import processors as P, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Q = synth.biquad_rbj_lowpass()
errors = []


def make(kind, F, C, lines, dtype, taps):
    kw = dict(dtype=dtype, lines=lines, max_batch=1)
    if kind == "gain":
        return P.Gain(0.37, F, C, **kw)
    if kind == "fir":
        return P.Fir(taps, F, C, **kw)
    if kind == "biquad":
        return P.Biquad(Q, F, C, **kw)
    return P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(Q, F, C, **kw), P.Gain(0.5, F, C, **kw)])


def cases(rng, n):
    out = []
    for _ in range(n):
        kind = str(rng.choice(["gain", "fir", "biquad", "chain"]))
        F = int(rng.choice([64, 256, 1024]))
        C = int(rng.choice([1, 2, 4, 8]))
        lines = int(rng.integers(2, 200))
        dtype = np.float32 if rng.random() < 0.5 else np.float64
        ntaps = int(rng.choice([8, 48, 130]))
        passes = int(rng.integers(1, 5))
        use_lines = bool(rng.random() < 0.5)
        frames = [F if rng.random() < 0.7 else int(rng.integers(1, F + 1)) for _ in range(passes)]
        xs = [rng.uniform(-1, 1, size=(lines, f, C)).astype(dtype) for f in frames]
        out.append((kind, F, C, lines, dtype, ntaps, use_lines, xs))
    return out


def run_case(case):
    kind, F, C, lines, dtype, ntaps, use_lines, xs = case
    taps = synth.fir_lowpass_taps(ntaps)
    outs = []
    with make(kind, F, C, lines, dtype, taps) as h:
        h.start()
        for k, x in enumerate(xs):
            if use_lines:
                rows = [x[l] for l in range(lines)]
                if k == len(xs) - 1 and lines > 3:
                    rows[2] = None
                got = h.process_lines(rows)
                outs.append(np.stack([g if g is not None else np.zeros(x.shape[1:], dtype) for g in got]))
            else:
                outs.append(h.process(x))
        h.flush()
    return outs


def check(tid, todo, refs):
    try:
        for i, case in enumerate(todo):
            got = run_case(case)
            for a, b in zip(got, refs[i]):
                assert np.array_equal(a, b), (tid, i, case[:7])
    except Exception as e:  # noqa: BLE001
        errors.append(e)


t0 = time.time()
rng = np.random.default_rng(seed)
todo = cases(rng, iters)
os.environ["PIPE_HIP_OVERLAP_MIN_BYTES"] = str(1 << 40)      # the serial path: the reference
refs = [run_case(c) for c in todo]
os.environ["PIPE_HIP_OVERLAP_MIN_BYTES"] = "1"               # every call with two or more Lines is chunked
check(0, todo, refs)                                         # calls back to back
half = len(todo) // 2
ts = [threading.Thread(target=check, args=(1, todo[:half], refs[:half])),
      threading.Thread(target=check, args=(2, todo[half:], refs[half:]))]   # two handles' pools at the same time
[t.start() for t in ts]
[t.join() for t in ts]
assert not errors, errors[0]
print(f"stress ok: {len(todo)} shapes x (serial, overlapped, overlapped from two threads) in {time.time() - t0:.0f} s")
