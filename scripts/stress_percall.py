"""Soak test of the ProcessFunc form (one buffer per call, host pointers): random stage kinds, tap and
section counts, channel counts, buffer sizes and ragged call sequences, synchronous calls mixed with
submit/collect pairs and restarts, every buffer compared BIT FOR BIT with the oracle.  Exercises the
small-call FIR kernel (taps in LDS, history from the last tile's planes), the LDS forms of the exact
biquad (one lane per series / one lane per section) and the completion event on the last launch.
scripts/stress_percall.py [iterations] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from pipe_amd import processors as P, synth  # noqa: E402


class _Gain:  # the oracle's gain is a function
    def __init__(self, g):
        self.g = g

    def process(self, y):
        return O.gain(y, self.g)


iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def sections(n):
    return np.vstack([synth.biquad_rbj_lowpass(fc=float(rng.uniform(100, 8000)), q=float(rng.uniform(0.5, 2.0)))
                      for _ in range(n)])


t0 = time.time()
buffers = 0
for it in range(iters):
    kind = str(rng.choice(["fir", "fir", "biquad", "biquad", "chain", "gain"]))
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    C = int(rng.choice([1, 2, 2, 3, 8, 9, 17]))
    F = int(rng.choice([64, 512, 1000, 4096, 8192]))
    ntaps = int(rng.choice([1, 2, 5, 19, 20, 23, 64, 255, 256, 300, 1024]))
    S = int(rng.choice([1, 1, 2, 3, 5, 8]))
    taps = synth.fir_lowpass_taps(ntaps) if ntaps > 2 else np.array([0.75, -0.5][:ntaps])
    q = sections(S)
    g = float(rng.uniform(-2, 2))
    if kind == "fir":
        make, ref = (lambda: P.Fir(taps, F, C, dtype=dtype)), [O.Fir(taps, C)]
    elif kind == "biquad":
        make, ref = (lambda: P.Biquad(q, F, C, dtype=dtype)), [O.Biquad(q, C)]
    elif kind == "gain":
        make, ref = (lambda: P.Gain(g, F, C, dtype=dtype)), [_Gain(g)]
    else:
        make = lambda: P.Chain([P.Fir(taps, F, C, dtype=dtype), P.Biquad(q, F, C, dtype=dtype), P.Gain(g, F, C, dtype=dtype)])
        ref = [O.Fir(taps, C), O.Biquad(q, C), _Gain(g)]

    def want(x):
        y = x.astype(np.float64)
        for r in ref:
            y = np.asarray(r.process(y)).reshape(x.shape)
        return y.astype(dtype)

    ncalls = int(rng.integers(2, 9))
    lens = [int(rng.choice([F, F, F, 0, 1, 15, 16, 17, int(rng.integers(0, F + 1))])) for _ in range(ncalls)]
    x = rng.uniform(-1, 1, size=(sum(lens) + 1, C)).astype(np.float32).astype(dtype)
    # (round 4) a float32 biquad buffer of >= 1024 frames takes the tile form, which is not bit-exact: this soak is
    # about the ordered forms, so those handles are pinned to them; and every other armable handle (gain, FIR,
    # FIR -> gain) runs with the next buffer's work queued behind a doorbell (PIPE_HIP_PARAM_RESIDENT)
    resident = kind in ("fir", "gain") and C * F * np.dtype(dtype).itemsize <= (1 << 20) and rng.random() < 0.5
    with make() as p:
        p.start()
        if kind in ("biquad", "chain"):
            p.set_exact(True) if hasattr(p, "set_exact") else p._set_param(3, [1.0])
        if resident:
            p.set_resident(True)
        pos = 0
        i = 0
        while i < len(lens):
            n = lens[i]
            if i + 1 < len(lens) and rng.random() < 0.4:   # two buffers in flight
                m = lens[i + 1]
                p.submit(x[pos:pos + n])
                p.submit(x[pos + n:pos + n + m])
                got = [p.collect(), p.collect()]
                exp = [want(x[pos:pos + n]), want(x[pos + n:pos + n + m])]
                i += 2
                pos += n + m
            else:
                got = [p.process(x[pos:pos + n])]
                exp = [want(x[pos:pos + n])]
                i += 1
                pos += n
            for a, b in zip(got, exp):
                buffers += 1
                if a.shape != b.shape or not np.array_equal(a, b):
                    print("MISMATCH", dict(it=it, kind=kind, dtype=str(np.dtype(dtype)), C=C, F=F, ntaps=ntaps, S=S, lens=lens,
                                           at=pos, resident=bool(resident)), flush=True)
                    sys.exit(1)
print(f"ok: {iters} random handles, {buffers} buffers bit-exact, {time.time() - t0:.1f} s")
