"""Per-call (ProcessFunc form) cost of one 4096x2 buffer: FIR-256, gain, chain."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (HIP runtime first)
from pipe_amd import processors as P, synth  # noqa: E402

F, C = 4096, 2
taps = synth.fir_lowpass_taps(256, f32_rounded=True)
x = synth.samples(synth.line_seed(0), 0, F * C, np.float32).reshape(F, C)


def timed(fn, reps=400, warm=30):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


for dtype in (np.float32, np.float64):
    xin = x.astype(dtype)
    kw = dict(dtype=dtype)
    for name, mk in (("gain", lambda: P.Gain(0.5, F, C, **kw)), ("fir256", lambda: P.Fir(taps, F, C, **kw)),
                     ("biquad", lambda: P.Biquad(synth.biquad_rbj_lowpass(), F, C, **kw)),
                     ("chain fir+biquad+gain", lambda: P.Chain([P.Fir(taps, F, C, **kw),
                                                                P.Biquad(synth.biquad_rbj_lowpass(), F, C, **kw),
                                                                P.Gain(0.5, F, C, **kw)]))):
        with mk() as p:
            p.start()
            dt = timed(lambda: p.process(xin))
            print(json.dumps({"stage": name, "io": str(np.dtype(dtype)), "us_per_buffer": round(dt * 1e6, 2)}), flush=True)
