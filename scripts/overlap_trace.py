#!/usr/bin/env python3
"""One traced large host call (PIPE_HIP_OVERLAP_TRACE=1): per-chunk timestamps of the overlapped path."""
import ctypes as CT
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pipe_amd import _lib as LIB  # noqa: E402
from pipe_amd import processors as P  # noqa: E402
from pipe_amd import synth  # noqa: E402

L_, F, C = 512, 4096, 8
taps = synth.fir_lowpass_taps(256, f32_rounded=True)
kw = dict(dtype=np.float32, lines=L_, max_batch=1)
x = np.random.default_rng(1).uniform(-1, 1, (L_, F, C)).astype(np.float32)
y = np.empty_like(x)
n = CT.c_int32()
with P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(synth.biquad_rbj_lowpass(), F, C, **kw), P.Gain(0.5, F, C, **kw)]) as p:
    p.start()
    for k in range(4):
        if k == 3:
            os.environ["PIPE_HIP_OVERLAP_TRACE"] = "1"
        LIB.check(LIB.lib().pipe_hip_process(p._h, x.ctypes.data, F, y.ctypes.data, F, CT.byref(n)), "process")
