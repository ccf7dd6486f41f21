#!/usr/bin/env python3
"""PCIe-inclusive rate of a large synchronous host call (pipe_hip_process, host buffers in and
out): BASELINE configs[3]'s chain on 512 Lines x 4096 x 8 float32 (67 MB each way), and
configs[2]'s FIR on 64 Lines x 4096 x 2.  A/B: chunks of Lines with overlapped transfers
(default) against the serial path (PIPE_HIP_OVERLAP_MIN_BYTES above the call size)."""
import ctypes as CT
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pipe_amd import _lib as LIB  # noqa: E402
from pipe_amd import processors as P  # noqa: E402
from pipe_amd import synth  # noqa: E402


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def run(name, mk, L_, F, C, reps):
    x = np.ascontiguousarray(np.random.default_rng(1).uniform(-1, 1, (L_, F, C)).astype(np.float32))
    y = np.empty_like(x)
    n = CT.c_int32()
    fn = LIB.lib().pipe_hip_process
    for label, thresh in (("overlapped chunks of Lines", None), ("serial (memcpy, H2D, kernels, D2H, memcpy)", str(1 << 40))):
        if thresh is None:
            os.environ.pop("PIPE_HIP_OVERLAP_MIN_BYTES", None)
        else:
            os.environ["PIPE_HIP_OVERLAP_MIN_BYTES"] = thresh
        with mk() as p:
            p.start()
            dt = timed(lambda: LIB.check(fn(p._h, x.ctypes.data, F, y.ctypes.data, F, CT.byref(n)), "process"), reps)
        print(json.dumps({"what": name, "path": label, "ms_per_call": round(dt * 1e3, 3),
                          "msamples_per_s": round(x.size / dt / 1e6, 1),
                          "host_gb_s_each_way": round(x.nbytes / dt / 1e9, 2)}), flush=True)


taps = synth.fir_lowpass_taps(256, f32_rounded=True)
q = synth.biquad_rbj_lowpass()
kw3 = dict(dtype=np.float32, lines=512, max_batch=1)
run("configs[3] chain from host buffers: 512 Lines x 4096 x 8 f32",
    lambda: P.Chain([P.Fir(taps, 4096, 8, **kw3), P.Biquad(q, 4096, 8, **kw3), P.Gain(0.7071067811865476, 4096, 8, **kw3)]),
    512, 4096, 8, 20)
kw2 = dict(dtype=np.float32, lines=2048, max_batch=1)
run("FIR-256 from host buffers: 2048 Lines x 4096 x 2 f32",
    lambda: P.Fir(taps, 4096, 2, **kw2), 2048, 4096, 2, 20)
