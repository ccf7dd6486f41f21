#!/usr/bin/env python3
"""Where the overlap-save FIR starts to beat the ordered form on the matrix pipe: device-resident float32 calls of
(Lines, buffers) from a few hundred to a few thousand 1024-point transforms, default dispatch against
PIPE_HIP_FIR_OLS_MIN_ITEMS=1 (always overlap-save).  scripts/fir_small_calls_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pipe_amd import processors as P, synth  # noqa: E402

F, C, N = 4096, int(os.environ.get("PROBE_C", "2")), int(os.environ.get("PROBE_TAPS", "256"))
taps = synth.fir_lowpass_taps(N, f32_rounded=True)
st = torch.cuda.Stream()
SHAPES = [(1, 8), (1, 16), (1, 32), (1, 64), (1, 96), (1, 128), (1, 192), (1, 256), (1, 384), (1, 512), (1, 1024),
          (8, 1), (16, 1), (32, 1), (64, 1), (128, 1), (256, 1), (512, 1), (64, 4)]
if os.environ.get("PROBE_SHAPES"):   # "lines,buffers;..."
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["PROBE_SHAPES"].split(";")]
for lines, K in SHAPES:
    n = lines * K * F * C
    items = lines * -(-K * F // (1025 - N if N <= 512 else 512)) * -(-C // 2)
    d_in = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty_like(d_in)
    row = []
    for knob in (None, "1"):
        if knob is None:
            os.environ.pop("PIPE_HIP_FIR_OLS_MIN_ITEMS", None)
        else:
            os.environ["PIPE_HIP_FIR_OLS_MIN_ITEMS"] = knob
        with P.Fir(taps, F, C, dtype=np.float32, lines=lines, max_batch=K) as p:
            p.start()
            for _ in range(30):
                p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
            torch.cuda.synchronize()
            p.set_profiling(True)
            p.kernel_time(reset=True)
            t0 = time.perf_counter()
            for _ in range(200):
                p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 200
            kms, kn = p.kernel_time(reset=True)
            p.set_profiling(False)
            row.append((p.kernel_name(), dt * 1e6, kms / max(kn, 1) * 1e3))
    os.environ.pop("PIPE_HIP_FIR_OLS_MIN_ITEMS", None)
    print(f"{lines:4d} Lines x {K:5d} buffers ({items:6d} transforms): default {row[0][0]:32s} {row[0][1]:8.1f} us a call (kernel {row[0][2]:7.1f})"
          f"   forced {row[1][0]:32s} {row[1][1]:8.1f} us a call (kernel {row[1][2]:7.1f})", flush=True)
