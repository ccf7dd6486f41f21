#!/usr/bin/env python3
"""The 160/147 resampler (24 taps a phase) on small and medium device-resident calls: which kernel the library's
dispatch takes and what a call costs, over Lines x channels x buffers a call -- looking for shapes that fall between
the kernels' entry rules (as the FIR's and the biquad's did in round 6).  scripts/resampler_small_calls_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pipe_amd import processors as P, synth  # noqa: E402

T, up, down, F = 24, 160, 147, 4096
DT = np.float64 if os.environ.get("PROBE_DTYPE") == "f64" else np.float32
TDT = torch.float64 if DT is np.float64 else torch.float32
st = torch.cuda.Stream()
SHAPES = [(1, 2, 1), (1, 2, 4), (1, 2, 16), (1, 2, 64), (1, 2, 256), (16, 2, 1), (64, 2, 1), (256, 2, 1), (1024, 2, 1),
          (1, 8, 1), (1, 8, 16), (1, 8, 64), (16, 8, 1), (64, 8, 1), (256, 8, 1), (1, 1, 16), (64, 1, 1), (1, 4, 16), (64, 4, 1), (64, 3, 1), (64, 6, 1)]
if os.environ.get("PROBE_SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["PROBE_SHAPES"].split(";")]
for lines, C, K in SHAPES:
    n_in = K * F
    cap = -(-n_in * up // down) + 1
    d_in = torch.empty(lines * n_in * C, dtype=TDT, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty(lines * cap * C, dtype=TDT, device="cuda")
    with P.Resampler(synth.resampler_proto(up, down, T), T, up, down, F, C, dtype=DT, lines=lines, max_batch=K) as p:
        p.start()
        for _ in range(20):
            p.resample_batch(d_in, n_in, d_out, cap, stream=st.cuda_stream)
        torch.cuda.synchronize()
        p.set_profiling(True)
        p.kernel_time(reset=True)
        t0 = time.perf_counter()
        for _ in range(100):
            p.resample_batch(d_in, n_in, d_out, cap, stream=st.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        ms, n = p.kernel_time(reset=True)
        samples = lines * n_in * C
        print(f"{lines:5d} Lines x {C} ch x {K:4d} buffers: {p.kernel_name():34s} {dt * 1e6:8.1f} us a call (kernel {ms / max(n, 1) * 1e3:7.1f})  {samples / dt / 1e9:7.1f} Gsamples/s in", flush=True)
