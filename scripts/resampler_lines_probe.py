#!/usr/bin/env python3
"""Resampler 160/147 (24 taps per phase, 2 channels) over the same 4.19 M input frames cut into
1 / 64 / 512 / 1024 Lines: what a Line's first tile (its head is the history) and its ragged last one
cost a launch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pipe_amd import processors as P  # noqa: E402
from pipe_amd import synth  # noqa: E402

T, up, down, F, C = 24, 160, 147, 4096, 2
total = 1024 * F
st = torch.cuda.Stream()
for lines in (1, 64, 512, 1024):
    n_in = total // lines
    cap = -(-n_in * up // down) + 1
    d_in = torch.empty(total * C, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty(lines * cap * C, dtype=torch.float32, device="cuda")
    with P.Resampler(synth.resampler_proto(up, down, T), T, up, down, F, C, dtype=np.float32, lines=lines,
                     max_batch=n_in // F) as p:
        p.start()
        for _ in range(3):
            p.resample_batch(d_in, n_in, d_out, cap, stream=st.cuda_stream)
        torch.cuda.synchronize()
        p.set_profiling(True)
        p.kernel_time(reset=True)
        for _ in range(50):
            p.resample_batch(d_in, n_in, d_out, cap, stream=st.cuda_stream)
        torch.cuda.synchronize()
        ms, n = p.kernel_time(reset=True)
        print(f"lines {lines:5d} x {n_in:8d} frames  {p.kernel_name():34s} avg kernel ms {ms / max(n, 1):.5f}", flush=True)
