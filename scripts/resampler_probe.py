#!/usr/bin/env python3
"""The resampler launch bench.py times for configs[4] (160/147, 24 taps per phase, 1024 buffers of
4096 x 2 float32), alone in a process: for rocprofv3 --pmc passes and A/B runs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pipe_amd import processors as P  # noqa: E402
from pipe_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
T, up, down, F, C, K = 24, 160, 147, 4096, int(os.environ.get("PROBE_CHANNELS", "2")), 1024
n_in = K * F
cap = -(-n_in * up // down) + 1
d_in = torch.empty(n_in * C, dtype=torch.float32, device="cuda")
P.synth_fill(d_in, synth.line_seed(0))
d_out = torch.empty(cap * C, dtype=torch.float32, device="cuda")
st = torch.cuda.Stream()
with P.Resampler(synth.resampler_proto(up, down, T), T, up, down, F, C, dtype=np.float32, max_batch=K) as p:
    p.start()
    for _ in range(3):
        p.resample_batch(d_in, n_in, d_out, cap, stream=st.cuda_stream)
    torch.cuda.synchronize()
    p.set_profiling(True)
    p.kernel_time(reset=True)
    for _ in range(reps):
        n_out = p.resample_batch(d_in, n_in, d_out, cap, stream=st.cuda_stream)
    torch.cuda.synchronize()
    ms, n = p.kernel_time(reset=True)
    print(p.kernel_name(), "avg kernel ms", ms / max(n, 1), "launches", n, "in bytes", n_in * C * 4, "out bytes", n_out * C * 4)
