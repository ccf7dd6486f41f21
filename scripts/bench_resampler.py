"""A/B timing of the resampler kernels at BASELINE configs[5]'s shape (44.1 -> 48 kHz, 24 taps per
phase, 1024 buffers of 4096x2 float32 per launch) and its reverse; environment switches select the
kernel (PIPE_HIP_RESAMPLE_PLANES=1: per-channel planes, the round-1 kernel)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pipe_amd import processors as P, synth  # noqa: E402

torch.cuda.set_stream(torch.cuda.Stream())  # device-resident calls launch directly on this stream (processors._TorchOrder)

for up, down, C in ((160, 147, 2), (147, 160, 2), (160, 147, 8), (2, 1, 2)):
    F, T, K = 4096, 24, 1024
    proto = synth.resampler_proto(up, down, T)
    n_in = K * F * 2 // C
    d_in = torch.empty(n_in * C, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    cap = -(-n_in * up // down) + 1
    d_out = torch.empty(cap * C, dtype=torch.float32, device="cuda")
    with P.Resampler(proto, T, up, down, F, C, dtype=np.float32, max_batch=n_in // F) as p:
        p.start()
        p.set_profiling(True)
        for _ in range(3):
            n = p.resample_batch(d_in, n_in, d_out, cap)
        p.kernel_time()
        for _ in range(20):
            n = p.resample_batch(d_in, n_in, d_out, cap)
        torch.cuda.synchronize()
        ms, cnt = p.kernel_time()
        ms /= max(cnt, 1)
        gb = (n_in + n) * C * 4 / 1e9
        print(f"{up}/{down} C={C} {p.kernel_name():36s} {ms*1e3:8.1f} us  {gb/ms*1e3:8.1f} GB/s  frac {gb/ms*1e3/8000:.3f}", flush=True)
