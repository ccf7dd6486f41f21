#!/usr/bin/env python3
"""The time-segmented biquad (large float32 calls) over shapes of the same sample count: many Lines of few
frames to ONE long stream.  scripts/biquad_shapes_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pipe_amd import processors as P, synth  # noqa: E402

F = int(os.environ.get("PROBE_F", "4096"))   # frames per buffer
st = torch.cuda.Stream()
S = int(os.environ.get("PROBE_SECTIONS", "1"))
q = np.vstack([synth.biquad_rbj_lowpass(), synth.biquad_rbj_lowpass(fc=4000.0, q=1.3), synth.biquad_rbj_lowpass(fc=300.0, q=4.0),
               synth.biquad_rbj_lowpass(fc=2500.0, q=0.9)][:S])
SHAPES = os.environ.get("PROBE_SHAPES")   # "lines,channels,buffers;..." instead of the list below
for lines, C, K in [tuple(int(v) for v in t.split(",")) for t in SHAPES.split(";")] if SHAPES else ((4096, 1, 1), (2048, 2, 1), (512, 8, 1), (64, 2, 32), (8, 2, 256), (1, 2, 2048), (64, 1, 64), (1, 1, 4096), (1, 8, 512),
                     (1024, 3, 1), (4, 3, 256), (512, 6, 1), (2, 6, 256), (1, 5, 512), (2048, 8, 1), (4096, 8, 1), (128, 8, 4), (256, 7, 2), (4096, 6, 1)):
    n = lines * K * F * C
    if os.environ.get("PROBE_ONLY_C") and str(C) not in os.environ["PROBE_ONLY_C"].split(","):
        continue
    d_in = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty_like(d_in)
    kw = dict(dtype=np.float32, lines=lines, max_batch=K)
    # PROBE_CHAIN=gb: gain -> biquad as a staged chain (the biquad reads float64, as behind an overlap-save FIR)
    mk = (lambda: P.Chain([P.Gain(1.0, F, C, **kw), P.Biquad(q, F, C, **kw)])) if os.environ.get("PROBE_CHAIN") == "gb" \
        else (lambda: P.Biquad(q, F, C, **kw))
    with mk() as p:
        p.start()
        for _ in range(10):
            p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        print(f"biquad {lines:5d} Lines x {C} ch x {K:4d} buffers: {p.kernel_name():36s} {dt * 1e3:8.4f} ms  {n / dt / 1e9:7.1f} Gsamples/s", flush=True)
