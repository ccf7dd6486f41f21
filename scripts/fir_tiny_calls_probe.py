#!/usr/bin/env python3
"""The ordered FIR on the smallest device-resident calls: the VALU small-call kernel (below 32 passes of 1024 frames x
2 channels) against the matrix-pipe form forced (PIPE_HIP_FIR_MFMA_MIN_PASSES=1), over tap counts -- the rule was set
at 256 taps.  scripts/fir_tiny_calls_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pipe_amd import processors as P, synth  # noqa: E402

F, C = 4096, int(os.environ.get("PROBE_C", "2"))
st = torch.cuda.Stream()
SHAPES = [(1, 1), (1, 2), (1, 4), (1, 7), (4, 1), (7, 1)]
for N in (int(v) for v in os.environ.get("PROBE_TAPS", "32,256,1024,4096").split(",")):
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    for lines, K in SHAPES:
        n = lines * K * F * C
        d_in = torch.empty(n, dtype=torch.float32, device="cuda")
        P.synth_fill(d_in, synth.line_seed(0))
        d_out = torch.empty_like(d_in)
        row = []
        for knob in (None, "1"):
            if knob is None:
                os.environ.pop("PIPE_HIP_FIR_MFMA_MIN_PASSES", None)
            else:
                os.environ["PIPE_HIP_FIR_MFMA_MIN_PASSES"] = knob
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
                row.append((p.kernel_name(), dt * 1e6, kms / max(kn, 1) * 1e3))
        os.environ.pop("PIPE_HIP_FIR_MFMA_MIN_PASSES", None)
        print(f"{N:5d} taps {lines:3d} Lines x {K:3d} buffers: default {row[0][0]:26s} {row[0][1]:7.1f} us a call (kernel {row[0][2]:7.1f})"
              f"   matrix pipe forced {row[1][0]:26s} {row[1][1]:7.1f} us a call (kernel {row[1][2]:7.1f})", flush=True)
