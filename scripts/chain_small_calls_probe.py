#!/usr/bin/env python3
"""The FIR -> biquad -> gain chain on small and medium device-resident calls: host time a call of the library's own
dispatch (staged below 2 transforms a CU, fused above) against the fused kernel forced (PIPE_HIP_FIR_OLS_MIN_ITEMS=1)
and against the staged chain forced (a threshold nothing reaches).  scripts/chain_small_calls_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pipe_amd import processors as P, synth  # noqa: E402

F, N = 4096, int(os.environ.get("PROBE_TAPS", "256"))
taps = synth.fir_lowpass_taps(N, f32_rounded=True)
q = synth.biquad_rbj_lowpass()
st = torch.cuda.Stream()
SHAPES = [(4, 8, 1), (8, 8, 1), (16, 8, 1), (32, 8, 1), (64, 8, 1), (128, 8, 1), (256, 8, 1), (16, 2, 1), (64, 2, 1), (256, 2, 1), (1, 2, 64), (1, 2, 256), (4, 8, 8)]
if os.environ.get("PROBE_SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["PROBE_SHAPES"].split(";")]
for lines, C, K in SHAPES:
    n = lines * K * F * C
    items = lines * -(-K * F // 768) * -(-C // 2)
    d_in = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty_like(d_in)
    row = []
    for knob in (None, "1", "1000000000"):
        if knob is None:
            os.environ.pop("PIPE_HIP_FIR_OLS_MIN_ITEMS", None)
        else:
            os.environ["PIPE_HIP_FIR_OLS_MIN_ITEMS"] = knob
        kw = dict(dtype=np.float32, lines=lines, max_batch=K)
        with P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(q, F, C, **kw), P.Gain(0.7071067811865476, F, C, **kw)]) as p:
            p.start()
            for _ in range(30):
                p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
            torch.cuda.synchronize()
            row.append((p.kernel_name(), (time.perf_counter() - t0) / 200 * 1e6))
    os.environ.pop("PIPE_HIP_FIR_OLS_MIN_ITEMS", None)
    print(f"{lines:4d} Lines x {C} ch x {K:4d} buffers ({items:6d} transforms): default {row[0][0][:44]:44s} {row[0][1]:7.1f} us | fused forced {row[1][0][:30]:30s} {row[1][1]:7.1f} us"
          f" | staged forced {row[2][0][:26]:26s} {row[2][1]:7.1f} us", flush=True)
