#!/usr/bin/env python3
"""The fused-chain launch of BASELINE configs[3] (512 Lines x 8 ch x 4096 frames, FIR-256 -> biquad -> gain),
alone in a process: kernel time from events on the kernel's dispatch, for A/B runs of environment knobs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pipe_amd import processors as P  # noqa: E402
from pipe_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
L, C, F, N = int(os.environ.get("PROBE_LINES", "512")), 8, 4096, 256
K = int(os.environ.get("PROBE_BUFFERS", "1"))   # consecutive buffers per Line and launch
sections = int(os.environ.get("PROBE_SECTIONS", "1"))
taps = synth.fir_lowpass_taps(N, f32_rounded=True)
q = np.vstack([synth.biquad_rbj_lowpass(1000.0 * (k + 1)) for k in range(sections)])
kw = dict(dtype=np.float32, lines=L, max_batch=K)
n = L * K * F * C
d_in = torch.empty(n, dtype=torch.float32, device="cuda")
P.synth_fill(d_in, synth.line_seed(0))
d_out = torch.empty_like(d_in)
st = torch.cuda.Stream()
with P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(q, F, C, **kw), P.Gain(0.7071067811865476, F, C, **kw)]) as p:
    p.start()
    for _ in range(600 if K == 1 else 40):
        p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
    torch.cuda.synchronize()
    p.set_profiling(True)
    p.kernel_time(reset=True)
    for _ in range(reps if K == 1 else max(20, reps // K)):
        p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
    torch.cuda.synchronize()
    ms, k = p.kernel_time(reset=True)
    p.flush()
    print(os.path.basename(os.environ.get("PIPE_HIP_LIB", "default")), f"K={K}", p.kernel_name(), "avg kernel ms", round(ms / max(k, 1), 5), "frac",
          round(n * 8 / (ms / max(k, 1) * 1e-3) / 8e12, 4))
