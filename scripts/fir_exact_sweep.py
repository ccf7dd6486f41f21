#!/usr/bin/env python3
"""The ordered-fma FIR (256 taps, 2 channels, float32) by call size: the matrix-pipe kernel against the VALU
kernel, kernel time from events on the dispatch.  scripts/fir_exact_sweep.py  (A/B for the size threshold
of fir_mfma_takes)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:  # child: one form
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from pipe_amd import processors as P, synth
    F, C, N = 4096, 2, 256
    taps = synth.fir_lowpass_taps(N)
    st = torch.cuda.Stream()
    for lines, K in ((1, 1), (1, 2), (1, 4), (1, 8), (1, 16), (1, 32), (1, 64), (1, 128), (16, 1), (64, 1), (256, 1)):
        n = lines * K * F * C
        d_in = torch.empty(n, dtype=torch.float32, device="cuda")
        P.synth_fill(d_in, synth.line_seed(0))
        d_out = torch.empty_like(d_in)
        with P.Fir(taps, F, C, dtype=np.float32, lines=lines, max_batch=K) as p:
            p.start()
            p.set_exact(True)
            for _ in range(20):
                p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
            torch.cuda.synchronize()
            p.set_profiling(True)
            p.kernel_time(reset=True)
            for _ in range(200):
                p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
            torch.cuda.synchronize()
            ms, k = p.kernel_time(reset=True)
            print(f"{sys.argv[1]:5s} lines {lines:4d} x {K:4d} buffers: {p.kernel_name():28s} {ms / max(k, 1) * 1e3:9.1f} us", flush=True)
    sys.exit(0)
for form, env in (("mfma", {"PIPE_HIP_FIR_MFMA_MIN_PASSES": "1"}), ("valu", {"PIPE_HIP_FIR_NO_MFMA": "1"})):
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, os.path.abspath(__file__), form], env=e, check=False)
