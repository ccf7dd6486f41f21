"""Throughput of shapes OFF the BASELINE configs (other tap counts, odd channel counts, generic resampler
phases, wide mixes): a scan for pathological paths, one JSON line per shape.  GPU box only."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pipe_amd import processors as P, synth  # noqa: E402

st = torch.cuda.Stream()


def run(name, p, d_in, d_out, frames, samples, reps=10):
    for _ in range(3):
        p.process_batch(d_in, d_out, frames, stream=st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        p.process_batch(d_in, d_out, frames, stream=st.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"what": name, "kernel": p.kernel_name(), "ms": round(dt * 1e3, 4), "gsamples_per_s": round(samples / dt / 1e9, 1)}),
          flush=True)


F = 4096
if len(sys.argv) > 1 and sys.argv[1] == "resampler":
    for (up, down, T, C) in ((160, 147, 24, 2), (147, 160, 24, 2), (3, 2, 20, 2), (2, 3, 20, 2), (160, 147, 48, 2), (320, 147, 24, 2),
                             (160, 147, 24, 3), (160, 147, 10, 6), (1, 2, 64, 2), (2, 1, 64, 2), (7, 5, 13, 1)):
        K = 1024
        proto = synth.resampler_proto(up, down, T)
        n_in = K * F
        x = torch.rand(n_in * C, dtype=torch.float32, device="cuda")
        cap = -(-n_in * up // down) + 1
        y = torch.empty(cap * C, dtype=torch.float32, device="cuda")
        with P.Resampler(proto, T, up, down, F, C, dtype=np.float32, max_batch=K) as p:
            p.start()
            n_out = 0
            for _ in range(3):
                n_out = p.resample_batch(x, n_in, y, cap, stream=st.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                p.resample_batch(x, n_in, y, cap, stream=st.cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            print(json.dumps({"what": f"resampler {up}/{down} T={T} C={C}", "kernel": p.kernel_name(), "ms": round(dt * 1e3, 4),
                              "gsamples_out_per_s": round(n_out * C / dt / 1e9, 1), "gfma_per_s": round(n_out * C * T / dt / 1e9, 0)}),
                  flush=True)
    sys.exit(0)
for ntaps in (16, 64, 255, 256, 300, 512, 1024, 2048, 4096):
    for L, C, K in ((64, 2, 64), (512, 8, 1), (64, 3, 16)):
        taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
        with P.Fir(taps, F, C, dtype=np.float32, lines=L, max_batch=K) as p:
            p.start()
            n = L * K * F * C
            x = torch.rand(n, dtype=torch.float32, device="cuda")
            y = torch.empty_like(x)
            run(f"fir taps={ntaps} lines={L} C={C} K={K}", p, x, y, K * F, n)
