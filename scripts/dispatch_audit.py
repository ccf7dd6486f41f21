#!/usr/bin/env python3
"""Audit of the library's size-dependent dispatch rules: for every shape of a sweep, the library's own choice against
each alternative form that an environment knob can force, host time a call (200 calls back to back on one stream).
A shape whose default is more than 25 % slower than an alternative is flagged -- round 6 found five rules that were
right only at the one shape their round had measured (profiles/r06_fir_small_calls.txt, r06_biquad_dispatch_gap.txt).
    python scripts/dispatch_audit.py            (one MI355X, ~1 minute)
Exit status 1 when a shape is flagged."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pipe_amd import processors as P, synth  # noqa: E402

F = 4096
st = torch.cuda.Stream()
flagged = []


def timed(make, call, env):
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        os.environ[k] = v
    try:
        with make() as p:
            p.start()
            for _ in range(20):
                call(p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                call(p)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 200 * 1e6, p.kernel_name()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def audit(what, make, call, alternatives, by_contract=False):
    base, name = timed(make, call, {})
    line = f"{what:58s} default {name[:40]:40s} {base:7.1f} us"
    worst = 1.0
    for label, env in alternatives:
        t, n = timed(make, call, env)
        line += f" | {label} {t:7.1f}"
        if n != name:
            worst = max(worst, base / t)
    if worst > 1.25 and by_contract:
        line += f"   ({worst:.2f} x: by contract -- float32 buffers below 1024 frames a Line keep the ordered forms, include/pipe_hip.h)"
    elif worst > 1.25:
        line += f"   <-- {worst:.2f} x slower than an alternative"
        flagged.append(what)
    print(line, flush=True)


# ---- FIR: the ordered form on the matrix pipe against overlap-save
for N in (32, 256, 512, 1024, 4096):
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    for lines, C, K in ((1, 2, 96), (1, 2, 256), (1, 2, 512), (1, 2, 1024), (96, 2, 1), (256, 2, 1), (512, 2, 1), (32, 8, 1), (1, 8, 64), (1, 1, 512)):
        n = lines * K * F * C
        d_in = torch.empty(n, dtype=torch.float32, device="cuda")
        P.synth_fill(d_in, synth.line_seed(0))
        d_out = torch.empty_like(d_in)
        audit(f"FIR {N:4d} taps, {lines:3d} Lines x {C} ch x {K:4d} buffers",
              lambda: P.Fir(taps, F, C, dtype=np.float32, lines=lines, max_batch=K),
              lambda p: p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream),
              [("overlap-save", {"PIPE_HIP_FIR_OLS_MIN_ITEMS": "1"}), ("ordered", {"PIPE_HIP_FIR_OLS_MIN_ITEMS": "1000000000"})])

# ---- biquad: the ordered forms against the time-segmented ones
for S in (1, 2, 4):
    q = np.vstack([synth.biquad_rbj_lowpass(), synth.biquad_rbj_lowpass(fc=4000.0, q=1.3), synth.biquad_rbj_lowpass(fc=300.0, q=4.0),
                   synth.biquad_rbj_lowpass(fc=2500.0, q=0.9)][:S])
    for lines, C, frames in ((1, 2, 4096), (16, 8, 4096), (100, 2, 4096), (1, 16, 4096), (256, 2, 1024), (64, 2, 512), (2048, 2, 256), (512, 8, 4096), (4, 3, 8192)):
        n = lines * frames * C
        d_in = torch.empty(n, dtype=torch.float32, device="cuda")
        P.synth_fill(d_in, synth.line_seed(0))
        d_out = torch.empty_like(d_in)
        audit(f"biquad {S} section(s), {lines:4d} Lines x {C:2d} ch x {frames:5d} frames",
              lambda: P.Biquad(q, frames, C, dtype=np.float32, lines=lines, max_batch=1),
              lambda p: p.process_batch(d_in, d_out, frames, stream=st.cuda_stream),
              [("segmented", {"PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES": "1"}), ("ordered", {"PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES": "1000000000000"})],
              by_contract=frames < 1024 and lines * C * frames < (1 << 20))

# ---- chain: staged against fused
taps = synth.fir_lowpass_taps(256, f32_rounded=True)
q1 = synth.biquad_rbj_lowpass()
for lines, C, K in ((8, 8, 1), (16, 8, 1), (32, 8, 1), (64, 8, 1), (64, 2, 1), (256, 2, 1), (1, 2, 128)):
    n = lines * K * F * C
    d_in = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty_like(d_in)
    kw = dict(dtype=np.float32, lines=lines, max_batch=K)
    audit(f"FIR -> biquad -> gain, {lines:3d} Lines x {C} ch x {K:4d} buffers",
          lambda: P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(q1, F, C, **kw), P.Gain(0.7071067811865476, F, C, **kw)]),
          lambda p: p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream),
          [("fused", {"PIPE_HIP_FIR_OLS_MIN_ITEMS": "1"}), ("staged", {"PIPE_HIP_FIR_OLS_MIN_ITEMS": "1000000000"})])

# ---- resampler: the row form against the tiled kernel (4 channels and more, float32)
T, up, down = 24, 160, 147
proto = synth.resampler_proto(up, down, T)
for lines, C, K in ((1, 8, 1), (1, 8, 16), (16, 8, 1), (1, 16, 4), (1, 4, 16), (16, 4, 1), (1, 6, 16), (1, 8, 64), (64, 4, 4)):
    n_in = K * F
    cap = -(-n_in * up // down) + 1
    d_in = torch.empty(lines * n_in * C, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty(lines * cap * C, dtype=torch.float32, device="cuda")
    audit(f"resampler 160/147, {lines:3d} Lines x {C:2d} ch x {K:4d} buffers",
          lambda: P.Resampler(proto, T, up, down, F, C, dtype=np.float32, lines=lines, max_batch=K),
          lambda p: p.resample_batch(d_in, n_in, d_out, cap, stream=st.cuda_stream),
          [("rows", {"PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS": "1"}), ("tiled", {"PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS": "1000000000"})])

print(f"{len(flagged)} shape(s) flagged" + (": " + "; ".join(flagged) if flagged else ""))
sys.exit(1 if flagged else 0)
