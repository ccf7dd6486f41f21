#!/usr/bin/env python3
"""Short Lines through the time-segmented biquad (float32, >= 2^20 samples a call): the LDS-tile form against the lane
walk by frames a Line and channels (PIPE_HIP_BIQUAD_TILE_MIN_FRAMES).  scripts/dev/tile_min_frames_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pipe_amd import processors as P, synth
st = torch.cuda.Stream()
q = synth.biquad_rbj_lowpass()
def timed(mk, call, env):
    old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
    try:
        with mk() as p:
            p.start()
            for _ in range(20): call(p)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): call(p)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 100 * 1e6, p.kernel_name()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
for C in (1, 2, 3, 4, 8):
    for frames in (128, 192, 256, 320, 384, 448, 512):
        for total in (1 << 21, 1 << 23):
            lines = max(1, total // (frames * C))
            n = lines * frames * C
            d_in = torch.empty(n, dtype=torch.float32, device="cuda"); P.synth_fill(d_in, synth.line_seed(0)); d_out = torch.empty_like(d_in)
            mk = lambda: P.Biquad(q, frames, C, dtype=np.float32, lines=lines, max_batch=1)
            call = lambda p: p.process_batch(d_in, d_out, frames, stream=st.cuda_stream)
            a = timed(mk, call, {}); b = timed(mk, call, {"PIPE_HIP_BIQUAD_TILE_MIN_FRAMES": "64"}); c = timed(mk, call, {"PIPE_HIP_BIQUAD_TILE_MIN_FRAMES": "100000"})
            print(f"{C} ch x {frames:3d} frames x {lines:6d} Lines: default {a[1][:28]:28s} {a[0]:7.1f} | tile {b[1][:20]:20s} {b[0]:7.1f} | walk {c[1][:20]:20s} {c[0]:7.1f}  tile/walk {b[0]/c[0]:.2f}", flush=True)
