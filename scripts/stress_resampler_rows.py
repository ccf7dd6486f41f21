"""Soak test of the resampler's row form (resampler_rows.hip): random ratios, tap counts, channel counts, Line counts and
sequences of device-resident calls long enough for it (the threshold knob lowered to one block), every output frame
compared bit for bit with the oracle, nothing written past a call's outputs.
    scripts/stress_resampler_rows.py [iterations] [seed]"""
import math
import os
import sys
import time

os.environ["PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS"] = "1"

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from pipe_amd import processors as P, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
RATIOS = [(160, 147), (147, 160), (2, 1), (1, 2), (3, 2), (2, 3), (4, 3), (8, 7), (5, 4), (80, 147), (1, 1), (7, 5),
          (147, 80), (256, 255), (3, 1), (161, 147), (441, 320), (40, 147), (8, 1), (16, 1), (160, 3)]
t0 = time.time()
kinds = {}
for it in range(iters):
    up, down = RATIOS[int(rng.integers(len(RATIOS)))]
    g = math.gcd(up, down)
    up, down = up // g, down // g
    T = int(rng.choice([8, 12, 16, 24, 24, 32]))
    C = int(rng.choice([2, 2, 4, 6, 8, 10, 16]))
    lines = int(rng.choice([1, 1, 2, 3]))
    big = max(up, down)
    row_in = down * max(1 if big >= 144 else 144 // big, -(-(T - 1) // down))
    rpb = 64 // (C // 2)
    F = 4096
    ncalls = int(rng.integers(1, 4))
    # calls of 0.8 .. 4 blocks of rows, not aligned to anything; now and then a short one in between
    calls = []
    for _ in range(ncalls):
        calls.append(int(rng.uniform(0.8, 4.0) * rpb * row_in) + int(rng.integers(0, row_in)))
        if rng.random() < 0.3:
            calls.append(int(rng.integers(1, row_in)))
    total = sum(calls)
    x = rng.uniform(-1, 1, size=(lines, total, C)).astype(np.float32)
    refs = [O.Resampler(synth.resampler_proto(up, down, T), T, up, down, C) for _ in range(lines)]
    proto = synth.resampler_proto(up, down, T)
    with P.Resampler(proto, T, up, down, F, C, dtype=np.float32, lines=lines, max_batch=max(calls) // F + 1) as p:
        p.start()
        pos = 0
        for n in calls:
            cap = -(-n * up // down) + 1
            xin = np.ascontiguousarray(x[:, pos:pos + n, :])
            d_in = torch.from_numpy(xin).cuda()
            d_out = torch.full((lines * cap * C,), float("nan"), dtype=torch.float32, device="cuda")
            n_out = p.resample_batch(d_in, n, d_out, cap)
            torch.cuda.synchronize()
            name = p.kernel_name().split("<")[0]
            kinds[name] = kinds.get(name, 0) + 1
            got = d_out.cpu().numpy().reshape(lines, cap, C)
            for l in range(lines):
                want = refs[l].process(xin[l].astype(np.float64)).reshape(-1, C).astype(np.float32)
                assert want.shape[0] == n_out, (it, up, down, T, C, lines, n)
                assert np.array_equal(got[l, :n_out], want), (it, up, down, T, C, lines, n, name)
                assert np.isnan(got[l, n_out:]).all(), (it, up, down, T, C, lines, n, name)
            pos += n
print(f"stress_resampler_rows: {iters} streams, calls by kernel {kinds}, all bit-exact, {time.time() - t0:.1f} s")
