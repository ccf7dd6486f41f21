"""Soak test of the resampler (pair / tiled / gather kernels): random ratios, tap counts, channel
counts, Line counts and call sequences (device-resident batches and per-buffer calls mixed), every output
frame compared bit for bit with the oracle.  scripts/stress_resampler.py [iterations] [seed]"""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from pipe_amd import processors as P, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
RATIOS = [(160, 147), (147, 160), (2, 1), (1, 2), (3, 2), (2, 3), (4, 3), (8, 7), (5, 4), (320, 147), (80, 147), (1, 1),
          (7, 5), (147, 80), (256, 255), (3, 1), (1, 3), (441, 320)]
t0 = time.time()
kinds = {}
for it in range(iters):
    up, down = RATIOS[int(rng.integers(len(RATIOS)))]
    g = math.gcd(up, down)
    up, down = up // g, down // g
    T = int(rng.choice([8, 12, 16, 24, 32, 10, 48]))
    C = int(rng.choice([2, 2, 2, 1, 4, 3]))
    lines = int(rng.choice([1, 1, 2, 5]))
    dtype = np.float32 if rng.random() < 0.7 else np.float64
    F = int(rng.choice([512, 1024, 4096]))
    K = int(rng.integers(1, 9))
    proto = synth.resampler_proto(up, down, T)
    ncalls = int(rng.integers(1, 5))
    calls = [int(rng.integers(1, K * F + 1)) for _ in range(ncalls)]
    if rng.random() < 0.3:
        calls = [c // 2 * 2 or 2 for c in calls]
    total = sum(calls)
    x = rng.uniform(-1, 1, size=(lines, total, C)).astype(dtype)
    refs = [O.Resampler(proto, T, up, down, C) for _ in range(lines)]
    tt = torch.float32 if dtype == np.float32 else torch.float64
    with P.Resampler(proto, T, up, down, F, C, dtype=dtype, lines=lines, max_batch=K) as p:
        p.start()
        pos = 0
        for n in calls:
            cap = -(-n * up // down) + 1
            xin = np.ascontiguousarray(x[:, pos:pos + n, :])
            d_in = torch.from_numpy(xin).cuda()
            d_out = torch.full((lines * cap * C,), float("nan"), dtype=tt, device="cuda")
            n_out = p.resample_batch(d_in, n, d_out, cap)
            torch.cuda.synchronize()
            kinds[p.kernel_name().split("<")[0]] = kinds.get(p.kernel_name().split("<")[0], 0) + 1
            got = d_out.cpu().numpy().reshape(lines, cap, C)[:, :n_out]
            for l in range(lines):
                want = refs[l].process(xin[l].astype(np.float64)).reshape(-1, C).astype(dtype)
                assert want.shape[0] == n_out, (it, up, down, T, C, lines, calls, n_out, want.shape)
                assert np.array_equal(got[l], want), (it, up, down, T, C, lines, str(dtype), calls, pos, l, p.kernel_name())
            pos += n
    if it % 20 == 0:
        print(f"{it:4d} {up}/{down} T {T} C {C} lines {lines} {np.dtype(dtype).name} calls {calls} ok [{time.time() - t0:.0f} s]", flush=True)
print("stress ok", kinds)
