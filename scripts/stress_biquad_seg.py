"""Soak test of the time-segmented biquad (LDS-tile and lane-walk forms, both tile scans): random channel
counts, Line counts, section counts and pole positions, call sequences from a few frames to hundreds of tiles,
in place and out of place, staged chains around the stage (float64 in or out); every output compared with the
oracle under the bound include/pipe_hip.h states (one float32 ulp measured at max(|y|, 2^-19 kappa x the Line's
full scale)), and the state carried from call to call with it.  scripts/stress_biquad_seg.py [iterations] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES"] = "1"
from oracle import oracle as O  # noqa: E402
from pipe_amd import processors as P, synth  # noqa: E402
from test_gpu_biquad_seg import kappa  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
kinds, worst = {}, 0.0
for it in range(iters):
    C = int(rng.choice([1, 2, 4, 8, 2, 1, 3, 6, 5, 7, 9, 12]))
    lines = int(rng.choice([1, 1, 2, 5, 37, 300]))
    S = int(rng.choice([1, 2, 2, 3, 4, 5]))
    q = np.vstack([synth.biquad_rbj_lowpass(fc=float(np.exp(rng.uniform(np.log(60.0), np.log(12000.0)))),
                                            q=float(rng.uniform(0.5, 6.0))) for _ in range(S)])
    k = kappa(q)
    budget = int(rng.choice([3000, 60000, 1 << 20, 1 << 22]))   # samples per Line and call, roughly
    ncalls = int(rng.integers(1, 4))
    calls = [max(1, int(rng.integers(1, max(2, budget // (C * max(1, lines // 4)))))) for _ in range(ncalls)]
    if rng.random() < 0.3:
        calls = [max(32, c // 4096 * 4096) for c in calls]
    total = sum(calls)
    shape = str(rng.choice(["b", "b", "bg", "gb", "bgg"]))
    inplace = shape == "b" and rng.random() < 0.3
    for knob in ("PIPE_HIP_BIQUAD_NO_TILE", "PIPE_HIP_BIQUAD_NO_WAVE_SCAN", "PIPE_HIP_BIQUAD_TILE_SEG32", "PIPE_HIP_BIQUAD_TWO_PASS",
                 "PIPE_HIP_BIQUAD_NO_SPLIT", "PIPE_HIP_BIQUAD_SPLIT_COPIES"):
        os.environ.pop(knob, None)
        if rng.random() < 0.2:
            os.environ[knob] = "1"
    x = rng.uniform(-1, 1, size=(lines, total, C)).astype(np.float32)
    F = max(calls)
    kw = dict(dtype=np.float32, lines=lines, max_batch=1)
    gains = [0.7071067811865476, 1.25]
    gi = iter(gains)
    stages = [P.Biquad(q, F, C, **kw) if ch == "b" else P.Gain(next(gi), F, C, **kw) for ch in shape]
    with (P.Chain(stages) if len(stages) > 1 else stages[0]) as p:
        p.start()
        outs, pos = [], 0
        for n in calls:
            d_in = torch.from_numpy(np.ascontiguousarray(x[:, pos:pos + n, :])).cuda()
            d_out = d_in if inplace else torch.full_like(d_in, float("nan"))
            p.process_batch(d_in, d_out, n)
            torch.cuda.synchronize()
            nm = p.kernel_name().split("<")[0] + ("/seg" if "segmented" in p.kernel_name() else "") + ("/halves" if "two halves" in p.kernel_name() else "")
            kinds[nm] = kinds.get(nm, 0) + 1
            outs.append(d_out.cpu().numpy())
            pos += n
    got = np.concatenate(outs, axis=1)
    for l in range(lines) if lines <= 5 else sorted(set(int(v) for v in rng.integers(0, lines, 4))):
        w = x[l].astype(np.float64).reshape(-1)
        rb, gi = O.Biquad(q, C), iter(gains)
        for ch in shape:
            w = rb.process(w) if ch == "b" else O.gain(w, next(gi))
        want = np.asarray(w).reshape(total, C).astype(np.float32)
        floor = np.float32(2.0 ** -19 * min(k, 1024.0) * np.abs(want).max())
        ulp = np.spacing(np.maximum(np.abs(want), floor)).astype(np.float64)
        d = np.abs(got[l].astype(np.float64) - want.astype(np.float64)) / ulp
        worst = max(worst, float(d.max()))
        assert d.max() <= 1.0, (it, C, lines, S, q.tolist(), k, calls, shape, inplace, l, float(d.max()),
                                np.argwhere(d > 1.0)[:4].tolist(), {v: os.environ.get(v) for v in os.environ if "BIQUAD" in v})
    if it % 10 == 0:
        print(f"{it:4d} C {C} lines {lines} S {S} kappa {k:7.1f} calls {calls} {shape}{' in place' if inplace else ''} ok "
              f"[{time.time() - t0:.0f} s]", flush=True)
print("stress ok", kinds, "largest distance / bound", round(worst, 3))
