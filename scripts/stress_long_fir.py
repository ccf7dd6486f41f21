"""Soak test of the partitioned overlap-save FIR (513 .. 4096 taps, frequency-domain delay line): random tap
counts, channel counts, Line counts and call sequences; every Line against the oracle within the
overlap-save contract (one float32 ulp at the filter's full scale, tests/test_gpu_fir_ols.py).
scripts/stress_long_fir.py [iterations] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PIPE_HIP_FIR_OLS_MIN_ITEMS"] = "1"
from oracle import oracle as O  # noqa: E402
from pipe_amd import processors as P, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
worst_all = 0.0
for it in range(iters):
    ntaps = int(rng.choice([513, 600, 1000, 1024, 1025, 1536, 2048, 3000, 4096, int(rng.integers(513, 4097))]))
    C = int(rng.choice([1, 2, 2, 4, 6]))
    lines = int(rng.choice([1, 2, 3, 7]))
    ncalls = int(rng.integers(1, 4))
    budget = 120_000 // (lines * C)
    calls = [int(rng.integers(1, max(2, budget // ncalls))) for _ in range(ncalls)]
    total = sum(calls)
    taps = synth.fir_lowpass_taps(ntaps, fc=float(rng.uniform(0.02, 0.4)), f32_rounded=True)
    x = rng.uniform(-1, 1, size=(lines, total, C)).astype(np.float32)
    with P.Fir(taps, 4096, C, dtype=np.float32, lines=lines, max_batch=64) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        outs, pos, names = [], 0, []
        for n in calls:
            xin = d_in[:, pos:pos + n, :].clone()   # (a fresh, 16-byte aligned buffer: a misaligned one takes the direct form)
            y = torch.full_like(xin, float("nan"))
            p.process_batch(xin, y, n)
            torch.cuda.synchronize()
            names.append(p.kernel_name())
            outs.append(y)
            pos += n
        got = torch.cat(outs, dim=1).cpu().numpy()
    assert not np.isnan(got).any(), (it, ntaps, C, lines, calls)
    assert all("partitioned" in n for n in names), (names, ntaps, C, lines, calls)
    floor = 2.0 ** -24 * np.abs(taps).sum()
    for l in range(lines):
        want = O.Fir(taps, C).process(x[l].astype(np.float64)).reshape(total, C)
        mag = np.maximum(np.abs(want), floor).astype(np.float32)
        d = np.abs(got[l].astype(np.float64) - want.astype(np.float32).astype(np.float64)) / np.spacing(mag).astype(np.float64)
        worst_all = max(worst_all, float(d.max()))
        assert d.max() <= 1.0, (it, ntaps, C, lines, calls, l, float(d.max()))
    if it % 10 == 0:
        print(f"{it:4d} taps {ntaps} C {C} lines {lines} calls {calls} ok [{time.time() - t0:.0f} s]", flush=True)
print("stress ok, worst", worst_all, "ulp")
