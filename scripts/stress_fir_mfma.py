"""Soak test of the ordered-fma FIR on the float64 matrix pipe (pipe_amd/csrc/fir_mfma.hip): random tap counts
(16 .. 4096), channel counts (odd ones too), Line counts, dtypes, call sequences (calls shorter than the
filter, ragged ends), now and then an Inf / NaN in the stream; every Line against the oracle, bit for bit.
scripts/stress_fir_mfma.py [iterations] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PIPE_HIP_FIR_MFMA_MIN_PASSES"] = "1"
from oracle import oracle as O  # noqa: E402
from pipe_amd import processors as P, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
nonfinite_cases = 0
for it in range(iters):
    ntaps = int(rng.choice([16, 17, 31, 64, 255, 256, 257, 500, 1024, 2048, 4096, int(rng.integers(16, 4097))]))
    C = int(rng.choice([1, 2, 2, 3, 4, 5, 8]))
    lines = int(rng.choice([1, 2, 3, 9]))
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    ncalls = int(rng.integers(1, 5))
    budget = 60_000 // (lines * C)
    calls = [int(rng.integers(1, max(2, budget // ncalls))) for _ in range(ncalls)]
    total = sum(calls)
    taps = synth.fir_lowpass_taps(ntaps, fc=float(rng.uniform(0.02, 0.4)))
    x = rng.uniform(-1, 1, size=(lines, total, C)).astype(dtype)
    if rng.random() < 0.25:
        nonfinite_cases += 1
        for _ in range(int(rng.integers(1, 4))):
            x[int(rng.integers(lines)), int(rng.integers(total)), int(rng.integers(C))] = rng.choice([np.inf, -np.inf, np.nan])
    with P.Fir(taps, max(calls), C, dtype=dtype, lines=lines, max_batch=1) as p:
        p.start()
        p.set_exact(True)
        d_in = torch.from_numpy(x).cuda()
        outs, pos, names = [], 0, []
        for n in calls:
            xin = d_in[:, pos:pos + n, :].contiguous()
            y = torch.full_like(xin, 12345.0)
            p.process_batch(xin, y, n)
            torch.cuda.synchronize()
            names.append(p.kernel_name())
            outs.append(y)
            pos += n
        got = torch.cat(outs, dim=1).cpu().numpy()
    assert all("fir_mfma_kernel" in n for n in names), (names, ntaps, C, lines, calls)
    for l in range(lines):
        want = O.Fir(taps, C).process(x[l].astype(np.float64)).reshape(total, C).astype(dtype)
        assert np.array_equal(got[l], want, equal_nan=True), (it, ntaps, C, lines, calls, str(dtype), l, np.argwhere(got[l] != want)[:3])
    if it % 20 == 0:
        print(f"{it:4d} taps {ntaps} C {C} lines {lines} {dtype.__name__} calls {calls} ok [{time.time() - t0:.0f} s]", flush=True)
print(f"stress ok: {iters} cases bit for bit ({nonfinite_cases} with Inf / NaN in the input)")
