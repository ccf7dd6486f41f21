"""Soak test of the fused chain's concurrency machinery (block-local record ring and round counters,
global look-back records, tagged state slots): random shapes and call sequences, every launch of the
block-local form compared bit for bit with the global form on the same stream of calls, and spot
Lines with the oracle.  scripts/stress_fused.py [iterations] [seed]"""
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

torch.cuda.set_stream(torch.cuda.Stream())
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
filters = [synth.biquad_rbj_lowpass(), synth.biquad_rbj_lowpass(3000.0), np.array([[1.0, -1.0, 0.0, -0.9995, 0.0]]),
           # two sections (one ring of records per section)
           np.vstack([synth.biquad_rbj_lowpass(3000.0), synth.biquad_rbj_lowpass(700.0, q=2.0)]),
           np.vstack([synth.biquad_rbj_lowpass(5000.0, q=0.5), synth.biquad_rbj_lowpass(1500.0)]),
           # three and four sections (round 6: the global look-back)
           np.vstack([synth.biquad_rbj_lowpass(3000.0), synth.biquad_rbj_lowpass(700.0, q=2.0), synth.biquad_rbj_lowpass(1500.0, q=1.1)]),
           np.vstack([synth.biquad_rbj_lowpass(3000.0), synth.biquad_rbj_lowpass(700.0, q=2.0), synth.biquad_rbj_lowpass(1500.0, q=1.1),
                      synth.biquad_rbj_lowpass(5000.0, q=0.6)])]


def run(taps, q, g, x, calls, local):
    os.environ["PIPE_HIP_CHAIN_LOCAL"] = "1" if local else "0"
    lines, frames, C = x.shape
    kw = dict(dtype=np.float32, lines=lines, max_batch=1)
    outs, names = [], []
    with P.Chain([P.Fir(taps, frames, C, **kw), P.Biquad(q, frames, C, **kw), P.Gain(g, frames, C, **kw)]) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        pos = 0
        for n in calls:
            xin = d_in[:, pos:pos + n, :].contiguous()
            y = torch.full_like(xin, float("nan"))
            p.process_batch(xin, y, n)
            names.append(p.kernel_name())
            outs.append(y)
            pos += n
        p.flush()
        torch.cuda.synchronize()
    return torch.cat(outs, dim=1).cpu().numpy(), names


t0 = time.time()
for it in range(iters):
    lines = int(rng.choice([40, 256, 256, 512, 768, 2100]))
    C = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8]))   # (odd counts, round 6: the last channel alone in its pair)
    ntaps = int(rng.choice([16, 64, 256, 300]))
    q = filters[int(rng.integers(len(filters)))]
    ncalls = int(rng.integers(1, 4))
    budget = 40_000_000 // (lines * C)          # samples per Line that keep the case small
    calls = [int(rng.integers(700, max(800, min(9000, budget // ncalls)))) for _ in range(ncalls)]
    if rng.random() < 0.5:
        calls = [c // 32 * 32 for c in calls]   # buffers that end on a segment boundary: no tail kernel
    frames = sum(calls)
    taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
    x = rng.uniform(-1, 1, size=(lines, frames, C)).astype(np.float32)
    a, na = run(taps, q, 0.5, x, calls, True)
    b, nb = run(taps, q, 0.5, x, calls, False)
    assert not np.isnan(a).any() and np.array_equal(a, b), (it, lines, C, ntaps, calls, na, nb)
    l = int(rng.integers(lines))
    want = O.gain(O.Biquad(q, C).process(O.Fir(taps, C).process(x[l].astype(np.float64))), 0.5).reshape(-1, C)
    floor = 2.0 ** -24 * np.abs(want).max()
    ulp = np.spacing(np.maximum(np.abs(want), floor).astype(np.float32)).astype(np.float64)
    d = np.abs(a[l].astype(np.float64) - want.astype(np.float32).astype(np.float64)) / ulp
    assert d.max() <= 1.0, (it, lines, C, ntaps, calls, float(d.max()))
    print(f"{it:3d} lines {lines:4d} C {C} taps {ntaps:3d} calls {calls} {na[0][19:]:40s} ok  [{time.time() - t0:.0f} s]", flush=True)
print("stress ok")
