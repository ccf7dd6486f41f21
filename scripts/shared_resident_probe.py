#!/usr/bin/env python3
"""Per-call latency of N per-buffer FIR handles called round-robin (the synchronous host loop's order, run.go:112-132):
plain path, the exclusive doorbell (one handle holds it, the others stay plain) and PIPE_HIP_PARAM_RESIDENT_SHARED (all
of them in the device's one doorbell queue).  scripts/shared_resident_probe.py [calls per handle]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pipe_amd import processors as P, synth  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
F, C = 4096, 2
taps = synth.fir_lowpass_taps(256)
x = synth.samples(synth.line_seed(1), 0, F * C).reshape(F, C)
for dtype in (np.float32, np.float64):
    xin = x.astype(dtype)
    for N in (1, 2, 4, 8, 16):
        row = {}
        for mode in ("plain", "exclusive", "shared"):
            hs = [P.Fir(taps, F, C, dtype=dtype) for _ in range(N)]
            for h in hs:
                h.start()
                if mode == "exclusive":
                    h.set_resident(True)
                elif mode == "shared":
                    assert h.set_resident_shared(True)
            for _ in range(20):
                for h in hs:
                    h.process(xin)
            t = []
            for _ in range(calls):
                for h in hs:
                    t0 = time.perf_counter()
                    h.process(xin)
                    t.append(time.perf_counter() - t0)
            t.sort()
            row[mode] = (t[len(t) // 2] * 1e6, t[len(t) * 9 // 10] * 1e6)
            dropped = sum(h.resident_info()[2] for h in hs)
            for h in hs:
                h.close()
        print(f"{np.dtype(dtype).name} FIR-256 4096x2, {N:2d} handles round-robin: per call median (p90) us  plain {row['plain'][0]:5.1f} ({row['plain'][1]:5.1f})"
              f"  exclusive doorbell {row['exclusive'][0]:5.1f} ({row['exclusive'][1]:5.1f})  shared queue {row['shared'][0]:5.1f} ({row['shared'][1]:5.1f})  [dropped in the shared run: {dropped}]", flush=True)
