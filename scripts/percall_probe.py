#!/usr/bin/env python3
"""Per-buffer (ProcessFunc form) latency probe: one 4096x2 buffer per pipe_hip_process call.

Prints, per io dtype and per register blocking R of the direct kernel (PIPE_HIP_FIR_R; "auto" = the
library's choice): synchronous us per call, us per buffer with two buffers in flight, and the kernel's
own duration (HIP events around the launch).  GPU box only.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pipe_amd import processors as P  # noqa: E402
from pipe_amd import synth  # noqa: E402


def timed(fn, n, warm):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


def main():
    F, C, N = int(os.environ.get("PROBE_F", "4096")), int(os.environ.get("PROBE_C", "2")), 256
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    x = synth.samples(synth.line_seed(0), 0, F * C, np.float32).reshape(F, C)
    rs = sys.argv[1:] or ["auto"]
    for r in rs:
        # "auto" | R | "gain" (the launch + PCIe floor) | "N<taps>" (the tap loop's share) | "biquad" | "biquad<sections>" | "chain"
        os.environ.pop("PIPE_HIP_FIR_R", None)
        make = lambda dt: P.Fir(taps, F, C, dtype=dt)
        if r == "gain":
            make = lambda dt: P.Gain(0.5, F, C, dtype=dt)
        elif r == "biquad":
            make = lambda dt: P.Biquad(synth.biquad_rbj_lowpass(), F, C, dtype=dt)
        elif r == "resampler":  # 44.1 -> 48 kHz, as many frames in as fill the buffer going out
            proto = synth.resampler_proto(160, 147, 24)
            make = lambda dt: P.Resampler(proto, 24, 160, 147, F, C, dtype=dt)
        elif r.startswith("biquad"):  # biquad<sections>
            qs = np.stack([synth.biquad_rbj_lowpass(500.0 * (j + 1)) for j in range(int(r[6:]))])
            make = lambda dt: P.Biquad(qs, F, C, dtype=dt)
        elif r == "chain":
            make = lambda dt: P.Chain([P.Fir(taps, F, C, dtype=dt), P.Biquad(synth.biquad_rbj_lowpass(), F, C, dtype=dt),
                                       P.Gain(0.5, F, C, dtype=dt)])
        elif r.startswith("N"):
            tn = synth.fir_lowpass_taps(int(r[1:]), f32_rounded=True)
            make = lambda dt: P.Fir(tn, F, C, dtype=dt)
        elif r != "auto":
            os.environ["PIPE_HIP_FIR_R"] = r
        for dtype in (np.float32, np.float64):
            with make(dtype) as p:
                p.start()
                xin = x.astype(dtype)
                if r == "resampler":
                    xin = xin[:F * 147 // 160 - 1]
                best = None
                for _ in range(3):
                    dt = timed(lambda: p.process(xin), 400, 50)
                    best = dt if best is None else min(best, dt)
                p.set_profiling(True)
                p.kernel_time(reset=True)
                for _ in range(200):
                    p.process(xin)
                ms, n = p.kernel_time(reset=True)
                p.set_profiling(False)
                p.submit(xin)

                def pipelined():
                    p.submit(xin)
                    p.collect()
                best2 = None
                for _ in range(3):
                    dt2 = timed(pipelined, 400, 50)
                    best2 = dt2 if best2 is None else min(best2, dt2)
                p.collect()
                print(json.dumps({"R": r, "io": str(np.dtype(dtype)), "sync_us": round(best * 1e6, 2),
                                  "two_in_flight_us": round(best2 * 1e6, 2),
                                  "kernel_us": round(ms / max(n, 1) * 1e3, 2), "launches": n}), flush=True)


if __name__ == "__main__":
    main()
