#!/usr/bin/env python3
"""The shader clock under small device-resident FIR calls against the clock under the headline's load: is the 15 us
floor of a small overlap-save launch (profiles/r06_fir_small_calls.txt) a floor at 2.4 GHz or at what a lightly
loaded chip runs at?  Samples hwmon (bench.PowerLog) over 2 s windows.  scripts/small_call_clock_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pipe_amd import processors as P, synth  # noqa: E402

F, C, N = 4096, 2, 256
taps = synth.fir_lowpass_taps(N, f32_rounded=True)
st = torch.cuda.Stream()
card = bench.sysfs_card_of(torch, 0)
os.environ["PIPE_HIP_FIR_OLS_MIN_ITEMS"] = "1"
for lines, K, gap_us in ((1, 64, 0), (1, 64, 50), (1, 256, 0), (1, 2048, 0), (1, 32768, 0)):
    n = lines * K * F * C
    d_in = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty_like(d_in)
    with P.Fir(taps, F, C, dtype=np.float32, lines=lines, max_batch=K) as p:
        p.start()
        for _ in range(20):
            p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
        torch.cuda.synchronize()
        p.set_profiling(True)
        p.kernel_time(reset=True)
        with bench.PowerLog(card=card) as plog:
            t0 = time.perf_counter()
            calls = 0
            while time.perf_counter() - t0 < 2.0:
                for _ in range(20):
                    p.process_batch(d_in, d_out, K * F, stream=st.cuda_stream)
                    if gap_us:
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        while time.perf_counter() - t1 < gap_us * 1e-6:
                            pass
                torch.cuda.synchronize()
                calls += 20
        kms, kn = p.kernel_time(reset=True)
        pw = plog.summary() or {}
        print(f"{lines} Line x {K:6d} buffers, {gap_us:3d} us between calls: {p.kernel_name():30s} kernel {kms / max(kn, 1) * 1e3:9.1f} us"
              f"   sclk {pw.get('sclk_mhz')} MHz  power {pw.get('power_w')} W  ({calls} calls)", flush=True)
