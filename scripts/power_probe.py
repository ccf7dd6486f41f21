#!/usr/bin/env python3
"""One kernel in a loop for a few seconds (for scripts/clock_log.py: shader clock and socket power at
steady state), and its kernel time from events on the dispatch.
  python scripts/power_probe.py KIND [SECONDS]
KIND: ols (headline overlap-save FIR) | direct (bit-exact direct-form FIR) | gain | chain (fused configs[3])
      | resampler | idle (no launches: the socket's idle power)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pipe_amd import processors as P  # noqa: E402
from pipe_amd import synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "ols"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
F, N = 4096, 256
taps = synth.fir_lowpass_taps(N, f32_rounded=True)
st = torch.cuda.Stream()
s = st.cuda_stream
if kind == "idle":
    torch.cuda.synchronize()
    time.sleep(seconds)
    print(json.dumps({"kind": kind}))
    sys.exit(0)
if kind in ("ols", "direct", "gain"):
    C, K = 2, 32768  # 268 M scalar samples per launch (2.1 GB in + out)
    n = K * F * C
    d_in = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty_like(d_in)
    p = P.Gain(0.5, F, C, dtype=np.float32, max_batch=K) if kind == "gain" else P.Fir(taps, F, C, dtype=np.float32, max_batch=K)
    if kind == "direct":
        p.set_exact(True)
    call = lambda: p.process_batch(d_in, d_out, K * F, stream=s)  # noqa: E731
    flops = {"ols": 58.6, "direct": 2.0 * N, "gain": 1.0}[kind] * n
elif kind == "chain":
    L, C = 512, 8
    n = L * F * C
    d_in = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty_like(d_in)
    kw = dict(dtype=np.float32, lines=L, max_batch=1)
    p = P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(synth.biquad_rbj_lowpass(), F, C, **kw), P.Gain(0.7071067811865476, F, C, **kw)])
    call = lambda: p.process_batch(d_in, d_out, F, stream=s)  # noqa: E731
    flops = 85.0 * n
elif kind == "resampler":
    T, up, down, C, K = 24, 160, 147, 2, 1024
    n_in = K * F
    cap = -(-n_in * up // down) + 1
    d_in = torch.empty(n_in * C, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty(cap * C, dtype=torch.float32, device="cuda")
    p = P.Resampler(synth.resampler_proto(up, down, T), T, up, down, F, C, dtype=np.float32, max_batch=K)
    call = lambda: p.resample_batch(d_in, n_in, d_out, cap, stream=s)  # noqa: E731
    n = cap * C
    flops = 2.0 * 24.5 * n
else:
    sys.exit("unknown kind")
p.start()
for _ in range(5):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
launches = 0
p.set_profiling(True)
p.kernel_time(reset=True)
ms_total, k_total = 0.0, 0
while time.perf_counter() - t0 < seconds:
    for _ in range(50):
        call()
    launches += 50
    if launches % 500 == 0:
        ms, k = p.kernel_time(reset=True)
        ms_total += ms
        k_total += k
torch.cuda.synchronize()
ms, k = p.kernel_time(reset=True)
ms_total += ms
k_total += k
wall = time.perf_counter() - t0
avg = ms_total / max(k_total, 1)
print(json.dumps({"kind": kind, "lib": os.environ.get("PIPE_HIP_LIB", "shipped"), "kernel": p.kernel_name(), "launches": launches,
                  "avg_kernel_ms": round(avg, 5), "busy_fraction": round(ms_total * 1e-3 / wall, 3),
                  "scalar_samples_per_launch": n, "gsamples_per_s": round(n / (avg * 1e-3) / 1e9, 2),
                  "f64_tflops_as_issued": round(flops / (avg * 1e-3) / 1e12, 2)}))
