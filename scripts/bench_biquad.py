"""biquad alone on the BASELINE config-3 shape (512 Lines x 8 ch x 4096 frames)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pipe_amd import processors as P, synth  # noqa: E402

F, C, L = 4096, 8, 512
q = synth.biquad_rbj_lowpass()
n = L * F * C
for tin, tout in ((torch.float32, torch.float32), (torch.float64, torch.float32)):
    d_in = torch.empty(n, dtype=tin, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty(n, dtype=tout, device="cuda")
    with P.Biquad(q, F, C, dtype=np.float32, lines=L, max_batch=1) as p:
        p.start()
        run = (lambda: p.process_batch(d_in, d_out, F)) if tin == torch.float32 else None
        if run is None:
            continue
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(json.dumps({"io": "f32->f32", "kernel": p.kernel_name(), "ms": dt * 1e3, "gsamples_s": n / dt / 1e9}))
