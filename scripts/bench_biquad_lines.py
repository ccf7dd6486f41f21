"""Exact (or relaxed) biquad over many Lines, one 4096-frame call: which kernel form runs and how long it takes.
scripts/bench_biquad_lines.py LINES CHANNELS [SECTIONS] [relaxed]"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from pipe_amd import processors as P, synth
L, F, C = int(sys.argv[1]), 4096, int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1
exact = not (len(sys.argv) > 4 and sys.argv[4] == "relaxed")
q = np.vstack([synth.biquad_rbj_lowpass(500.0 * (j + 1)) for j in range(S)])
st = torch.cuda.Stream()
with P.Biquad(q, F, C, dtype=np.float32, lines=L, max_batch=1) as p:
    p.start(); p.set_exact(exact)
    x = torch.rand(L * F * C, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
    for _ in range(5): p.process_batch(x, y, F, stream=st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): p.process_batch(x, y, F, stream=st.cuda_stream)
    torch.cuda.synchronize()
    print(json.dumps({"lines": L, "C": C, "S": S, "exact": exact, "lds_env": os.environ.get("PIPE_HIP_BIQUAD_LDS"), "kernel": p.kernel_name(), "us": round((time.perf_counter() - t0) / 50 * 1e6, 1)}))
