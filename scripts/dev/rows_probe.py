"""Debug probe of the resampler's row form: one resident call of K pipe buffers per process, checked against the oracle."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if len(sys.argv) > 1:
    import numpy as np
    import torch
    from oracle import oracle as O
    from pipe_amd import processors as P, synth
    K, calls = int(sys.argv[1]), int(sys.argv[2])
    up, down, T, C, F = 160, 147, 24, 2, 4096
    proto = synth.resampler_proto(up, down, T)
    n = K * F
    x = synth.samples(synth.line_seed(3), 0, n * C).reshape(n, C).astype(np.float32)
    cap = -(-n * up // down) + 1
    d_in = torch.from_numpy(x).cuda()
    ref = O.Resampler(proto, T, up, down, C)
    with P.Resampler(proto, T, up, down, F, C, dtype=np.float32, max_batch=K) as p:
        p.start()
        for c in range(calls):
            d_out = torch.full(((cap + 4096) * C,), float("nan"), dtype=torch.float32, device="cuda")
            n_out = p.resample_batch(d_in, n, d_out, cap)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy().reshape(-1, C)
            want = ref.process(x.astype(np.float64)).reshape(-1, C).astype(np.float32)
            bad = np.flatnonzero((got[:n_out] != want).any(axis=1))
            tail = np.isnan(got[n_out:]).all()
            print(f"K={K} call {c}: {p.kernel_name()} n_out {n_out} mismatching frames {bad.size} first {bad[:5]} tail untouched {tail}", flush=True)
    sys.exit(0)

for K, calls in ((8, 2), (40, 2), (160, 2), (320, 2), (512, 2), (1024, 3)):
    r = subprocess.run([sys.executable, __file__, str(K), str(calls)], capture_output=True, text=True,
                       env=dict(os.environ, PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS="1"))
    print(r.stdout.strip())
    if r.returncode:
        print(f"K={K}: rc {r.returncode}: {r.stderr.strip()[-400:]}")
