"""Row form against the wave kernel over sizes: kernel time of one resident launch (PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS=1 / -1)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if len(sys.argv) > 1:
    import numpy as np
    import torch
    from pipe_amd import processors as P, synth
    torch.cuda.set_stream(torch.cuda.Stream())
    for up, down, C, lines, K in ((160, 147, 2, 1, 256), (160, 147, 2, 1, 1024), (160, 147, 2, 1, 4096), (160, 147, 2, 4, 1024), (160, 147, 2, 16, 1024),
                                  (160, 147, 2, 64, 64), (160, 147, 2, 64, 8), (160, 147, 4, 1, 1024), (160, 147, 8, 1, 256), (160, 147, 8, 8, 256), (147, 160, 2, 8, 1024), (2, 1, 2, 8, 1024)):
        F, T = 4096, 24
        proto = synth.resampler_proto(up, down, T)
        n_in = K * F
        d_in = torch.empty(lines * n_in * C, dtype=torch.float32, device="cuda")
        P.synth_fill(d_in, synth.line_seed(0))
        cap = -(-n_in * up // down) + 1
        d_out = torch.empty(lines * cap * C, dtype=torch.float32, device="cuda")
        with P.Resampler(proto, T, up, down, F, C, dtype=np.float32, lines=lines, max_batch=K) as p:
            p.start()
            p.set_profiling(True)
            for _ in range(3):
                n = p.resample_batch(d_in, n_in, d_out, cap)
            p.kernel_time()
            for _ in range(10):
                n = p.resample_batch(d_in, n_in, d_out, cap)
            torch.cuda.synchronize()
            ms, cnt = p.kernel_time()
            ms /= max(cnt, 1)
            gb = lines * (n_in + n) * C * 4 / 1e9
            print(f"{up}/{down} C={C} lines={lines:3d} K={K:5d} {p.kernel_name():32s} {ms*1e3:9.1f} us  frac {gb/ms*1e3/8000:.3f}", flush=True)
    sys.exit(0)

for v in ("1", "-1"):
    print(f"== PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS={v}")
    r = subprocess.run([sys.executable, __file__, "x"], capture_output=True, text=True, env=dict(os.environ, PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS=v))
    print(r.stdout.strip())
    if r.returncode:
        print(r.stderr[-500:])
