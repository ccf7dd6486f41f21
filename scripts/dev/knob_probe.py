import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from pipe_amd import processors as P, synth
st = torch.cuda.Stream()
def timed(make, call, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with make() as p:
            p.start()
            for _ in range(20): call(p)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): call(p)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 100 * 1e6, p.kernel_name()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
F = 4096
# (a) resampler rows against tiled at small sizes
T, up, down = 24, 160, 147
for lines, C, K in ((1, 8, 4), (1, 8, 16), (1, 8, 32), (16, 8, 1), (32, 8, 1), (1, 4, 16), (16, 4, 1), (1, 8, 64)):
    n_in = K * F; cap = -(-n_in * up // down) + 1
    d_in = torch.empty(lines * n_in * C, dtype=torch.float32, device="cuda"); P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty(lines * cap * C, dtype=torch.float32, device="cuda")
    mk = lambda: P.Resampler(synth.resampler_proto(up, down, T), T, up, down, F, C, dtype=np.float32, lines=lines, max_batch=K)
    call = lambda p: p.resample_batch(d_in, n_in, d_out, cap, stream=st.cuda_stream)
    a = timed(mk, call, {}); b = timed(mk, call, {"PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS": "1"}); c = timed(mk, call, {"PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS": "1000000"})
    print(f"resampler {lines:3d} Lines x {C} ch x {K:3d} buffers: default {a[1][:28]:28s} {a[0]:7.1f} | rows from 1 block {b[1][:28]:28s} {b[0]:7.1f} | never rows {c[1][:28]:28s} {c[0]:7.1f}", flush=True)
# (b) 3 - 4 sections over more than 2048 series
for S in (3, 4):
    q = np.vstack([synth.biquad_rbj_lowpass(), synth.biquad_rbj_lowpass(fc=4000.0, q=1.3), synth.biquad_rbj_lowpass(fc=300.0, q=4.0), synth.biquad_rbj_lowpass(fc=2500.0, q=0.9)][:S])
    for lines, C, frames in ((512, 8, 4096), (2048, 2, 4096), (1024, 4, 4096), (300, 8, 4096), (4096, 2, 1024)):
        n = lines * frames * C
        d_in = torch.empty(n, dtype=torch.float32, device="cuda"); P.synth_fill(d_in, synth.line_seed(0)); d_out = torch.empty_like(d_in)
        mk = lambda: P.Biquad(q, frames, C, dtype=np.float32, lines=lines, max_batch=1)
        call = lambda p: p.process_batch(d_in, d_out, frames, stream=st.cuda_stream)
        a = timed(mk, call, {}); b = timed(mk, call, {"PIPE_HIP_BIQUAD_SPLIT_MAX_SERIES": "100000000"}); c = timed(mk, call, {"PIPE_HIP_BIQUAD_SPLIT_MAX_SERIES": "1"})
        print(f"biquad {S} sections {lines:5d} Lines x {C} ch x {frames} frames: default {a[1][:40]:40s} {a[0]:7.1f} | split always {b[1][:40]:40s} {b[0]:7.1f} | split never {c[1][:40]:40s} {c[0]:7.1f}", flush=True)
# (c) short Lines: tile against lane walk
q = synth.biquad_rbj_lowpass()
for lines, C, frames in ((4096, 2, 256), (4096, 2, 384), (2048, 2, 512), (2048, 2, 768), (8192, 1, 256), (1024, 8, 256), (1024, 8, 512)):
    n = lines * frames * C
    d_in = torch.empty(n, dtype=torch.float32, device="cuda"); P.synth_fill(d_in, synth.line_seed(0)); d_out = torch.empty_like(d_in)
    mk = lambda: P.Biquad(q, frames, C, dtype=np.float32, lines=lines, max_batch=1)
    call = lambda p: p.process_batch(d_in, d_out, frames, stream=st.cuda_stream)
    a = timed(mk, call, {}); b = timed(mk, call, {"PIPE_HIP_BIQUAD_TILE_MIN_FRAMES": "64"}); c = timed(mk, call, {"PIPE_HIP_BIQUAD_TILE_MIN_FRAMES": "100000"})
    print(f"biquad 1 section {lines:5d} Lines x {C} ch x {frames} frames: default {a[1][:36]:36s} {a[0]:7.1f} | tile from 64 frames {b[1][:36]:36s} {b[0]:7.1f} | never tile {c[1][:36]:36s} {c[0]:7.1f}", flush=True)
