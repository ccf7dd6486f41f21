#!/usr/bin/env python3
"""does the sixth, two-thirds-empty transform of a 4096-frame buffer (4096 = 5 x 768 + 256 with
256 taps) cost the fused chain what its share of the arithmetic says?  The A/B needs no new kernel: the same chain at
buffer sizes that need 5, 6 (one of them 1/3 full) and 6 full tiles per Line and launch, 512 Lines x 8 channels,
rotating through sets of > 640 MiB (every launch streams from HBM).  If the padded tile cost its share, samples per
second at 3840 frames (5 full tiles) would be 6 / 5.33 = 1.125 x those at 4096; what the launch really is: 3 units
per wave (6144 units on 2048 waves), and a wave's three units take the time of its slowest.
    PYTHONPATH=. python scripts/chain_tail_ab.py > profiles/r05_chain_tail_ab.txt"""
import sys

import numpy as np
import torch

from pipe_amd import processors as P
from pipe_amd import synth

L, C, N = 512, 8, 256
taps = synth.fir_lowpass_taps(N, f32_rounded=True)
q = synth.biquad_rbj_lowpass()
st = torch.cuda.Stream()
src = torch.empty(320 << 20, dtype=torch.float32, device="cuda")
P.synth_fill(src, synth.line_seed(3))
arena = torch.empty(320 << 20, dtype=torch.float32, device="cuda")
rows = []
for F, what in ((3840, "5 full tiles"), (4096, "5 full tiles + one 1/3 full (configs[3])"), (4608, "6 full tiles"),
                (3072, "4 full tiles"), (7680, "10 full tiles"), (8192, "10 full + one 2/3 full"), (2304, "3 full tiles")):
    n = L * F * C
    kw = dict(dtype=np.float32, lines=L, max_batch=1)
    sets = max(2, min(24, -(-(640 << 20) // (2 * n * 4))))
    a = -(-n // 64) * 64
    sets = min(sets, src.numel() // a)
    with P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(q, F, C, **kw), P.Gain(0.7071067811865476, F, C, **kw)]) as p:
        p.start()
        calls = [(src[k * a:k * a + n], arena[k * a:k * a + n]) for k in range(sets)]
        for i in range(600):
            x, y = calls[i % sets]
            p.process_batch(x, y, F, stream=st.cuda_stream)
        torch.cuda.synchronize()
        p.set_profiling(True)
        p.kernel_time(reset=True)
        for i in range(400):
            x, y = calls[i % sets]
            p.process_batch(x, y, F, stream=st.cuda_stream)
        torch.cuda.synchronize()
        ms, k = p.kernel_time(reset=True)
        ms /= max(k, 1)
        tiles = -(-F // 768)
        units = L * (C // 2) * tiles // 2
        rows.append((F, what, ms, n / (ms * 1e-3) / 1e9, n * 8 / (ms * 1e-3) / 8e12, tiles, units, units / 2048, p.kernel_name()))
        p.flush()
print("fused chain, 512 Lines x 8 ch x F frames per launch, streaming; kernel:", rows[0][-1])
print(f"{'F':>6} {'tiles':>5} {'units':>6} {'units/wave':>10} {'kernel ms':>10} {'Gsamples/s':>11} {'of HBM':>7}  what")
for F, what, ms, gs, fr, tiles, units, upw, _ in rows:
    print(f"{F:>6} {tiles:>5} {units:>6} {upw:>10.2f} {ms:>10.5f} {gs:>11.1f} {fr:>7.4f}  {what}")
base = {r[0]: r for r in rows}
print(f"3840 vs 4096 frames: {base[3840][3] / base[4096][3]:.3f} x the samples per second (the arithmetic's share says 1.125); "
      f"kernel ms {base[3840][2]:.5f} vs {base[4096][2]:.5f}: the sixth tile costs {(base[4096][2] - base[3840][2]) * 1e3:.1f} us of {base[4096][2] * 1e3:.1f}")
print(f"4608 (six FULL tiles) vs 4096: kernel ms {base[4608][2]:.5f} vs {base[4096][2]:.5f}: filling the sixth tile costs {(base[4608][2] - base[4096][2]) * 1e3:.1f} us more")
