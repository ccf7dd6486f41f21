import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PIPE_HIP_FIR_OLS_MIN_ITEMS"] = "1"
import numpy as np, torch
from pipe_amd import processors as P, synth
for (lines, C, frames, N) in [(1, 6, 4000, 100), (3, 6, 16666, 100), (1, 4, 4000, 100), (1, 10, 3000, 256), (2, 8, 5000, 33)]:
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    x = np.random.default_rng(1).uniform(-1, 1, (lines, frames, C)).astype(np.float32)
    outs = {}
    for exact in (False, True):
        with P.Fir(taps, frames, C, dtype=np.float32, lines=lines, max_batch=1) as p:
            p.start()
            if exact:
                p.set_exact(True)
            d_in = torch.from_numpy(x).cuda()
            y = torch.full_like(d_in, float("nan"))
            p.process_batch(d_in, y, frames)
            torch.cuda.synchronize()
            outs[exact] = (y.cpu().numpy(), p.kernel_name())
    got, ref = outs[False][0], outs[True][0]
    bad = np.abs(got - ref) > 1e-5
    print(lines, C, frames, N, outs[False][1], "bad:", int(bad.sum()), "nan:", int(np.isnan(got).sum()))
    if bad.any():
        idx = np.argwhere(bad)
        L = 1025 - N
        print("  first bad", idx[:5].tolist(), "last bad", idx[-3:].tolist())
        fr = np.unique(idx[:, 1] // L)
        ch = np.unique(idx[:, 2])
        print("  bad tiles", fr.tolist()[:20], "bad channels", ch.tolist(), "bad lines", np.unique(idx[:, 0]).tolist())
