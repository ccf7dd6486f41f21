import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PIPE_HIP_FIR_OLS_MIN_ITEMS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_chain_fused as T
T.setup_module(None)
from pipe_amd import synth
for (lines, C, frames, ntaps, q, g, calls) in [(1, 2, 700, 64, T.TWO_SECTIONS, None, [700]),
                                               (1, 2, 2000, 64, T.TWO_SECTIONS, None, [2000]),
                                               (1, 4, 700, 64, T.TWO_SECTIONS, None, [700])]:
    taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
    x = np.random.default_rng(7).uniform(-1, 1, size=(lines, frames, C)).astype(np.float32)
    got, names = T.run_chain(taps, q, g, x, calls)
    want = T.oracle_chain(taps, q, g, x[0])
    nan = np.isnan(got)
    bad = ~nan & (np.abs(got[0] - want) > 1e-4)
    print(lines, C, frames, names[0][-30:], "nan", int(nan.sum()), "bad", int(bad.sum()))
    if nan.any():
        idx = np.argwhere(nan)
        fr = np.unique(idx[:, 1])
        print("  nan frames: first", fr[:8].tolist(), "count", len(fr), "channels", np.unique(idx[:, 2]).tolist())
        print("  got[0,:4]", got[0, :4].tolist(), "want", want[:4].tolist())
