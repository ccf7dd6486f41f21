import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PIPE_HIP_FIR_OLS_MIN_ITEMS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_chain_fused as T
T.setup_module(None)
from pipe_amd import synth
for (lines, C, frames, ntaps, q, g, calls) in [(5, 6, 30000, 64, T.TWO_SECTIONS, 0.9, [9999, 20001]),
                                               (5, 6, 30000, 64, T.LOWPASS, 0.9, [9999, 20001]),
                                               (1, 2, 30000, 64, T.TWO_SECTIONS, 0.9, [9999, 20001]),
                                               (5, 6, 30000, 64, T.TWO_SECTIONS, 0.9, [9600, 20400])]:
    taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
    x = np.random.default_rng(7).uniform(-1, 1, size=(lines, frames, C)).astype(np.float32)
    got, names = T.run_chain(taps, q, g, x, calls)
    nan = np.isnan(got)
    print(lines, C, len(q), calls, "nan count", int(nan.sum()))
    if nan.any():
        idx = np.argwhere(nan)
        print("  first", idx[0].tolist(), "lines", np.unique(idx[:, 0]).tolist(), "channels", np.unique(idx[:, 2]).tolist(),
              "first frame per line", [int(idx[idx[:, 0] == l][:, 1].min()) for l in np.unique(idx[:, 0])])
