"""Host-loop cost of many Lines: pipe.Run (one ProcessFunc launch per Line per pass) against the
stage-major RunBatched (one launch per pass for all Lines).  BASELINE configs[2] shape by default:
64 Lines x 2 ch x 4096-frame buffers, FIR-256 -> biquad -> gain chain, float64 pool buffers.
Differential timing (two stream lengths) removes handle creation and pool warm-up."""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pipe_amd import host as H, synth  # noqa: E402


def wall(mode, lines_n, buffers, buf, ch):
    taps = synth.fir_lowpass_taps(256)
    params = H.chain_params(taps, synth.biquad_rbj_lowpass(), 0.5)
    lines = [H.Line(limit=buffers * buf, channels=ch, src_kind=H.SRC_CONST, value=0.25, discard=True,
                    procs=[H.Proc(H.PROC_HIP_CHAIN, params)]) for _ in range(lines_n)]
    t0 = time.perf_counter()
    err, _ = H.run(buf, lines, mode)
    assert not err.failed, err.message
    return time.perf_counter() - t0


def main():
    lines_n, buf, ch = 64, 4096, 2
    for name, mode in (("run", H.MODE_RUN), ("run_batched", H.MODE_RUN_BATCHED)):
        wall(mode, lines_n, 2, buf, ch)  # warm
        a = min(wall(mode, lines_n, 8, buf, ch) for _ in range(2))
        b = min(wall(mode, lines_n, 40, buf, ch) for _ in range(2))
        per_pass = (b - a) / 32
        print(json.dumps({"executor": name, "lines": lines_n, "channels": ch, "buffer_size": buf,
                          "ms_per_pass": per_pass * 1e3,
                          "Msamples_per_s": lines_n * buf * ch / per_pass / 1e6}))


if __name__ == "__main__":
    main()
