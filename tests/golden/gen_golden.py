#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (run in the build container).

The reference ships no FIR / biquad / resampler / gain / mix, so
these vectors are NOT reference outputs: they freeze the oracle's float64
arithmetic contract (oracle/dsp_oracle.h) after it has been cross-checked against
scipy here.  The pipe-loop vectors (counts) ARE the reference's own known answers
and are listed with their pipe_test.go / mock_test.go line numbers.

    python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np
import scipy.signal as ss

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O  # noqa: E402
from pipe_amd import synth  # noqa: E402

EPS = np.finfo(np.float64).eps


def stream(seed, frames, channels):
    return synth.samples(synth.line_seed(seed), 0, frames * channels).reshape(frames, channels)


def main():
    # FIR-256, 2 ch: buffers of 600, 200 (< history) and 224 frames
    C, N = 2, 256
    h = synth.fir_lowpass_taps(N)
    cuts = [0, 600, 800, 1024]
    x = stream(40, cuts[-1], C)
    f = O.Fir(h, C)
    y = np.concatenate([f.process(x[a:b]).reshape(-1, C) for a, b in zip(cuts[:-1], cuts[1:])])
    ref = ss.lfilter(h, [1.0], x, axis=0)
    assert np.max(np.abs(y - ref)) <= 4 * N * EPS * np.abs(h).sum()
    np.savez(os.path.join(HERE, "fir256_2ch.npz"), seed=40, cuts=cuts, taps=h, y_f64=y,
             y_f32=y.astype(np.float32))

    # biquad cascade (2 sections), 8 ch
    C = 8
    q = np.vstack([synth.biquad_rbj_lowpass(), synth.biquad_rbj_lowpass(fc=4000.0, q=1.3)])
    cuts = [0, 300, 301, 512]
    x = stream(41, cuts[-1], C)
    b = O.Biquad(q, C)
    y = np.concatenate([b.process(x[a:b_]).reshape(-1, C) for a, b_ in zip(cuts[:-1], cuts[1:])])
    sos = np.array([[c[0], c[1], c[2], 1.0, c[3], c[4]] for c in q])
    assert np.max(np.abs(y - ss.sosfilt(sos, x, axis=0))) <= 1e-12
    np.savez(os.path.join(HERE, "biquad2_8ch.npz"), seed=41, cuts=cuts, coeffs=q, y_f64=y,
             y_f32=y.astype(np.float32))

    # resampler 160/147, 24 taps/phase, 2 ch
    C, T, up, down = 2, 24, 160, 147
    proto = synth.resampler_proto(up, down, T)
    cuts = [0, 400, 403, 800]
    x = stream(42, cuts[-1], C)
    r = O.Resampler(proto, T, up, down, C)
    parts = [r.process(x[a:b_]).reshape(-1, C) for a, b_ in zip(cuts[:-1], cuts[1:])]
    y = np.concatenate(parts)
    ref = ss.upfirdn(proto, x, up=up, down=down, axis=0)[: y.shape[0]]
    assert np.max(np.abs(y - ref)) <= 64 * T * EPS * np.abs(proto).max() * up
    np.savez(os.path.join(HERE, "resample_160_147_2ch.npz"), seed=42, cuts=cuts, proto=proto,
             taps_per_phase=T, up=up, down=down, out_lens=[p.shape[0] for p in parts], y_f64=y,
             y_f32=y.astype(np.float32))

    # fused chain FIR-64 -> biquad -> gain(1/sqrt2), 2 ch, + 2-input mix
    C = 2
    h64 = synth.fir_lowpass_taps(64)
    q1 = synth.biquad_rbj_lowpass()
    g = 0.7071067811865476
    x = stream(43, 512, C)
    y = O.gain(O.Biquad(q1, C).process(O.Fir(h64, C).process(x)), g).reshape(-1, C)
    x2 = stream(44, 512, C)
    m = O.mix([x, x2])
    np.savez(os.path.join(HERE, "chain_mix_2ch.npz"), seed=43, seed2=44, taps=h64, coeffs=q1, gain=g,
             y_f64=y, y_f32=y.astype(np.float32), mix_f64=m, mix_f32=(x.astype(np.float32).astype(np.float64)
                                                                    + x2.astype(np.float32).astype(np.float64)).astype(np.float32))

    # the reference's own known answers for the buffer loop
    np.savez(os.path.join(HERE, "pipe_loop_known_answers.npz"),
             buffer_size=512,
             # pipe_test.go:337,363,394,399,404  (limit frames -> messages)
             limits=[1040, 1640, 3048, 4096], messages=[3, 4, 6, 8],
             # pipe_test.go:84-105
             simple_pipe=[862, 862 * 512, 2],
             # mock_test.go:69-92 (buffer 5, C=2): limit -> calls
             source_limits=[11, 2500], source_calls=[3, 500])
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
