"""An independent statement of the arithmetic contract of oracle/dsp_oracle.h in exact rational
arithmetic (fractions.Fraction): every operation is computed exactly and rounded ONCE to binary64
(float(Fraction) is correctly rounded, round-half-even) -- which is what IEEE-754 fma / mul / add do.
The C oracle must agree bit for bit.  This does not pin the DSP oracle to the reference (the reference
has no DSP stages: SURVEY.md F1/F2, "parity unpinned"); it pins the C restatement to its own written
contract through a second, independent implementation, so that a slip in the C code (an unfused
multiply-add, a reordered sum, a wrong tap index) cannot hide behind "the oracle is the specification".
"""
from fractions import Fraction as Fr

import numpy as np

from oracle import oracle as O
from pipe_amd import synth


def fma(a, b, c) -> float:
    return float(Fr(a) * Fr(b) + Fr(c))


def mul(a, b) -> float:
    return float(Fr(a) * Fr(b))


def add(a, b) -> float:
    return float(Fr(a) + Fr(b))


def sig(seed, frames, channels):
    return synth.samples(synth.line_seed(seed), 0, frames * channels).reshape(frames, channels).astype(np.float64)


def test_fir_is_the_ordered_fma_chain_with_history_across_calls():
    C, N = 3, 17
    h = synth.fir_lowpass_taps(N)
    x = sig(1, 90, C)
    want = np.empty_like(x)
    for c in range(C):
        for n in range(x.shape[0]):
            acc = 0.0
            for k in range(N):
                xv = x[n - k, c] if n - k >= 0 else 0.0
                acc = fma(h[k], xv, acc)
            want[n, c] = acc
    f = O.Fir(h, C)
    got = np.concatenate([f.process(x[:37]), f.process(x[37:38]), f.process(x[38:])]).reshape(-1, C)
    assert np.array_equal(got, want)


def test_biquad_is_df2t_with_the_written_operation_order():
    C = 2
    q = np.vstack([synth.biquad_rbj_lowpass(), synth.biquad_rbj_lowpass(3000.0, q=2.0)])
    x = sig(2, 120, C)
    want = np.empty_like(x)
    for c in range(C):
        st = [[0.0, 0.0] for _ in q]
        for n in range(x.shape[0]):
            v = x[n, c]
            for s, (b0, b1, b2, a1, a2) in enumerate(q):
                y = fma(b0, v, st[s][0])
                t = fma(b1, v, st[s][1])
                st[s][0] = fma(-a1, y, t)
                st[s][1] = fma(-a2, y, mul(b2, v))
                v = y
            want[n, c] = v
    b = O.Biquad(q, C)
    got = np.concatenate([b.process(x[:50]), b.process(x[50:])]).reshape(-1, C)
    assert np.array_equal(got, want)


def test_gain_and_mix_round_once_per_operation():
    x = sig(3, 64, 2)
    g = 0.7071067811865476
    assert np.array_equal(O.gain(x, g).reshape(x.shape), np.vectorize(lambda v: mul(v, g))(x))
    a, b, c = sig(4, 40, 2).ravel(), sig(5, 40, 2).ravel(), sig(6, 40, 2).ravel()
    want = np.array([add(add(p, q_), r) for p, q_, r in zip(a, b, c)])
    assert np.array_equal(O.mix([a, b, c]).ravel(), want)


def test_resampler_is_the_polyphase_fma_chain():
    C, T, up, down = 2, 8, 3, 2
    proto = synth.resampler_proto(up, down, T)
    x = sig(7, 60, C)
    r = O.Resampler(proto, T, up, down, C)
    got = np.concatenate([r.process(x[:23]), r.process(x[23:])]).reshape(-1, C)
    n_out = -(-x.shape[0] * up // down)
    assert got.shape[0] == n_out
    want = np.empty((n_out, C))
    for m in range(n_out):
        n, p = (m * down) // up, (m * down) % up
        for c in range(C):
            acc = 0.0
            for j in range(T):
                xv = x[n - j, c] if n - j >= 0 else 0.0
                acc = fma(proto[p + j * up], xv, acc)
            want[m, c] = acc
    assert np.array_equal(got, want)

