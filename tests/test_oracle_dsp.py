"""DSP bodies of the oracle cross-checked against scipy (independent float64
implementations).  The reference has no FIR/biquad/resampler, so
this -- not the reference -- is what keeps the oracle honest: "parity unpinned".
scipy sums in a different order, so the comparison is to a few float64 ulps of
the a-priori bound N * eps * sum|h||x|, not bitwise."""
import numpy as np
import pytest
import scipy.signal as ss

from oracle import oracle as O
from pipe_amd import synth

EPS = np.finfo(np.float64).eps


def test_synth_matches_numpy_definition():
    a = O.synth(synth.line_seed(3), 1000, 4096)
    b = synth.samples(synth.line_seed(3), 1000, 4096)
    assert np.array_equal(a, b)
    assert a.min() >= -1.0 and a.max() < 1.0
    assert np.array_equal(a.astype(np.float32).astype(np.float64), a)  # exact in f32
    # known answer for SplitMix64(seed=0): first output of the canonical generator
    assert int(synth.splitmix64(0, 0, 1)[0]) == 0xE220A8397B1DCDAF


@pytest.mark.parametrize("channels", [1, 2, 8])
@pytest.mark.parametrize("ntaps", [1, 2, 31, 256])
def test_fir_matches_lfilter(channels, ntaps):
    rng = np.random.default_rng(ntaps * 10 + channels)
    frames = 1500
    x = rng.uniform(-1, 1, (frames, channels))
    h = synth.fir_lowpass_taps(ntaps) if ntaps > 2 else rng.uniform(-1, 1, ntaps)
    y = O.Fir(h, channels).process(x)
    ref = ss.lfilter(h, [1.0], x, axis=0)
    bound = 4 * ntaps * EPS * np.abs(h).sum()
    assert np.max(np.abs(y - ref)) <= bound


def test_fir_streaming_equals_one_shot_and_reset():
    rng = np.random.default_rng(1)
    C, N = 2, 256
    h = synth.fir_lowpass_taps(N)
    x = rng.uniform(-1, 1, (3000, C))
    whole = O.Fir(h, C).process(x)
    f = O.Fir(h, C)
    # ragged buffers incl. some shorter than the history (N-1) and an empty one
    cuts = [0, 512, 512 + 7, 1024, 1024, 1024 + 100, 2048, 3000]
    parts = [f.process(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts), whole)
    f.reset()
    assert np.array_equal(f.process(x[:700]), whole[:700])


def test_fir_impulse_response_is_taps_bit_exact():
    h = synth.fir_lowpass_taps(256)
    x = np.zeros((600, 2))
    x[5, 0] = 1.0
    x[9, 1] = -1.0
    y = O.Fir(h, 2).process(x)
    assert np.array_equal(y[5:261, 0], h)
    assert np.array_equal(y[9:265, 1], -h)


@pytest.mark.parametrize("channels", [1, 2, 8])
def test_biquad_matches_sosfilt(channels):
    rng = np.random.default_rng(channels)
    x = rng.uniform(-1, 1, (4096, channels))
    q = synth.biquad_rbj_lowpass()
    two = np.vstack([q, synth.biquad_rbj_lowpass(fc=4000.0, q=1.3)])
    for coeffs in (q, two):
        y = O.Biquad(coeffs, channels).process(x)
        sos = np.array([[c[0], c[1], c[2], 1.0, c[3], c[4]] for c in coeffs])
        ref = ss.sosfilt(sos, x, axis=0)
        assert np.max(np.abs(y - ref)) <= 1e-12
    b = O.Biquad(q, channels)
    parts = [b.process(x[:1000]), b.process(x[1000:1001]), b.process(x[1001:])]
    assert np.array_equal(np.concatenate(parts), O.Biquad(q, channels).process(x))


@pytest.mark.parametrize("up,down", [(160, 147), (147, 160), (2, 1), (1, 3)])
def test_resampler_matches_upfirdn(up, down):
    rng = np.random.default_rng(up * 1000 + down)
    C, T = 2, 24
    frames = 2048
    x = rng.uniform(-1, 1, (frames, C))
    proto = synth.resampler_proto(up, down, T)
    r = O.Resampler(proto, T, up, down, C)
    y = r.process(x).reshape(-1, C)
    n_out = -(-frames * up // down)
    assert y.shape[0] == n_out
    ref = ss.upfirdn(proto, x, up=up, down=down, axis=0)[:n_out]
    assert np.max(np.abs(y - ref)) <= 64 * T * EPS * np.abs(proto).max() * up
    # streaming in ragged pieces == one shot
    r2 = O.Resampler(proto, T, up, down, C)
    cuts = [0, 3, 500, 500, 1111, 2048]
    parts = [r2.process(x[a:b]).reshape(-1, C) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts), y)


def test_resampler_capacity_contract():
    # 4096 frames at 44.1k -> 4459 frames at 48k does not fit a
    # 4096-frame out buffer; 3763 input frames is the largest that does.
    T, up, down = 24, 160, 147
    proto = synth.resampler_proto(up, down, T)
    r = O.Resampler(proto, T, up, down, 2)
    assert r.out_frames(4096) == 4459
    assert r.out_frames(3763) == 4096
    with pytest.raises(OverflowError):
        r.process(np.zeros((4096, 2)), out_cap_frames=4096)
    assert r.process(np.zeros((3763, 2)), out_cap_frames=4096).size == 4096 * 2


def test_gain_and_mix():
    rng = np.random.default_rng(5)
    a, b, c = (rng.uniform(-1, 1, 1000) for _ in range(3))
    assert np.array_equal(O.gain(a, 0.5), a * 0.5)
    assert np.array_equal(O.gain(a, 0.7071067811865476), a * 0.7071067811865476)
    assert np.array_equal(O.mix([a, b]), a + b)
    assert np.array_equal(O.mix([a, b, c]), (a + b) + c)
