"""Committed golden fixtures (tests/golden/, made by gen_golden.py): the oracle must
keep reproducing them bit for bit (CPU), and so must the HIP Processors (GPU)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from pipe_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def stream(seed, frames, channels):
    return synth.samples(synth.line_seed(int(seed)), 0, frames * channels).reshape(frames, channels)


def pieces(cuts):
    cuts = [int(c) for c in cuts]
    return list(zip(cuts[:-1], cuts[1:]))


# ----------------------------------------------------------------------------- CPU
def test_oracle_reproduces_dsp_goldens():
    g = load("fir256_2ch.npz")
    x = stream(g["seed"], int(g["cuts"][-1]), 2)
    f = O.Fir(g["taps"], 2)
    y = np.concatenate([f.process(x[a:b]).reshape(-1, 2) for a, b in pieces(g["cuts"])])
    assert np.array_equal(y, g["y_f64"]) and np.array_equal(y.astype(np.float32), g["y_f32"])

    g = load("biquad2_8ch.npz")
    x = stream(g["seed"], int(g["cuts"][-1]), 8)
    b = O.Biquad(g["coeffs"], 8)
    y = np.concatenate([b.process(x[a:b_]).reshape(-1, 8) for a, b_ in pieces(g["cuts"])])
    assert np.array_equal(y, g["y_f64"])

    g = load("resample_160_147_2ch.npz")
    x = stream(g["seed"], int(g["cuts"][-1]), 2)
    r = O.Resampler(g["proto"], int(g["taps_per_phase"]), int(g["up"]), int(g["down"]), 2)
    parts = [r.process(x[a:b_]).reshape(-1, 2) for a, b_ in pieces(g["cuts"])]
    assert [p.shape[0] for p in parts] == list(g["out_lens"])
    assert np.array_equal(np.concatenate(parts), g["y_f64"])

    g = load("chain_mix_2ch.npz")
    x = stream(g["seed"], 512, 2)
    y = O.gain(O.Biquad(g["coeffs"], 2).process(O.Fir(g["taps"], 2).process(x)), float(g["gain"])).reshape(-1, 2)
    assert np.array_equal(y, g["y_f64"])
    assert np.array_equal(O.mix([x, stream(g["seed2"], 512, 2)]), g["mix_f64"])


def test_oracle_and_host_loop_reproduce_reference_known_answers():
    from pipe_amd import host as H
    g = load("pipe_loop_known_answers.npz")
    buf = int(g["buffer_size"])
    for limit, messages in zip(g["limits"], g["messages"]):
        err, res = O.run_lines(buf, [O.Line(limit=int(limit), channels=1, procs=[O.Proc(O.PROC_COPY)])])
        assert err.ok and res[0].sink.messages == messages and res[0].sink.samples == limit
        herr, hres = H.run(buf, [H.Line(limit=int(limit), channels=1, procs=[H.Proc(H.PROC_MOCK)])])
        assert not herr.failed and hres[0].sink.messages == messages and hres[0].sink.samples == limit
    n_msg, n_frames, ch = (int(v) for v in g["simple_pipe"])
    err, res = O.run_lines(buf, [O.Line(limit=n_frames, channels=ch, procs=[O.Proc(O.PROC_COPY)])])
    assert err.ok and (res[0].source.messages, res[0].source.samples) == (n_msg, n_frames)
    for limit, calls in zip(g["source_limits"], g["source_calls"]):
        err, res = O.run_lines(5, [O.Line(limit=int(limit), channels=2, procs=[])])
        assert res[0].source.messages == calls


# ----------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,key", [(np.float32, "y_f32"), (np.float64, "y_f64")])
def test_hip_reproduces_dsp_goldens(dtype, key):
    from pipe_amd import processors as P

    g = load("fir256_2ch.npz")
    x = stream(g["seed"], int(g["cuts"][-1]), 2).astype(dtype)
    with P.Fir(g["taps"], 1024, 2, dtype=dtype) as p:
        p.start()
        y = np.concatenate([p.process(x[a:b]) for a, b in pieces(g["cuts"])])
    assert np.array_equal(y, g[key])

    g = load("biquad2_8ch.npz")
    x = stream(g["seed"], int(g["cuts"][-1]), 8).astype(dtype)
    with P.Biquad(g["coeffs"], 512, 8, dtype=dtype) as p:
        p.start()
        y = np.concatenate([p.process(x[a:b_]) for a, b_ in pieces(g["cuts"])])
    assert np.array_equal(y, g[key])

    g = load("resample_160_147_2ch.npz")
    x = stream(g["seed"], int(g["cuts"][-1]), 2).astype(dtype)
    with P.Resampler(g["proto"], int(g["taps_per_phase"]), int(g["up"]), int(g["down"]), 512, 2, dtype=dtype) as p:
        p.start()
        parts = [p.process(x[a:b_], out_cap_frames=600) for a, b_ in pieces(g["cuts"])]
    assert [q.shape[0] for q in parts] == list(g["out_lens"])
    assert np.array_equal(np.concatenate(parts), g[key])

    g = load("chain_mix_2ch.npz")
    x = stream(g["seed"], 512, 2).astype(dtype)
    kw = dict(dtype=dtype)
    with P.Chain([P.Fir(g["taps"], 512, 2, **kw), P.Biquad(g["coeffs"], 512, 2, **kw),
                  P.Gain(float(g["gain"]), 512, 2, **kw)]) as p:
        p.start()
        assert np.array_equal(p.process(x), g[key])
    with P.Mix(2, 512, 2, dtype=dtype) as p:
        p.start()
        got = p.process([x, stream(g["seed2"], 512, 2).astype(dtype)])
    assert np.array_equal(got, g["mix_f32" if dtype == np.float32 else "mix_f64"])
