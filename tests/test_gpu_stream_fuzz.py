"""Seeded random streams through the per-buffer ProcessFunc form: random call lengths (including 0,
1 and full buffers), channel counts, dtypes, filter sizes, a StartFunc reset and a parameter
mutation somewhere in the stream.  Small calls always take the ordered-fma kernels, so every
output must equal the oracle's bit for bit; state that leaks, a history carried wrongly or a tile
geometry that mishandles some length shows up as a mismatch at a reproducible seed."""
import numpy as np
import pytest

from oracle import oracle as O
from pipe_amd import synth

pytestmark = pytest.mark.gpu

P = None


def setup_module(module):
    global P
    import torch
    from pipe_amd import processors as _p
    assert torch.cuda.is_available()
    P = _p


def lengths(rng, F, calls):
    pool = [0, 1, 2, F, F, F - 1, F // 2, 3]
    return [int(rng.choice(pool)) if rng.random() < 0.5 else int(rng.integers(0, F + 1)) for _ in range(calls)]


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("seed", range(12))
def test_fir_stream_fuzz(seed, resident):
    # (resident: the same streams with the next buffer's work queued behind a doorbell, PIPE_HIP_PARAM_RESIDENT --
    # every length change, the reset and the mutation then find queued work to take back)
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.choice([1, 2, 3, 6, 8, 17]))
    F = int(rng.choice([64, 257, 512, 4096]))
    N = int(rng.choice([1, 2, 15, 16, 33, 100, 256, 300]))
    dtype = np.float32 if seed % 2 else np.float64
    h1 = rng.standard_normal(N) / max(N, 1)
    h2 = rng.standard_normal(N) / max(N, 1)
    if dtype == np.float32:
        h1 = h1.astype(np.float32).astype(np.float64)
        h2 = h2.astype(np.float32).astype(np.float64)
    lens = lengths(rng, F, 14)
    reset_at, mutate_at = sorted(rng.choice(np.arange(2, 13), size=2, replace=False))
    ref = O.Fir(h1, C)
    with P.Fir(h1, F, C, dtype=dtype) as p:
        p.start()
        if resident:
            p.set_resident(True)
        for k, n in enumerate(lens):
            if k == reset_at:
                p.start()
                ref = O.Fir(h1 if k < mutate_at else h2, C)
            if k == mutate_at:
                p.set_taps(h2)
                ref.set_taps(h2)
            x = rng.uniform(-1, 1, size=(n, C)).astype(dtype)
            got = p.process(x)
            want = ref.process(x.astype(np.float64)).reshape(n, C).astype(dtype)
            assert got.shape == (n, C)
            assert np.array_equal(got, want), (seed, k, n, C, F, N)


@pytest.mark.parametrize("seed", range(8))
def test_biquad_and_chain_stream_fuzz(seed):
    rng = np.random.default_rng(2000 + seed)
    C = int(rng.choice([1, 2, 5, 8]))
    F = int(rng.choice([128, 512, 2048]))
    S = int(rng.choice([1, 2, 3]))
    dtype = np.float32 if seed % 2 else np.float64
    q = np.vstack([synth.biquad_rbj_lowpass(fc=float(rng.uniform(200, 8000)), q=float(rng.uniform(0.5, 3)))
                   for _ in range(S)])
    N = int(rng.choice([8, 64, 256]))
    h = rng.standard_normal(N) / N
    if dtype == np.float32:
        h = h.astype(np.float32).astype(np.float64)
    g = float(rng.uniform(0.1, 2.0))
    lens = lengths(rng, F, 10)
    kw = dict(dtype=dtype)
    rb, rf, rb2 = O.Biquad(q, C), O.Fir(h, C), O.Biquad(q, C)
    with P.Biquad(q, F, C, **kw) as pb, P.Chain([P.Fir(h, F, C, **kw), P.Biquad(q, F, C, **kw),
                                                  P.Gain(g, F, C, **kw)]) as pc:
        pb.start()
        pc.start()
        for k, n in enumerate(lens):
            x = rng.uniform(-1, 1, size=(n, C)).astype(dtype)
            x64 = x.astype(np.float64)
            got = pb.process(x)
            assert np.array_equal(got, rb.process(x64).reshape(n, C).astype(dtype)), (seed, k, n)
            got = pc.process(x)
            want = O.gain(rb2.process(rf.process(x64)), g).reshape(n, C).astype(dtype)
            assert np.array_equal(got, want), (seed, k, n, "chain")


@pytest.mark.parametrize("seed", range(8))
def test_resampler_and_mix_stream_fuzz(seed):
    rng = np.random.default_rng(3000 + seed)
    C = int(rng.choice([1, 2, 3, 8]))
    F = int(rng.choice([100, 512, 1024]))
    up, down = [(160, 147), (147, 160), (3, 2), (2, 3), (1, 4), (5, 1), (7, 7), (48, 441)][seed]
    T = int(rng.choice([4, 12, 24]))
    dtype = np.float32 if seed % 2 else np.float64
    proto = synth.resampler_proto(up, down, T)
    lens = lengths(rng, F, 10)
    cap = -(-F * up // down) + 1
    ref = O.Resampler(proto, T, up, down, C)
    with P.Resampler(proto, T, up, down, F, C, dtype=dtype) as p, P.Mix(3, F, C, dtype=dtype) as m:
        p.start()
        m.start()
        for k, n in enumerate(lens):
            x = rng.uniform(-1, 1, size=(n, C)).astype(dtype)
            got = p.process(x, out_cap_frames=cap)
            want = ref.process(x.astype(np.float64)).reshape(-1, C).astype(dtype)
            assert got.shape == want.shape, (seed, k, n)
            assert np.array_equal(got, want), (seed, k, n)
            if n:
                xs = [rng.uniform(-1, 1, size=(n, C)).astype(dtype) for _ in range(3)]
                assert np.array_equal(m.process(xs), O.mix([a.astype(np.float64) for a in xs]).astype(dtype))
