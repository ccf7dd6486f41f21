"""pipe_hip_process_lines: one multiLineExecutor pass (run.go:112-132) of many Lines through
one handle.  Every Line's state must advance by exactly its own frames -- also when a Source
returns a SHORT READ IN THE MIDDLE of its stream and keeps going (pipe.go:404-406: the pipe
only slices the buffer, nothing ends), when a read is empty, and when Lines end at different
passes.  Checked bit for bit (float64 buffers, as the reference pipe carries them) against the
oracle's per-Line streaming loop fed the same chunks."""
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


F, C, NT = 512, 2, 48
TAPS = synth.fir_lowpass_taps(NT)
Q = synth.biquad_rbj_lowpass()
G = 0.7071067811865476

# frames each of 5 Lines brings per pass; None = the Line has ended (EOF earlier)
PASSES = [
    [512, 512, 512, 512, 512],
    [512, 200, 512, 512, 512],     # Line 1: short read mid-stream
    [512, 512, 512, 0, 512],       # Line 3: empty read
    [512, 512, 77, 512, 512],      # Line 2: short read, runs split on both sides
    [300, 512, 512, 512, None],    # Line 0 short (its last), Line 4 ended
    [None, 512, 512, 1, None],
    [None, 512, None, None, None],
]


def oracle_chain():
    return O.Fir(TAPS, C), O.Biquad(Q, C)


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ragged_passes_advance_every_line_by_its_own_frames(pinned, dtype):
    L_ = 5
    kw = dict(dtype=dtype, lines=L_, max_batch=1)
    stages = [P.Fir(TAPS, F, C, **kw), P.Biquad(Q, F, C, **kw), P.Gain(G, F, C, **kw)]
    streams = [synth.samples(synth.line_seed(300 + l), 0, len(PASSES) * F * C).reshape(-1, C) for l in range(L_)]
    pos = [0] * L_
    refs = [oracle_chain() for _ in range(L_)]
    with P.Chain(stages) as p:
        p.start()
        for frames in PASSES:
            xs = []
            for l, n in enumerate(frames):
                if n is None:
                    xs.append(None)
                else:
                    xs.append(streams[l][pos[l]:pos[l] + n].astype(dtype))
            got = p.process_lines(xs, pinned=pinned)
            for l, n in enumerate(frames):
                if n is None:
                    assert got[l] is None
                    continue
                x = streams[l][pos[l]:pos[l] + n]
                pos[l] += n
                fir, bq = refs[l]
                want = O.gain(bq.process(fir.process(x)), G).reshape(n, C) if n else np.zeros((0, C))
                assert got[l].shape == (n, C)
                assert np.array_equal(got[l], want.astype(dtype)), f"line {l}, pass {frames}"
