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


# ---- large calls: chunks of Lines, H2D(k + 1) | kernels(k) | D2H(k - 1) ------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("entry", ["process", "process_lines", "process_lines_pinned", "process_lines_slab"])
def test_overlapped_chunks_of_lines_change_no_bit(entry, dtype, monkeypatch):
    """A call above PIPE_HIP_OVERLAP_MIN_BYTES is cut into chunks of whole Lines whose transfers and
    kernels overlap (abi.hip).  State is per Line, so the result must be the unchunked call's, bit for
    bit, over several passes (history and cascade state carried), with a Line that ends and a short
    last buffer; and equal to the oracle's per-Line loop."""
    L_, Fb, Cb = 37, 256, 4                      # 37 Lines: chunks of 10, 10, 10, 7
    lens = [Fb, Fb, 100]
    streams = [synth.samples(synth.line_seed(700 + l), 0, sum(lens) * Cb).reshape(-1, Cb) for l in range(L_)]

    def run(overlap):
        monkeypatch.setenv("PIPE_HIP_OVERLAP_MIN_BYTES", "1" if overlap else str(1 << 40))
        kw = dict(dtype=dtype, lines=L_, max_batch=1)
        outs = []
        with P.Chain([P.Fir(TAPS, Fb, Cb, **kw), P.Biquad(Q, Fb, Cb, **kw), P.Gain(G, Fb, Cb, **kw)]) as p:
            p.start()
            pos = 0
            for k, n in enumerate(lens):
                if entry == "process":
                    x = np.stack([s[pos:pos + n] for s in streams]).astype(dtype)
                    outs.append(p.process(x))
                else:
                    xs = [s[pos:pos + n].astype(dtype) for s in streams]
                    if k == 2:
                        xs[5] = None             # Line 5 has ended: its slot rides along as silence
                    # (pinned rows: read and written in place -- one allocation per row by the row kernels, rows
                    # carved from one block by the DMA engines while they are all there and equally long)
                    got = p.process_lines(xs, pinned={"process_lines": False, "process_lines_pinned": True,
                                                      "process_lines_slab": "slab"}[entry])
                    outs.append(np.stack([g if g is not None else np.zeros((n, Cb), dtype) for g in got]))
                pos += n
            p.flush()
        return outs

    plain, chunked = run(False), run(True)
    for a, b in zip(plain, chunked):
        assert np.array_equal(a, b)
    for l in (0, 9, 10, 29, 30, 36):
        fir, bq = O.Fir(TAPS, Cb), O.Biquad(Q, Cb)
        pos = 0
        for k, n in enumerate(lens):
            want = O.gain(bq.process(fir.process(streams[l][pos:pos + n])), G).reshape(n, Cb).astype(dtype)
            assert np.array_equal(chunked[k][l], want), (l, k)
            pos += n


def test_overlapped_call_at_the_config3_shape_against_the_exact_chain():
    """BASELINE configs[3] from HOST buffers: 512 Lines x 4096 x 8 float32 in one pipe_hip_process call
    (67 MB each way), the call cut into chunks of Lines.  Windows of Lines take the staged chain, whose
    float32 result is the oracle's rounded value: spot Lines against the oracle, bit for bit."""
    L_, Fb, Cb = 512, 4096, 8
    taps = synth.fir_lowpass_taps(256, f32_rounded=True)
    x = np.stack([synth.samples(synth.line_seed(900 + l), 0, Fb * Cb, np.float32).reshape(Fb, Cb) for l in range(L_)])
    kw = dict(dtype=np.float32, lines=L_, max_batch=1)
    with P.Chain([P.Fir(taps, Fb, Cb, **kw), P.Biquad(Q, Fb, Cb, **kw), P.Gain(G, Fb, Cb, **kw)]) as p:
        p.start()
        p._set_param(3, [1.0])      # PIPE_HIP_PARAM_EXACT: the ordered forms, so that every bit is the oracle's
        got = p.process(x)
        p.flush()
    assert got.shape == x.shape and not np.isnan(got).any()
    for l in (0, 63, 64, 255, 256, 448, 511):
        want = O.gain(O.Biquad(Q, Cb).process(O.Fir(taps, Cb).process(x[l].astype(np.float64))), G).reshape(Fb, Cb)
        assert np.array_equal(got[l], want.astype(np.float32)), l
