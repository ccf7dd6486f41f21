"""The bit-exact direct-form FIR on the float64 matrix pipe (pipe_amd/csrc/fir_mfma.hip) against the
oracle's ordered fma chain: bit for bit, every dtype pair, odd channel counts, tap counts around the
block sizes, Lines shorter than the filter, state carried over several calls, and Inf / NaN in the
input (a pass that stages a non-finite value takes the plain ordered loop: zero padding taps must not
meet it).  PIPE_HIP_FIR_MFMA_MIN_PASSES=1 sends these small calls to the kernel that large ones take."""
import numpy as np
import pytest

from oracle import oracle as O
from pipe_amd import synth

pytestmark = pytest.mark.gpu

P = None
torch = None


def setup_module(module):
    global P, torch
    import torch as _t
    from pipe_amd import processors as _p
    assert _t.cuda.is_available(), "-m gpu tests need a GPU; refusing to pass silently"
    P, torch = _p, _t


def sig(seed, frames, channels, dtype):
    return synth.samples(synth.line_seed(seed), 0, frames * channels).reshape(frames, channels).astype(dtype)


def run_calls(taps, x, calls, dtype, lines, monkeypatch, exact=True):
    """x: (lines, total, C); `calls`: frames per call.  Returns the concatenated output and the kernel names."""
    monkeypatch.setenv("PIPE_HIP_FIR_MFMA_MIN_PASSES", "1")
    C = x.shape[-1]
    names = []
    with P.Fir(taps, max(calls), C, dtype=dtype, lines=lines, max_batch=1) as p:
        p.start()
        if exact:
            p.set_exact(True)
        d = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        outs, pos = [], 0
        for n in calls:
            xin = d[:, pos:pos + n, :].contiguous()
            y = torch.full_like(xin, float("nan"))
            p.process_batch(xin, y, n)
            torch.cuda.synchronize()
            names.append(p.kernel_name())
            outs.append(y)
            pos += n
        return torch.cat(outs, dim=1).cpu().numpy(), names


def oracle_lines(taps, x, dtype):
    lines, total, C = x.shape
    return np.stack([O.Fir(taps, C).process(x[l].astype(np.float64)).reshape(total, C).astype(dtype) for l in range(lines)])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("channels,ntaps", [(2, 256), (1, 16), (3, 17), (2, 255), (2, 257), (8, 64), (5, 300), (2, 1024), (2, 2048), (2, 4096), (3, 3000)])
def test_mfma_fir_bit_exact_over_several_calls(dtype, channels, ntaps, monkeypatch):
    lines = 3
    calls = [2500, 1024, 4096, 37]
    total = sum(calls)
    taps = synth.fir_lowpass_taps(ntaps, fc=0.13)
    x = np.stack([sig(400 + l, total, channels, dtype) for l in range(lines)])
    got, names = run_calls(taps, x, calls, dtype, lines, monkeypatch)
    assert all("fir_mfma_kernel" in n for n in names), names
    want = oracle_lines(taps, x, dtype)
    assert np.array_equal(got, want), np.argwhere(got != want)[:4]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mfma_fir_inside_a_staged_chain(dtype, monkeypatch):
    """FIR -> biquad -> gain per buffer: the stages pass float64 between them, so a float32 chain runs the FIR as
    float32 in -> float64 out (the mixed instantiations of the kernel); bit for bit against the oracle chain."""
    monkeypatch.setenv("PIPE_HIP_FIR_MFMA_MIN_PASSES", "1")
    F, C, N = 2048, 8, 256
    h = synth.fir_lowpass_taps(N)
    q = synth.biquad_rbj_lowpass()
    g = 0.7071067811865476
    lens = [F, F, 300]
    x = sig(8, sum(lens), C, dtype)
    kw = dict(dtype=dtype)
    rf, rb = O.Fir(h, C), O.Biquad(q, C)
    with P.Chain([P.Fir(h, F, C, **kw), P.Biquad(q, F, C, **kw), P.Gain(g, F, C, **kw)]) as p:
        p.start()
        pos = 0
        for n in lens:
            got = p.process(x[pos:pos + n])
            want = O.gain(rb.process(rf.process(x[pos:pos + n].astype(np.float64))), g)
            assert np.array_equal(got, np.asarray(want, dtype=np.float64).reshape(n, C).astype(dtype)), pos
            pos += n


@pytest.mark.parametrize("bad", [np.inf, -np.inf, np.nan])
def test_mfma_fir_nonfinite_input_matches_the_oracle(bad, monkeypatch):
    """One Inf / NaN in the stream: the outputs it reaches are the oracle's (NaN / Inf exactly where the ordered
    chain makes them), and the outputs next to its reach -- where a padding zero tap would meet it -- are clean."""
    C, N = 2, 64
    calls = [3000, 3000]
    taps = synth.fir_lowpass_taps(N, fc=0.2)
    x = sig(5, sum(calls), C, np.float64)[None]
    x[0, 1234, 1] = bad
    x[0, 2999, 0] = bad   # its reach crosses into the second call through the history
    got, names = run_calls(taps, x, calls, np.float64, 1, monkeypatch)
    assert all("fir_mfma_kernel" in n for n in names)
    want = oracle_lines(taps, x, np.float64)
    assert np.array_equal(got, want, equal_nan=True)
    assert np.isfinite(got[0, 1233, 1]) and np.isfinite(got[0, 1234 + N, 1]) and not np.isfinite(got[0, 1234, 1])


def test_mfma_fir_lines_shorter_than_the_filter(monkeypatch):
    C, N = 2, 500
    calls = [100, 7, 300, 1]
    taps = synth.fir_lowpass_taps(N, fc=0.05)
    x = np.stack([sig(900 + l, sum(calls), C, np.float32) for l in range(4)])
    got, names = run_calls(taps, x, calls, np.float32, 4, monkeypatch)
    assert all("fir_mfma_kernel" in n for n in names)
    assert np.array_equal(got, oracle_lines(taps, x, np.float32))


def test_large_exact_calls_take_the_matrix_pipe_and_small_ones_do_not(ab_switch):
    """The shipped threshold: a call that gives every CU a pass goes to fir_mfma_kernel, one pipe buffer stays
    on the small-call VALU kernel; both bit-exact (the large one against the VALU form run with the switch off)."""
    C, F, K, N = 2, 4096, 256, 256
    taps = synth.fir_lowpass_taps(N)
    x = sig(3, K * F, C, np.float32)[None]
    outs = {}
    want = O.Fir(taps, C).process(x[0, :8192].astype(np.float64)).reshape(8192, C).astype(np.float32)
    for sw in ("", "1"):
        if sw and not ab_switch("PIPE_HIP_FIR_NO_MFMA", sw):
            return  # (the shipped forms are compared with the oracle above; the VALU form of large calls is an A/B leg)
        with P.Fir(taps, F, C, dtype=np.float32, lines=1, max_batch=K) as p:
            p.start()
            p.set_exact(True)
            d = torch.from_numpy(x).cuda()
            y = torch.empty_like(d)
            p.process_batch(d, y, K * F)
            torch.cuda.synchronize()
            outs[sw] = (y.cpu().numpy(), p.kernel_name())
            one = p.process(x[0, :F])
            assert "fir_direct_kernel" in p.kernel_name()
        if not sw:
            assert "fir_mfma_kernel" in outs[""][1]
            assert np.array_equal(outs[""][0][0, :8192], want)
    assert "fir_direct_kernel" in outs["1"][1]
    assert np.array_equal(outs[""][0], outs["1"][0])
