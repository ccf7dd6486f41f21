"""The time-segmented form of the biquad Processor (float32 results of large calls).

Tolerance (north_star: "within 1 ULP float32"), written out:
    |gpu - (float)oracle_f64| <= 1 ulp_f32(oracle)
and almost every sample is equal: pass 3 runs the oracle's own ordered recurrence, only its
per-segment start states carry an O(1e-16) relative difference.  float64 buffers, small calls and
PIPE_HIP_PARAM_EXACT keep the one-lane-per-series form, which is bit-exact; both are checked
against each other here.
"""
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
    assert _t.cuda.is_available()
    P, torch = _p, _t


def coeffs(sections):
    q = [synth.biquad_rbj_lowpass(), synth.biquad_rbj_lowpass(fc=4000.0, q=1.3),
         synth.biquad_rbj_lowpass(fc=300.0, q=4.0)]  # the last one rings for ~1000 frames
    return np.vstack(q[:sections])


def run(q, x, lines, calls, exact, dtype_out=np.float32, monkeypatch=None):
    # x: [lines][frames][C] float32; `calls` consecutive launches (state must carry between them)
    L_, frames, C = x.shape
    bounds = np.linspace(0, frames, calls + 1).astype(int)
    with P.Biquad(q, int(np.diff(bounds).max()), C, dtype=np.float32, lines=L_, max_batch=1) as p:
        p.start()
        if exact:
            p.set_exact(True)
        outs = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            xin = torch.from_numpy(np.ascontiguousarray(x[:, a:b, :])).cuda()
            yout = torch.empty_like(xin)
            p.process_batch(xin, yout, int(b - a))
            outs.append(yout)
        torch.cuda.synchronize()
        return torch.cat(outs, dim=1).cpu().numpy(), p.kernel_name()


def oracle(q, x):
    L_, frames, C = x.shape
    return np.stack([O.Biquad(q, C).process(x[l].astype(np.float64)).reshape(frames, C) for l in range(L_)])


@pytest.mark.parametrize("sections", [1, 2, 3])
@pytest.mark.parametrize("lines,channels,frames,calls", [
    (3, 2, 70001, 2),     # ragged: last segment shorter, second call continues the state
    (70, 8, 4096, 1),     # config-3 shape in small: 560 series x 64 segments
    (1, 1, 262144, 3),    # a single series: all the parallelism comes from the segments
])
def test_segmented_matches_oracle_within_one_ulp(sections, lines, channels, frames, calls, monkeypatch):
    monkeypatch.setenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES", "1")
    q = coeffs(sections)
    x = np.stack([synth.samples(synth.line_seed(40 + l), 0, frames * channels, np.float32).reshape(frames, channels)
                  for l in range(lines)])
    got, name = run(q, x, lines, calls, exact=False)
    assert "segmented" in name
    want64 = oracle(q, x)
    want = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want)).astype(np.float64)
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert np.all(d <= ulp), float((d / np.maximum(ulp, 1e-300)).max())
    assert np.count_nonzero(got != want) <= max(2, got.size // 100000)   # "almost every sample"
    ex, name = run(q, x, lines, calls, exact=True)
    assert "segmented" not in name
    assert np.array_equal(ex, want)


def test_float64_buffers_never_take_the_segmented_form(monkeypatch):
    monkeypatch.setenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES", "1")
    q = coeffs(2)
    L_, F, C = 4, 8192, 2
    x = np.stack([synth.samples(synth.line_seed(l), 0, F * C, np.float64).reshape(F, C) for l in range(L_)])
    with P.Biquad(q, F, C, dtype=np.float64, lines=L_) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty_like(d_in)
        p.process_batch(d_in, d_out, F)
        torch.cuda.synchronize()
        assert "segmented" not in p.kernel_name()
        got = d_out.cpu().numpy()
    want = np.stack([O.Biquad(q, C).process(x[l]).reshape(F, C) for l in range(L_)])
    assert np.array_equal(got, want)


def test_config3_chain_uses_both_relaxed_forms_and_stays_within_one_ulp(monkeypatch):
    # FIR (overlap-save) -> biquad (segmented) + gain folded, float32 in/out, f64 in between
    monkeypatch.setenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES", "1")
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    L_, F, C = 24, 4096, 8
    h = synth.fir_lowpass_taps(256)
    q = coeffs(1)
    g = 0.7071067811865476
    x = np.stack([synth.samples(synth.line_seed(l), 0, F * C, np.float32).reshape(F, C) for l in range(L_)])
    kw = dict(dtype=np.float32, lines=L_)
    with P.Chain([P.Fir(h, F, C, **kw), P.Biquad(q, F, C, **kw), P.Gain(g, F, C, **kw)]) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty_like(d_in)
        p.process_batch(d_in, d_out, F)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
    for l in (0, 7, 23):
        w64 = O.gain(O.Biquad(q, C).process(O.Fir(h, C).process(x[l].astype(np.float64))), g).reshape(F, C)
        want = w64.astype(np.float32)
        floor = 2.0 ** -24 * np.abs(h).sum() * g
        ulp = np.spacing(np.maximum(np.abs(want), np.float32(floor))).astype(np.float64)
        d = np.abs(got[l].astype(np.float64) - want.astype(np.float64))
        assert np.all(d <= ulp)
        assert np.count_nonzero(got[l] != want) <= 4


def test_sections_that_are_not_strictly_stable_keep_the_ordered_recurrence(monkeypatch):
    """The relaxed forms (time-segmented biquad, fused chain) carry segment states through powers of
    the state-transition matrix -- sound for poles strictly inside the unit circle only.  An integrator
    (pole on the circle) and a mildly unstable section take the exact kernels whatever the call size:
    bit for bit the oracle's ordered recurrence."""
    monkeypatch.setenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES", "1")
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    lines, C, F = 8, 2, 8192
    x = np.stack([synth.samples(synth.line_seed(70 + l), 0, F * C, np.float32).reshape(F, C) for l in range(lines)]) * 1e-3
    # y[n] = x[n] + y[n-1] (pole at 1); poles at radius sqrt(1.0005) on the imaginary axis
    for q in (np.array([[1.0, 0.0, 0.0, -1.0, 0.0]]), np.array([[1.0, 0.0, 0.0, 0.0, 1.0005]])):
        got, name = run(q, x, lines, 2, exact=False)
        assert "segmented" not in name, name
        want = oracle(q, x).astype(np.float32)
        assert np.array_equal(got, want)
    # and in a chain: FIR -> such a section never takes the fused kernel
    h = synth.fir_lowpass_taps(64, f32_rounded=True)
    q = np.array([[1.0, 0.0, 0.0, -1.0, 0.0]])
    kw = dict(dtype=np.float32, lines=lines)
    with P.Chain([P.Fir(h, F, C, **kw), P.Biquad(q, F, C, **kw)]) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty_like(d_in)
        p.process_batch(d_in, d_out, F)
        torch.cuda.synchronize()
        assert "chain_fused" not in p.kernel_name(), p.kernel_name()
