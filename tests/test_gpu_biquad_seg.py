"""The time-segmented form of the biquad Processor (float32 results of large calls).

Tolerance (north_star: "within 1 ULP float32"), written out (include/pipe_hip.h, PIPE_HIP_PARAM_EXACT):
    |gpu - (float)oracle_f64| <= 1 ulp_f32 measured at max(|oracle|, 2^-19 * kappa * max|oracle| of the Line)
and almost every sample is equal: the last pass runs the oracle's own ordered recurrence, only its
per-segment start states differ, by the recurrence's rounding noise, up to ~200 kappa eps of full scale at single samples (kappa: `kappa()` below; 4 .. 21 for the
sections here) -- at a zero crossing 2^-22 below full scale that is more than an ulp of THAT sample.  float64 buffers, small calls and
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


def kappa(q):
    """Largest entry of any power of the cascade's one-frame zero-input transition matrix (state order s1_0, s2_0,
    s1_1, ...): what include/pipe_hip.h scales the relaxed biquad forms' bound with."""
    q = np.atleast_2d(q)
    n = 2 * len(q)
    m = np.zeros((n, n))
    for j in range(n):
        st, x = np.zeros(n), 0.0
        st[j] = 1.0
        for s, (b0, b1, b2, a1, a2) in enumerate(q):
            y = b0 * x + st[2 * s]
            st[2 * s] = -a1 * y + (b1 * x + st[2 * s + 1])
            st[2 * s + 1] = -a2 * y + b2 * x
            x = y
        m[:, j] = st
    p, worst = m.copy(), 0.0
    for _ in range(1 << 16):
        mx = np.abs(p).max()
        worst = max(worst, mx)
        if mx < 1e-3 * worst or worst > 1e6:
            break
        p = p @ m
    return worst


def relaxed_ulp(q, want):
    """include/pipe_hip.h: one float32 ulp measured at max(|y|, 2^-19 * kappa * the Line's full scale)."""
    floor = (2.0 ** -19 * kappa(q) * np.abs(want).max(axis=(1, 2), keepdims=True)).astype(np.float32)
    # (a FLOAT32 ulp whatever type `want` came in: the spacing of a float64 magnitude is 2^29 times finer)
    return np.spacing(np.maximum(np.abs(want).astype(np.float32), floor)).astype(np.float64)


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
    ulp = relaxed_ulp(q, want)
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert np.all(d <= ulp), float((d / np.maximum(ulp, 1e-300)).max())
    assert np.count_nonzero(got != want) <= max(2, got.size // 100000)   # "almost every sample"
    ex, name = run(q, x, lines, calls, exact=True)
    assert "segmented" not in name
    assert np.array_equal(ex, want)


@pytest.mark.parametrize("sections", [1, 2])
@pytest.mark.parametrize("lines,channels,frames,calls", [
    (1, 1, 8192 * 3 + 1, 1),    # one frame into the fourth tile
    (2, 2, 4096 * 5, 2),        # whole tiles, then whole tiles again
    (3, 4, 2048 * 2 + 2047, 3), # ragged calls, a partial segment at the end of each
    (5, 8, 1024 + 31, 1),       # a last tile of less than one segment
    (2, 2, 700, 2),             # calls shorter than a tile
    (1, 1, 1 << 20, 2),         # 64 tiles a call: chained by the wave scan, one tile a lane
    (1, 2, 4096 * 150 + 77, 1), # 151 tiles: three a lane, the last lanes idle
    (3, 8, 1024 * 333, 1),      # 24 series x 333 tiles: six a lane, the last used lane short
    (2, 3, 2720 * 3 + 11, 2),   # 3 channels: 85 segments a channel, one lane of 256 idle
    (1, 6, 1344 * 40, 1),       # 6 channels (42 segments a channel), 40 tiles: the wave scan
    (4, 5, 816 + 1, 1),         # 5 channels, tiles of 16-frame segments, one frame into the second tile
    (3, 7, 20000, 3),           # 7 channels, ragged calls
    (40, 1, 4096, 1),           # mono Lines of half a tile of 32-frame segments: tiles of 16-frame segments
    (7, 2, 2048 + 600, 2),      # the same choice with ragged calls
])
def test_tiled_form_matches_oracle_within_one_ulp(sections, lines, channels, frames, calls, monkeypatch, ab_switch):
    """The LDS-tiled segmented form (up to 8 channels; one or two sections) on tile / segment boundaries; the
    lane-walk form on the same input gives the same bits almost everywhere (same contract, other segment lengths)."""
    monkeypatch.setenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES", "1")
    monkeypatch.setenv("PIPE_HIP_BIQUAD_TILE_MIN_FRAMES", "1")   # (shipped: 512 frames a call, below that the lane walk)
    q = coeffs(3)[[0, 2]][:sections]   # the ringing section second
    x = np.stack([synth.samples(synth.line_seed(90 + l), 0, frames * channels, np.float32).reshape(frames, channels)
                  for l in range(lines)])
    got, name = run(q, x, lines, calls, exact=False)
    assert "biquad_tile_kernel" in name and "segmented" in name, name
    want = oracle(q, x).astype(np.float32)
    # a start state carries the recurrence's rounding noise, up to ~200 kappa eps of full scale (powers of a resonant section's transition matrix are
    # ill-conditioned by ~1 / sin(w0)), which at a deep zero crossing is several ulps of THAT sample
    ulp = relaxed_ulp(q, want)
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert np.all(d <= ulp), float((d / np.maximum(ulp, 1e-300)).max())
    assert np.count_nonzero(got != want) <= max(4, got.size // 100000)
    if not ab_switch("PIPE_HIP_BIQUAD_NO_TILE", "1"):
        return  # (A/B leg: the lane walk forced on a shape the tiles take)
    walk, name = run(q, x, lines, calls, exact=False)
    assert "biquad_kernel" in name, name
    assert np.count_nonzero(got != walk) <= max(4, got.size // 50000)


@pytest.mark.parametrize("shape", ["bg", "gb", "bgg"])
def test_tiled_form_in_a_staged_chain_folds_the_gain(shape, monkeypatch):
    """Staged chains around the biquad: biquad -> gain (float32 in and out, the gain folded into the store),
    gain -> biquad (float64 in, float32 out), biquad -> gain -> gain (float32 in, float64 out with the first gain)."""
    monkeypatch.setenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES", "1")
    L_, F, C = 6, 8192 + 100, 2
    q = coeffs(2)
    g1, g2 = 0.7071067811865476, 1.25
    x = np.stack([synth.samples(synth.line_seed(l), 0, F * C, np.float32).reshape(F, C) for l in range(L_)])
    kw = dict(dtype=np.float32, lines=L_)
    gains = iter((g1, g2))
    stages = [P.Biquad(q, F, C, **kw) if ch == "b" else P.Gain(next(gains), F, C, **kw) for ch in shape]
    with P.Chain(stages) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty_like(d_in)
        p.process_batch(d_in, d_out, F)
        torch.cuda.synchronize()
        if shape[0] == "b":
            assert "biquad_tile_kernel" in p.kernel_name(), p.kernel_name()
        got = d_out.cpu().numpy()
    for l in (0, L_ - 1):
        w = x[l].astype(np.float64).reshape(-1)
        rb = O.Biquad(q, C)
        gains = iter((g1, g2))
        for ch in shape:
            w = rb.process(w) if ch == "b" else O.gain(w, next(gains))
        want = np.asarray(w).reshape(F, C).astype(np.float32)
        ulp = relaxed_ulp(q, want[None])[0]
        d = np.abs(got[l].astype(np.float64) - want.astype(np.float64))
        assert np.all(d <= ulp)
        assert np.count_nonzero(got[l] != want) <= 4


def test_relaxed_bound_scales_with_kappa_and_ill_conditioned_cascades_stay_exact(monkeypatch, ab_switch):
    """100 Hz, Q = 2 (kappa 55): relaxed, inside the kappa-scaled bound, and its measured distance is reported
    against it; 5 Hz, Q = 4 (kappa in the thousands): the ordered recurrence, bit for bit."""
    monkeypatch.setenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES", "1")
    lines, C, F = 4, 2, 4096 * 24
    x = np.stack([synth.samples(synth.line_seed(300 + l), 0, F * C, np.float32).reshape(F, C) for l in range(lines)])
    q = np.vstack([synth.biquad_rbj_lowpass(fc=100.0, q=2.0)])
    assert 40 < kappa(q) < 70
    for no_tile in ("", "1"):
        if no_tile and not ab_switch("PIPE_HIP_BIQUAD_NO_TILE", "1"):
            continue  # (A/B leg: the lane walk on the same samples)
        got, name = run(q, x, lines, 2, exact=False)
        assert "segmented" in name and ("biquad_kernel" in name) == bool(no_tile), name
        want = oracle(q, x).astype(np.float32)
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        assert np.all(d <= relaxed_ulp(q, want)), float((d / relaxed_ulp(q, want)).max())
    monkeypatch.delenv("PIPE_HIP_BIQUAD_NO_TILE", raising=False)
    q = np.vstack([synth.biquad_rbj_lowpass(fc=5.0, q=4.0)])
    assert kappa(q) > 1024
    got, name = run(q, x, lines, 2, exact=False)
    assert "segmented" not in name, name
    assert np.array_equal(got, oracle(q, x).astype(np.float32))


@pytest.mark.parametrize("sections", [3, 4])
@pytest.mark.parametrize("lines,channels", [(1, 2), (3, 1), (2, 5), (40, 8)])
def test_three_and_four_sections_run_as_two_tile_passes(sections, lines, channels, monkeypatch, ab_switch):
    """The tile kernel holds two sections: a cascade of 3 or 4 runs as its two halves, a float64 stream between
    them, the halves' states slices of the handle's own -- so a short call in the ordered form right after a
    long one (and a long one after that) continue the same state."""
    monkeypatch.setenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES", "1")
    q = np.vstack([coeffs(3), synth.biquad_rbj_lowpass(fc=2500.0, q=0.9)])[[0, 2, 1, 3]][:sections]
    calls = [4096 * 3 + 5, 50, 2048 + 700]
    total = sum(calls)
    x = np.stack([synth.samples(synth.line_seed(700 + l), 0, total * channels, np.float32).reshape(total, channels)
                  for l in range(lines)])
    names, outs, pos = [], [], 0
    with P.Biquad(q, max(calls), channels, dtype=np.float32, lines=lines, max_batch=1) as p:
        p.start()
        for n in calls:
            xin = torch.from_numpy(np.ascontiguousarray(x[:, pos:pos + n, :])).cuda()
            y = torch.full_like(xin, float("nan"))
            p.process_batch(xin, y, n)
            torch.cuda.synchronize()
            names.append(p.kernel_name())
            outs.append(y.cpu().numpy())
            pos += n
    assert "two halves" in names[0] and "biquad_tile_kernel" in names[0] and "two halves" in names[2], names
    assert "segmented" not in names[1], names
    got = np.concatenate(outs, axis=1)
    want = oracle(q, x).astype(np.float32)
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert np.all(d <= relaxed_ulp(q, want)), float((d / relaxed_ulp(q, want)).max())
    assert np.count_nonzero(got != want) <= max(4, got.size // 50000)
    if not ab_switch("PIPE_HIP_BIQUAD_NO_SPLIT", "1"):
        return  # (A/B leg: the whole cascade by lane walk)
    walk, name = run(q, x, lines, 1, exact=False)
    assert "two halves" not in name and "segmented" in name
    assert np.count_nonzero(walk != got) <= max(8, got.size // 25000)
    # (A/B leg, the no-tile leg: the halves themselves by lane walk -- the one label the default build cannot report)
    monkeypatch.delenv("PIPE_HIP_BIQUAD_NO_SPLIT")
    assert ab_switch("PIPE_HIP_BIQUAD_NO_TILE", "1")
    halves, name = run(q, x, lines, 1, exact=False)
    assert name == "biquad_kernel<segmented, two halves of the cascade>", name
    d = np.abs(halves.astype(np.float64) - want.astype(np.float64))
    assert np.all(d <= relaxed_ulp(q, want))


def test_calls_shorter_than_512_frames_keep_the_lane_walk(monkeypatch):
    """The shipped threshold for stereo Lines: fewer than 320 frames a call would leave the tiles mostly empty (512 until
    round 6; by channels now: test_short_lines_take_the_tile_form_by_channels)."""
    monkeypatch.setenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES", "1")
    q = coeffs(1)
    for frames, form in ((319, "biquad_kernel"), (320, "biquad_tile_kernel"), (512, "biquad_tile_kernel")):
        x = np.stack([synth.samples(synth.line_seed(500 + l), 0, frames * 2, np.float32).reshape(frames, 2) for l in range(64)])
        got, name = run(q, x, 64, 1, exact=False)
        assert form in name and "segmented" in name, name
        want = oracle(q, x).astype(np.float32)
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        assert np.all(d <= relaxed_ulp(q, want))


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


@pytest.mark.parametrize("sections", [1, 2])
def test_one_float32_pipe_buffer_per_call_takes_the_tile_form(sections):
    """pipe_hip_process, one 4096 x 2 float32 buffer a call (called at pipe.go:438): the ordered recurrence is one
    wave's issue (90 us a buffer); a float32 buffer of 1024 frames or more of few series takes the tile form in one
    short launch, under this file's bound; float64 buffers and PIPE_HIP_PARAM_EXACT keep the ordered form."""
    q = coeffs(sections)
    F, C, B = 4096, 2, 6
    x = synth.samples(synth.line_seed(11), 0, B * F * C, np.float32).reshape(B, F, C)
    want = oracle(q, x.reshape(1, B * F, C))[0].reshape(B, F, C)
    with P.Biquad(q, F, C, dtype=np.float32) as bq, P.Biquad(q, F, C, dtype=np.float32) as ex, \
            P.Biquad(q, F, C, dtype=np.float64) as b64:
        for p in (bq, ex, b64):
            p.start()
        ex.set_exact(True)
        got = np.stack([bq.process(x[k] if k < B - 1 else x[k, :1500]) for k in range(B)][:-1])
        assert bq.kernel_name().startswith("biquad_tile_kernel"), bq.kernel_name()
        w32 = want[:B - 1].astype(np.float32)
        err = np.abs(got.astype(np.float64) - w32.astype(np.float64)) / relaxed_ulp(q, want[:B - 1].reshape(1, -1, C)).reshape(B - 1, F, C)
        assert err.max() <= 1.0, err.max()
        assert (got != w32).sum() <= max(4, got.size // 100000)
        for k in range(B - 1):
            assert np.array_equal(ex.process(x[k]), w32[k])
            assert np.array_equal(b64.process(x[k].astype(np.float64)), want[k])
        assert not ex.kernel_name().startswith("biquad_tile_kernel") and not b64.kernel_name().startswith("biquad_tile_kernel")


@pytest.mark.parametrize("lines,channels,sections", [(16, 8, 1), (100, 2, 1), (200, 1, 1), (37, 3, 1), (1, 2, 4), (16, 8, 3), (100, 2, 4),
                                                     (1, 16, 1), (4, 12, 2), (2, 32, 3)])
def test_long_buffers_of_many_series_take_the_tile_form(lines, channels, sections):
    """A multi-Line pipe's step -- every Line's one pipe buffer in one call (multiLineExecutor, run.go:112-132, through
    pipe_hip_process_lines / process_batch): 65 - 255 series of 4096 frames are neither 2^20 samples nor "at most 64
    series", and until round 6 fell between the two rules onto the ordered form (16 Lines x 8 ch: 117 us; the tile form:
    11, profiles/r06_biquad_dispatch_gap.txt).  The rule is the buffer's LENGTH: 1024 frames or more a Line.  Under this
    file's bound, two calls (the state carries); a call of 1000 frames a Line keeps the ordered form, bit for bit.
    Three and four sections (two tile passes over the halves of the cascade) come under the same rule: one 4096 x 2
    buffer through four sections 100 -> 24 us."""
    q = coeffs(sections)
    F = 4096
    x = np.stack([synth.samples(synth.line_seed(60 + l), 0, 2 * F * channels, np.float32).reshape(2 * F, channels) for l in range(lines)])
    want = oracle(q, x)
    got, name = run(q, x, lines, 2, exact=False)
    # (more than 8 channels: the lane walk over segments -- the tile form holds 8; one 4096 x 16 buffer 141 -> 14 us)
    assert name.startswith("biquad_tile_kernel") if channels <= 8 else "segmented" in name, name
    w32 = want.astype(np.float32)
    err = np.abs(got.astype(np.float64) - w32.astype(np.float64)) / relaxed_ulp(q, want)
    assert err.max() <= 1.0, err.max()
    # ("almost every sample": two tile passes round twice -- measured 1.3 per 100 000 with three sections of kappa 21)
    assert (got != w32).sum() <= max(4, got.size // (100000 if sections <= 2 else 50000))
    short, sname = run(q, x[:, :1000], lines, 1, exact=False)
    assert not sname.startswith("biquad_tile_kernel") and "segmented" not in sname, sname
    assert np.array_equal(short, w32[:, :1000])


@pytest.mark.parametrize("lines,channels,frames,tile", [(2800, 2, 384, True), (3300, 2, 320, True), (4200, 2, 256, False), (1100, 8, 256, True),
                                                        (1400, 8, 128, False), (5600, 1, 384, True), (6600, 1, 320, False), (1400, 4, 320, True),
                                                        (1900, 3, 384, True), (2200, 3, 320, False)])
def test_short_lines_take_the_tile_form_by_channels(lines, channels, frames, tile):
    """Thousands of short Lines in one call (>= 2^20 samples): the shortest Line the LDS-tile form takes depends on the
    channel count since round 6 (8 channels 256 frames, 2 and 4 channels 320, the others 384; it was 512 for all --
    profiles/r06_dispatch_audit.txt); below it the lane walk over segments.  Both under this file's bound."""
    q = coeffs(1)
    x = np.stack([synth.samples(synth.line_seed(200 + l), 0, frames * channels, np.float32).reshape(frames, channels) for l in range(lines)])
    want = oracle(q, x)
    got, name = run(q, x, lines, 1, exact=False)
    assert name.startswith("biquad_tile_kernel") == tile and "segmented" in name, name
    w32 = want.astype(np.float32)
    err = np.abs(got.astype(np.float64) - w32.astype(np.float64)) / relaxed_ulp(q, want)
    assert err.max() <= 1.0, err.max()
    assert (got != w32).sum() <= max(4, got.size // 100000)


@pytest.mark.parametrize("sections", [1, 2])
def test_float64_buffers_take_the_tile_form_only_when_asked(sections):
    """PIPE_HIP_PARAM_RELAXED_F64: the buffers the Go pipe carries (pipe.go:394,437: float64) may take the tile form
    as an explicit opt-in -- per 4096 x 2 call one short launch instead of 96 us of ordered recurrence.  The bound as
    include/pipe_hip.h states it: |y - oracle| <= 256 kappa 2^-53 max|oracle of the Line|.  Without the parameter, and
    with PIPE_HIP_PARAM_EXACT on top of it, float64 buffers are bit for bit the oracle's.  Queued ahead
    (PIPE_HIP_PARAM_RESIDENT) the same handle gives the same bits as on the plain path."""
    q = coeffs(sections)
    F, C, B = 4096, 2, 6
    x = synth.samples(synth.line_seed(12), 0, B * F * C, np.float64).reshape(B, F, C)
    want = oracle(q, x.reshape(1, B * F, C))[0].reshape(B, F, C)
    bound = 256.0 * kappa(q) * 2.0 ** -53 * np.abs(want).max()
    with P.Biquad(q, F, C, dtype=np.float64) as rel, P.Biquad(q, F, C, dtype=np.float64) as ex, \
            P.Biquad(q, F, C, dtype=np.float64) as res:
        for p in (rel, ex, res):
            p.start()
            p.set_relaxed_f64(True)
        ex.set_exact(True)
        assert res.set_resident(True)
        worst = 0.0
        for k in range(B):
            got = rel.process(x[k])
            assert rel.kernel_name().startswith("biquad_tile_kernel<f64,f64"), rel.kernel_name()
            worst = max(worst, np.abs(got - want[k]).max())
            assert np.array_equal(res.process(x[k]), got), k
            assert np.array_equal(ex.process(x[k]), want[k])
        assert worst <= bound, (worst, bound)
        assert not ex.kernel_name().startswith("biquad_tile_kernel")
        # a short buffer (the ordered form), then the stream goes on
        tail = x[0, :300]
        assert np.array_equal(rel.process(tail), res.process(tail))
        rel.set_relaxed_f64(False)
        rel.start()
        assert np.array_equal(rel.process(x[0]), want[0])


def test_a_tile_launch_that_gives_up_is_run_again_through_the_ordered_recurrence():
    """PIPE_HIP_PARAM_DEBUG on the biquad stage alone: the one-pass tile launch of a synchronous call fails on demand
    (tile 1 of every series publishes nothing); the carried state goes back to the copy tile 0 made of what it read
    and the call runs again through the ordered recurrence: bit for bit the oracle's, and the stream goes on."""
    q = coeffs(1)
    lines, C, F, B = 8, 2, 16384, 4   # 4 tiles of 4096 frames per Line and call
    x = np.stack([synth.samples(synth.line_seed(20 + l), 0, B * F * C, np.float32).reshape(B * F, C) for l in range(lines)])
    want = oracle(q, x)
    with P.Biquad(q, F, C, dtype=np.float32, lines=lines) as bq:
        bq.start()
        for k in range(B):
            if k == 2:
                bq._set_param(5, [1.0, 2000.0])  # PIPE_HIP_PARAM_DEBUG {tile, limit_us}
            got = bq.process(np.ascontiguousarray(x[:, k * F:(k + 1) * F])).reshape(lines, F, C)
            w = want[:, k * F:(k + 1) * F]
            if k == 2:
                assert not bq.kernel_name().startswith("biquad_tile_kernel"), bq.kernel_name()
                assert np.array_equal(got, w.astype(np.float32))
            else:
                assert bq.kernel_name().startswith("biquad_tile_kernel"), bq.kernel_name()
                err = np.abs(got.astype(np.float64) - w.astype(np.float32).astype(np.float64)) / relaxed_ulp(q, want)[:, k * F:(k + 1) * F]
                assert err.max() <= 1.0, (k, err.max())


def test_a_staged_chain_whose_tile_biquad_gives_up_still_answers_the_stream():
    """FIR -> biquad -> gain as the STAGED chain (a call too small for the fused kernel): the biquad stage's tile launch
    fails on demand; the stage is run again through the ordered recurrence on the chain's float64 intermediate, with the
    chain's gain still folded into its store."""
    q = coeffs(1)
    lines, C, F, N = 4, 2, 16384, 64
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    x = np.stack([synth.samples(synth.line_seed(40 + l), 0, 3 * F * C, np.float32).reshape(3 * F, C) for l in range(lines)])
    want = np.stack([O.Biquad(q, C).process(O.Fir(taps, C).process(x[l].astype(np.float64))).reshape(3 * F, C) * 0.25
                     for l in range(lines)])
    kw = dict(dtype=np.float32, lines=lines)
    with P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(q, F, C, **kw), P.Gain(0.25, F, C, **kw)]) as ch:
        ch.start()
        for k in range(3):
            if k == 1:
                ch.set_stage_param(1, 5, [1.0, 2000.0])  # PIPE_HIP_PARAM_DEBUG on the biquad stage
            got = ch.process(np.ascontiguousarray(x[:, k * F:(k + 1) * F])).reshape(lines, F, C)
            w = want[:, k * F:(k + 1) * F]
            err = np.abs(got.astype(np.float64) - w.astype(np.float32).astype(np.float64)) / relaxed_ulp(q / np.array([1, 1, 1, 1, 1.0]), want)[:, k * F:(k + 1) * F]
            assert err.max() <= 1.0, (k, err.max())
