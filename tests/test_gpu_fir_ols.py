"""The overlap-save (float64 FFT) form of the FIR Processor.

Tolerance (north_star: "within 1 ULP float32"), written out: for float32 buffers
    |gpu - (float)oracle_f64| <= 1 ulp_f32( max(|oracle|, 2^-24 * ||h||_1 * max|x|) )
i.e. one float32 ulp, measured no finer than 24 bits below the filter's full-scale
output (below that the oracle's own ordered sum is uncertain by N*2^-53*||h||_1).
The direct form stays bit-exact and is what float64 buffers, small calls and
PIPE_HIP_PARAM_EXACT use; both are checked against each other here.
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


def ulp_diff_f32(got, want64, floor_mag):
    """|got - want| in units of the float32 ulp at max(|want|, floor_mag)."""
    want32 = want64.astype(np.float32)
    mag = np.maximum(np.abs(want64), floor_mag).astype(np.float32)
    ulp = np.spacing(mag).astype(np.float64)
    return np.abs(got.astype(np.float64) - want32.astype(np.float64)) / ulp


def run_batch(taps, x, F, K, lines, exact):
    C = x.shape[-1]
    with P.Fir(taps, F, C, dtype=np.float32, lines=lines, max_batch=K) as p:
        p.start()
        if exact:
            p.set_exact(True)
        d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        d_out = torch.empty_like(d_in)
        outs = []
        half = (K // 2) * F
        # two launches: history must carry between them in this form too
        frames = x.shape[-2]
        for a, b in ((0, half), (half, frames)):
            xin = d_in[..., a:b, :].contiguous()
            yout = torch.empty_like(xin)
            p.process_batch(xin, yout, b - a)
            outs.append(yout)
        torch.cuda.synchronize()
        name = p.kernel_name()
        return torch.cat(outs, dim=-2).cpu().numpy(), name


@pytest.mark.parametrize("channels,ntaps,F,K", [(2, 1024, 1024, 24), (2, 513, 1024, 24), (4, 700, 512, 40), (2, 4096, 4096, 8),
                                                 (8, 1500, 2048, 6), (2, 2049, 300, 70),
                                                 (1, 1024, 1024, 24), (1, 2100, 512, 41), (1, 4096, 700, 9)])  # (one channel: two tiles per transform)
@pytest.mark.parametrize("form", ["delay_line", "partition_sum"])
def test_partitioned_ols_long_filters_within_one_ulp(channels, ntaps, F, K, form, monkeypatch, ab_switch):
    """513 .. 4096 taps: several <= 512-tap spectra whose products are summed in the frequency domain
    (fir_ols32p.hip) -- as a frequency-domain delay line (one forward transform per 512-frame tile, the
    last P spectra in a ring: the shipped form) and as a sum over the partitions' own windows (A/B form).
    Same contract as the one-spectrum form; two launches, so the N - 1 frames of history (deeper than a
    tile) carry across; the direct form on the same data stays bit-exact."""
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    if form == "partition_sum":
        if channels == 1:
            pytest.skip("the A/B form takes channel pairs only")
        if not ab_switch("PIPE_HIP_FIR_PARTITION_SUM", "1"):
            pytest.skip("the partition-sum form exists in the A/B build only (the shipped form is `delay_line`)")
    lines = 2
    taps = synth.fir_lowpass_taps(ntaps, fc=0.11, f32_rounded=True)
    x = np.stack([synth.samples(synth.line_seed(80 + l), 0, K * F * channels, np.float32).reshape(K * F, channels)
                  for l in range(lines)])
    got, name = run_batch(taps, x, F, K, lines, exact=False)
    assert "partitioned" in name, name
    ref, name2 = run_batch(taps, x, F, K, lines, exact=True)
    assert "fir_direct_kernel" in name2 or "fir_mfma_kernel" in name2  # (the ordered-fma form: VALU or matrix pipe, by size)
    floor = 2.0 ** -24 * np.abs(taps).sum() * 1.0
    for l in range(lines):
        want = O.Fir(taps, channels).process(x[l].astype(np.float64)).reshape(K * F, channels)
        assert np.array_equal(ref[l], want.astype(np.float32))
        d = ulp_diff_f32(got[l], want, floor)
        assert d.max() <= 1.0, f"line {l}: max {d.max()} ulp at {np.unravel_index(d.argmax(), d.shape)}"
        assert np.mean(got[l] != want.astype(np.float32)) < 1e-4


@pytest.mark.parametrize("channels,ntaps", [(2, 256), (2, 64), (2, 511), (4, 256), (3, 256), (1, 128)])
def test_ols_within_one_ulp_of_oracle_and_of_direct_form(channels, ntaps, monkeypatch):
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")  # take the FFT form even for this small batch
    F, K, lines = 1024, 24, 2
    taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
    x = np.stack([synth.samples(synth.line_seed(50 + l), 0, K * F * channels, np.float32).reshape(K * F, channels)
                  for l in range(lines)])
    got, name = run_batch(taps, x, F, K, lines, exact=False)
    assert "fir_ols_kernel" in name
    ref, name2 = run_batch(taps, x, F, K, lines, exact=True)
    assert "fir_direct_kernel" in name2 or "fir_mfma_kernel" in name2  # (the ordered-fma form: VALU or matrix pipe, by size)
    floor = 2.0 ** -24 * np.abs(taps).sum() * 1.0
    for l in range(lines):
        want = O.Fir(taps, channels).process(x[l].astype(np.float64)).reshape(K * F, channels)
        assert np.array_equal(ref[l], want.astype(np.float32))  # direct form: bit-exact
        d = ulp_diff_f32(got[l], want, floor)
        assert d.max() <= 1.0, f"line {l}: max {d.max()} ulp"
        # and almost always it is the very same float32
        assert np.mean(got[l] != want.astype(np.float32)) < 1e-4


def test_ols_impulse_and_linearity_properties(monkeypatch):
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    F, K, C, N = 4096, 64, 2, 256
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    n = K * F
    x = np.zeros((n, C), np.float32)
    pos = 100_000
    x[pos, 0] = 1.0
    x[pos + 7, 1] = -0.5
    with P.Fir(taps, F, C, dtype=np.float32, max_batch=K) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty_like(d_in)
        p.process_batch(d_in, d_out, n)
        torch.cuda.synchronize()
        assert "fir_ols_kernel" in p.kernel_name()
        y = d_out.cpu().numpy()
    h32 = taps.astype(np.float32)
    # impulse response == taps to float32 rounding (1 ulp), zeros elsewhere up to FFT noise
    assert np.max(np.abs(y[pos:pos + N, 0] - h32)) <= np.spacing(np.abs(h32).max())
    assert np.max(np.abs(y[pos + 7:pos + 7 + N, 1] + 0.5 * h32)) <= np.spacing(np.abs(h32).max())
    mask = np.ones(n, bool)
    mask[pos:pos + N + 7] = False
    assert np.max(np.abs(y[mask])) < 1e-14


def test_ols_set_taps_and_restart(monkeypatch):
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    F, K, C = 2048, 8, 2
    h1 = synth.fir_lowpass_taps(128, f32_rounded=True)
    h2 = synth.fir_lowpass_taps(128, fc=0.1, f32_rounded=True)
    x = synth.samples(synth.line_seed(60), 0, 2 * K * F * C, np.float32).reshape(2 * K * F, C)
    ref = O.Fir(h1, C)
    floor = 2.0 ** -24 * max(np.abs(h1).sum(), np.abs(h2).sum())
    with P.Fir(h1, F, C, dtype=np.float32, max_batch=K) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty_like(d_in)
        p.process_batch(d_in[:K * F], d_out[:K * F], K * F)
        p.set_taps(h2)  # mutation: history is kept, the next batch uses the new spectrum
        p.process_batch(d_in[K * F:], d_out[K * F:], K * F)
        torch.cuda.synchronize()
        y = d_out.cpu().numpy()
        w1 = ref.process(x[:K * F].astype(np.float64)).reshape(-1, C)
        ref.set_taps(h2)
        w2 = ref.process(x[K * F:].astype(np.float64)).reshape(-1, C)
        assert ulp_diff_f32(y[:K * F], w1, floor).max() <= 1.0
        assert ulp_diff_f32(y[K * F:], w2, floor).max() <= 1.0
        p.start()  # StartFunc: history zeroed
        p.process_batch(d_in[:K * F], d_out[:K * F], K * F)
        torch.cuda.synchronize()
        ref2 = O.Fir(h2, C)
        w3 = ref2.process(x[:K * F].astype(np.float64)).reshape(-1, C)
        assert ulp_diff_f32(d_out[:K * F].cpu().numpy(), w3, floor).max() <= 1.0


@pytest.mark.parametrize("K", [4096, 131072])
def test_ols_full_bench_size_against_bit_exact_form(K):
    # BASELINE-size streams (1 Line x 2 ch x K buffers of 4096 frames, float32, 256 taps; K = 131072
    # is bench.py's default step: 4.3 GB per buffer, offsets beyond 32 bits): the overlap-save form
    # against the bit-exact direct form on the device, every sample, and the direct form's last
    # frames against the oracle.
    F, C, N = 4096, 2, 256
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    n = K * F * C
    d_in = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    y_ols = torch.empty_like(d_in)
    y_ref = torch.empty_like(d_in)
    with P.Fir(taps, F, C, dtype=np.float32, max_batch=K) as p:
        p.start()
        p.process_batch(d_in, y_ols, K * F)
        torch.cuda.synchronize()
        assert "fir_ols_kernel" in p.kernel_name()
        p.start()
        p.set_exact(True)
        p.process_batch(d_in, y_ref, K * F)
        torch.cuda.synchronize()
        assert "fir_mfma_kernel" in p.kernel_name() or "fir_direct_kernel" in p.kernel_name()  # the ordered-fma form
    floor = float(np.float32(2.0 ** -24 * np.abs(taps).sum()))
    differ = y_ols != y_ref
    n_diff = int(differ.sum().item())
    # ulp distance through the ordered-integer view of IEEE floats
    def key(t):
        i = t.view(torch.int32).to(torch.int64)
        return torch.where(i < 0, -(i & 0x7FFFFFFF), i)
    big = y_ref.abs() >= floor
    ulps = (key(y_ols) - key(y_ref)).abs()
    max_ulp_big = int(ulps[big].max().item())
    small_abs = float((y_ols - y_ref).abs()[~big].max().item()) if int((~big).sum().item()) else 0.0
    print(f"\n[ols vs bit-exact] samples={n} differ={n_diff} ({n_diff / n:.2e}) max_ulp={max_ulp_big} "
          f"below-floor={int((~big).sum().item())} max_abs_below_floor={small_abs:.3e}")
    assert max_ulp_big <= 1
    assert small_abs <= float(np.spacing(np.float32(floor)))
    assert n_diff / n < 1e-5
    lo = n // C - 4000  # 64-bit addressing: the end of the stream against the oracle
    x = d_in[lo * C:].cpu().numpy().reshape(-1, C).astype(np.float64)
    want = O.Fir(taps, C).process(x).reshape(-1, C)[N:].astype(np.float32)
    assert np.array_equal(y_ref[(lo + N) * C:].cpu().numpy().reshape(-1, C), want)
    # ... and the overlap-save output ITSELF against the oracle (not only through the direct form):
    # the head of the stream (zero history, the Line's first tiles), a window in the middle, and
    # the last frames (byte offsets at the top of / beyond 32 bits for K = 131072)
    W = 6000
    frames_total = n // C
    for start in (0, (frames_total // 2 // 769) * 769 - 1000, frames_total - W):
        xs = d_in[start * C:(start + W) * C].cpu().numpy().reshape(-1, C).astype(np.float64)
        w64 = O.Fir(taps, C).process(xs).reshape(-1, C)
        skip = 0 if start == 0 else N          # a window inside the stream needs N-1 frames of run-in
        g = y_ols[(start + skip) * C:(start + W) * C].cpu().numpy().reshape(-1, C)
        d = ulp_diff_f32(g, w64[skip:], floor)
        assert d.max() <= 1.0, f"ols vs oracle at frame {start}: {d.max()} ulp"
        assert np.mean(g != w64[skip:].astype(np.float32)) < 1e-3


def test_large_f32_chain_uses_ols_and_folded_gain_within_one_ulp(monkeypatch):
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    # BASELINE config[3] shape (many Lines x 8 ch, FIR + biquad + gain), large enough for the
    # overlap-save FIR inside the chain; biquad+gain run as one pass.  Final float32 output is
    # within 1 ulp of the oracle chain, and the exact-mode chain is bit-exact.
    L_, F, C, N = 24, 4096, 8, 256
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    q = synth.biquad_rbj_lowpass()
    g = 0.7071067811865476
    x = np.stack([synth.samples(synth.line_seed(70 + l), 0, F * C, np.float32).reshape(F, C) for l in range(L_)])
    floor = 2.0 ** -24 * np.abs(taps).sum()
    for exact in (False, True):
        kw = dict(dtype=np.float32, lines=L_, max_batch=1)
        fir = P.Fir(taps, F, C, **kw)
        if exact:
            fir.set_exact(True)
        with P.Chain([fir, P.Biquad(q, F, C, **kw), P.Gain(g, F, C, **kw)]) as p:
            p.start()
            d_in = torch.from_numpy(x).cuda()
            d_out = torch.empty_like(d_in)
            p.process_batch(d_in, d_out, F)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy()
            name = p.kernel_name()
        assert ("fir_direct" in name or "fir_mfma" in name) if exact else ("chain_fused" in name or "fir_ols" in name)
        for l in (0, 7, 23):
            want = O.gain(O.Biquad(q, C).process(O.Fir(taps, C).process(x[l].astype(np.float64))), g).reshape(F, C)
            if exact:
                assert np.array_equal(got[l], want.astype(np.float32))
            else:
                assert ulp_diff_f32(got[l], want, floor).max() <= 1.0


@pytest.mark.parametrize("lines,channels,frames,ntaps", [
    (1, 2, 769 * 300 + 17, 256),   # 301 tiles: fewer blocks than CUs x 16 waves, ragged last tile
    (3, 6, 50_000, 100),           # 3 channel pairs: wave groups of 4 > pairs
    (7, 16, 20_000, 256),          # 8 pairs, several Lines, grid not a multiple of 8 XCDs
    (37, 2, 9_000, 33),            # many short Lines: most tiles are a Line's first (history) tile
    (2, 64, 12_345, 512),          # the channel limit and the longest supported filter
    (5, 5, 7_777, 16),             # odd channel count (scalar-pair path), shortest supported filter
    (64, 2, 64 * 4096, 256),       # BASELINE configs[2] at full size: 64 Lines x 64 buffers of 4096 x 2
    (512, 8, 4096, 256),           # BASELINE configs[3] at full size: 512 Lines x 8 ch x one 4096 buffer
])
def test_ols_item_dealing_shapes(lines, channels, frames, ntaps, monkeypatch):
    # every (Line, channel pair, tile) item must be produced exactly once whatever the grid,
    # the wave grouping and the XCD-aware order make of the shape: compare with the direct form
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
    rng = np.random.default_rng(1234 + lines * 100 + channels)
    x = rng.uniform(-1, 1, size=(lines, frames, channels)).astype(np.float32)
    outs = {}
    for exact in (False, True):
        with P.Fir(taps, frames, channels, dtype=np.float32, lines=lines, max_batch=1) as p:
            p.start()
            if exact:
                p.set_exact(True)
            d_in = torch.from_numpy(x).cuda()
            half = frames // 3
            ys = []
            for a, b in ((0, half), (half, frames)):   # two calls: the history must carry
                xin = d_in[:, a:b, :].contiguous()
                y = torch.full_like(xin, float("nan"))
                p.process_batch(xin, y, b - a)
                ys.append(y)
            torch.cuda.synchronize()
            outs[exact] = (torch.cat(ys, dim=1).cpu().numpy(), p.kernel_name())
    got, name = outs[False]
    ref, name2 = outs[True]
    assert "fir_ols_kernel" in name and ("fir_direct_kernel" in name2 or "fir_mfma_kernel" in name2)
    assert not np.isnan(got).any()
    floor = 2.0 ** -24 * np.abs(taps).sum()
    d = ulp_diff_f32(got, ref.astype(np.float64), floor)
    assert d.max() <= 1.0, float(d.max())
    assert np.mean(got != ref) < 1e-3
    # spot Lines against the oracle itself (first, one in the middle, last), both forms
    for l in sorted({0, lines // 2, lines - 1}):
        want = O.Fir(taps, channels).process(x[l].astype(np.float64)).reshape(frames, channels)
        assert np.array_equal(ref[l], want.astype(np.float32)), f"direct form, line {l}"
        assert ulp_diff_f32(got[l], want, floor).max() <= 1.0, f"overlap-save form, line {l}"


@pytest.mark.parametrize("ntaps,run_floor", [(1100, None), (2100, None), (1100, "4")])
def test_partitioned_delay_line_many_short_lines_three_calls(ntaps, run_floor, monkeypatch, ab_switch):
    """The delay line's runs on many short Lines (runs as short as P tiles -- what fills the chip -- or,
    with the older floor of 4 P, longer than a Line's tiles; halves without a run; Lines ending inside a
    tile; every run opening with P - 1 warm-up windows out of the history) over three calls of different
    lengths: every Line against the oracle."""
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    if run_floor and not ab_switch("PIPE_HIP_FIR_RUN_FLOOR", run_floor):
        pytest.skip("the older run floor is an A/B-build switch (the shipped floor is the `None` parametrisation)")
    lines, C, F = 37, 4, 700
    taps = synth.fir_lowpass_taps(ntaps, fc=0.07, f32_rounded=True)
    calls = [3 * F, F, 5 * F - 13]
    total = sum(calls)
    x = np.stack([synth.samples(synth.line_seed(300 + l), 0, total * C, np.float32).reshape(total, C) for l in range(lines)])
    with P.Fir(taps, F, C, dtype=np.float32, lines=lines, max_batch=8) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        outs, pos = [], 0
        for n in calls:
            xin = d_in[:, pos:pos + n, :].contiguous()
            y = torch.full_like(xin, float("nan"))
            p.process_batch(xin, y, n)
            torch.cuda.synchronize()
            assert "partitioned" in p.kernel_name()
            outs.append(y)
            pos += n
        got = torch.cat(outs, dim=1).cpu().numpy()
    assert not np.isnan(got).any()
    floor = 2.0 ** -24 * np.abs(taps).sum()
    for l in range(lines):
        want = O.Fir(taps, C).process(x[l].astype(np.float64)).reshape(total, C)
        d = ulp_diff_f32(got[l], want, floor)
        assert d.max() <= 1.0, f"line {l}: {d.max()} ulp at {np.unravel_index(d.argmax(), d.shape)}"


@pytest.mark.parametrize("ntaps", [256, 1024])
def test_streams_that_start_at_an_odd_frame_keep_the_fast_kernels(ntaps, monkeypatch):
    """A device buffer that starts at an odd frame of its allocation is aligned to a channel pair (8 bytes of
    float32), not to 16: the overlap-save kernels, the fused chain and the pair resampler access pairs / use
    element-aligned 16-byte loads, so such a view takes the same kernel and gives the same bits as an aligned copy."""
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    C, F, K, lines = 2, 4096, 8, 3
    taps = synth.fir_lowpass_taps(ntaps, fc=0.1, f32_rounded=True)
    n = lines * K * F * C
    data = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(data, synth.line_seed(5))
    big = torch.empty(n + 16, dtype=torch.float32, device="cuda")

    def view(off):  # the same samples in a buffer that starts `off` elements into its allocation
        v = big[off:off + n]
        v.copy_(data)
        return v
    outs = {}
    for off in (0, 2, 6):  # elements: 0, one frame, three frames
        src = view(off)
        dst_big = torch.full((n + 16,), float("nan"), dtype=torch.float32, device="cuda")
        dst = dst_big[off:off + n]
        with P.Fir(taps, F, C, dtype=np.float32, lines=lines, max_batch=K) as p:
            p.start()
            p.process_batch(src, dst, K * F)
            torch.cuda.synchronize()
            outs[off] = (dst.cpu().numpy(), p.kernel_name())
    assert outs[0][1] == outs[2][1] == outs[6][1] and "fir_ols_kernel" in outs[0][1] and "32x32" in outs[0][1], [v[1] for v in outs.values()]
    assert np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[0][0], outs[6][0])
    # the fused chain on the same views
    q = synth.biquad_rbj_lowpass()
    kw = dict(dtype=np.float32, lines=lines, max_batch=K)
    res = {}
    for off in (0, 2):
        src = view(off)
        dst = torch.full((n + 16,), float("nan"), dtype=torch.float32, device="cuda")[off:off + n]
        with P.Chain([P.Fir(taps[:256], F, C, **kw), P.Biquad(q, F, C, **kw), P.Gain(0.5, F, C, **kw)]) as ch:
            ch.start()
            ch.process_batch(src, dst, K * F)
            torch.cuda.synchronize()
            res[off] = (dst.cpu().numpy(), ch.kernel_name())
            ch.flush()
    assert res[0][1] == res[2][1] and "chain_fused_kernel" in res[0][1], (res[0][1], res[2][1])
    assert np.array_equal(res[0][0], res[2][0])


@pytest.mark.parametrize("calls", [[769 * 7 + 5, 769 * 4, 300, 769 * 2 + 1], [100000], [769, 1538, 1539]])
@pytest.mark.parametrize("ntaps", [256, 64, 511])
def test_mono_lines_ride_two_tiles_per_transform(calls, ntaps, monkeypatch):
    """One channel: a half-wave's complex sequence carries tile t and tile t + half-the-tiles of the same Line
    (fir_ols32_kernel<..., MONO>).  Odd and even tile counts, calls of one tile (the channel then rides alone),
    state carried from call to call, three Lines: within one ulp of the oracle like every overlap-save launch,
    (the channel-alone form it replaced is an A/B-build variant)."""
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    lines, C = 3, 1
    taps = synth.fir_lowpass_taps(ntaps, fc=0.09, f32_rounded=True)
    total = sum(calls)
    x = np.stack([synth.samples(synth.line_seed(600 + l), 0, total, np.float32).reshape(total, C) for l in range(lines)])
    with P.Fir(taps, max(calls), C, dtype=np.float32, lines=lines, max_batch=1) as p:
        p.start()
        d = torch.from_numpy(x).cuda()
        outs, pos = [], 0
        for n in calls:
            xin = d[:, pos:pos + n, :].contiguous()
            y = torch.full_like(xin, float("nan"))
            p.process_batch(xin, y, n)
            torch.cuda.synchronize()
            assert "32x32" in p.kernel_name(), p.kernel_name()
            outs.append(y)
            pos += n
        got = torch.cat(outs, dim=1).cpu().numpy()
    assert not np.isnan(got).any()
    floor = 2.0 ** -24 * np.abs(taps).sum()
    for l in range(lines):
        want = O.Fir(taps, C).process(x[l].astype(np.float64)).reshape(total, C)
        dd = ulp_diff_f32(got[l], want, floor)
        assert dd.max() <= 1.0, f"line {l}: {dd.max()} ulp at {np.unravel_index(dd.argmax(), dd.shape)}"
        assert np.mean(got[l] != want.astype(np.float32)) < 1e-3


# ---- float64 buffers: what a Go pipe carries (pipe.go:394,437) -------------------------------------------------------
def f64_bound(taps, xmax=1.0):
    """include/pipe_hip.h, PIPE_HIP_PARAM_RELAXED_F64 on a FIR: |y - oracle| <= 64 * 2^-53 * ||h||_1 * max|x|."""
    return 64.0 * 2.0 ** -53 * float(np.abs(taps).sum()) * xmax


def run_batch_f64(taps, x, F, K, lines, relaxed, exact=False):
    C = x.shape[-1]
    with P.Fir(taps, F, C, dtype=np.float64, lines=lines, max_batch=K) as p:
        p.start()
        if relaxed:
            p.set_relaxed_f64(True)
        if exact:
            p.set_exact(True)
        d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        outs, names = [], []
        half = (K // 2) * F
        frames = x.shape[-2]
        for a, b in ((0, half), (half, frames)):   # two launches: the history carries in float64
            xin = d_in[..., a:b, :].contiguous()
            yout = torch.full_like(xin, float("nan"))
            p.process_batch(xin, yout, b - a)
            torch.cuda.synchronize()
            names.append(p.kernel_name())
            outs.append(yout)
        return torch.cat(outs, dim=-2).cpu().numpy(), names


@pytest.mark.parametrize("channels,ntaps", [(2, 256), (2, 64), (4, 511), (3, 256), (1, 128), (2, 1100), (1, 2100)])
def test_float64_buffers_take_the_overlap_save_form_only_when_asked(channels, ntaps, monkeypatch):
    """Without PIPE_HIP_PARAM_RELAXED_F64 a float64 batch is the ordered sum, bit for bit the oracle's; with it the same
    batch takes the overlap-save kernel (float64 loads and stores around the same float64 transform) and stays within
    the bound the header states; PIPE_HIP_PARAM_EXACT wins over the opt-in."""
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    F, K, lines = 1024, 24, 2
    taps = synth.fir_lowpass_taps(ntaps, fc=0.11)   # float64 taps, not rounded to float32: a float64 host's own
    x = np.stack([synth.samples(synth.line_seed(700 + l), 0, K * F * channels, np.float64).reshape(K * F, channels)
                  for l in range(lines)])
    want = np.stack([O.Fir(taps, channels).process(x[l]).reshape(K * F, channels) for l in range(lines)])
    plain, names = run_batch_f64(taps, x, F, K, lines, relaxed=False)
    assert all("fir_mfma_kernel<f64,f64>" in n or "fir_direct_kernel<f64,f64>" in n for n in names), names
    assert np.array_equal(plain, want)
    got, names = run_batch_f64(taps, x, F, K, lines, relaxed=True)
    assert all(n.startswith("fir_ols_kernel<f64,f64,32x32") for n in names), names
    assert all(("partitioned" in n) == (ntaps > 512) for n in names), names
    err = np.abs(got - want).max()
    assert err <= f64_bound(taps), (err, f64_bound(taps), err / (2.0 ** -53 * np.abs(taps).sum()))
    assert not np.array_equal(got, want) or ntaps < 32   # (it IS another arithmetic: the bound is not vacuous)
    pinned, names = run_batch_f64(taps, x, F, K, lines, relaxed=True, exact=True)
    assert all("fir_ols_kernel" not in n for n in names), names
    assert np.array_equal(pinned, want)


def test_float64_chain_staged_with_the_relaxed_forms(monkeypatch):
    """FIR -> biquad -> gain on float64 buffers with PIPE_HIP_PARAM_RELAXED_F64 on the chain: the FIR's overlap-save form
    with float64 results, the biquad's tile form, the gain folded into its store -- against the oracle's chain within the
    sum of the two stages' bounds (the biquad's low-pass gain is <= 1: the FIR's error passes through it unamplified up
    to kappa)."""
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    lines, C, F = 6, 2, 4096 * 6
    taps = synth.fir_lowpass_taps(600)   # (more than 512 taps: up to 512 the chain takes the FUSED kernel, tests/test_gpu_chain_fused.py)
    q = synth.biquad_rbj_lowpass()
    g = 0.7071067811865476
    x = np.stack([synth.samples(synth.line_seed(720 + l), 0, F * C, np.float64).reshape(F, C) for l in range(lines)])
    kw = dict(dtype=np.float64, lines=lines, max_batch=6)
    res = {}
    for relaxed in (False, True):
        with P.Chain([P.Fir(taps, 4096, C, **kw), P.Biquad(q, 4096, C, **kw), P.Gain(g, 4096, C, **kw)]) as p:
            p.start()
            if relaxed:
                p.set_relaxed_f64(True)
            d_in = torch.from_numpy(x).cuda()
            d_out = torch.full_like(d_in, float("nan"))
            p.process_batch(d_in, d_out, F)
            torch.cuda.synchronize()
            res[relaxed] = (d_out.cpu().numpy(), p.kernel_name())
    want = np.stack([O.gain(O.Biquad(q, C).process(O.Fir(taps, C).process(x[l])), g).reshape(F, C) for l in range(lines)])
    assert np.array_equal(res[False][0], want), res[False][1]            # default: bit for bit
    assert "fir_ols_kernel<f64,f64" in res[True][1], res[True][1]
    from tests import _tol
    kap = _tol.kappa(q)
    bound = (256.0 * kap * 2.0 ** -53) * np.abs(want).max() + kap * f64_bound(taps)
    err = np.abs(res[True][0] - want).max()
    assert err <= bound, (err, bound)
