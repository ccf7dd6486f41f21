"""The row form of the resampler (pipe_amd/csrc/resampler_rows.hip: a wave's lanes are the same output of different
periods of the phase pattern, taps in scalar registers, the window in vector registers, a workgroup's stretch of the
stream through LDS) against the CPU oracle, bit for bit: every ratio / tap count / sample type / channel count the form
takes, streams cut into calls of unequal length (the call's first row starts mid-period, its window reaches into the
history), several Lines, calls too short for it (another kernel keeps them).  The threshold knob is lowered so that
sizes the oracle finishes in seconds take the form."""
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


def _tt(dtype):
    return torch.float32 if dtype == np.float32 else torch.float64


def _stream(p, ref_list, x, lens, cap, lines, C, dtype):
    """x: [lines][frames][C]; feeds the calls of `lens` frames through resample_batch, returns the kernel names."""
    names = []
    pos = 0
    for n in lens:
        xin = np.ascontiguousarray(x[:, pos:pos + n, :])
        d_in = torch.from_numpy(xin).cuda()
        d_out = torch.full((lines * cap * C,), float("nan"), dtype=_tt(dtype), device="cuda")
        n_out = p.resample_batch(d_in, n, d_out, cap)
        torch.cuda.synchronize()
        names.append(p.kernel_name())
        got = d_out.cpu().numpy().reshape(lines, cap, C)
        for l in range(lines):
            want = ref_list[l].process(xin[l].astype(np.float64)).reshape(-1, C).astype(dtype)
            assert want.shape[0] == n_out, (pos, n, l)
            assert np.array_equal(got[l, :n_out], want), (pos, n, l, names[-1])
            assert np.isnan(got[l, n_out:]).all(), (pos, n, l)   # nothing written past the call's outputs
        pos += n
    return names


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("up,down,T,C", [(160, 147, 24, 2), (147, 160, 24, 2), (2, 1, 16, 2), (1, 2, 32, 2), (3, 2, 12, 2), (160, 147, 8, 2),
                                         (320, 147, 24, 2), (7, 5, 24, 2), (160, 147, 24, 8), (147, 160, 16, 4), (2, 1, 24, 16), (160, 147, 24, 6), (3, 2, 16, 12),
                                         # upsamplers by a large factor: a row of 144 / up periods would hold fewer than the T - 1 frames a window's
                                         # refill reads out of the row above: rows of ceil((T - 1) / down) periods
                                         (8, 1, 24, 4), (16, 1, 12, 8), (160, 3, 24, 4), (8, 1, 24, 2)])
def test_rows_form_streams_bit_exact(monkeypatch, dtype, up, down, T, C):
    monkeypatch.setenv("PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS", "1")
    lines = 2
    proto = synth.resampler_proto(up, down, T)
    big = max(up, down)
    row_in = down * max(1 if big >= 144 else 144 // big, -(-(T - 1) // down))
    rpb = 64 // (C // 2)   # rows of a workgroup's block (a lane holds a pair of channels)
    # calls: 6 workgroups of rows; a short one (another kernel's); 3.2 workgroups starting mid-row; 3.1 more
    lens = [6 * rpb * row_in + 5, row_in // 2 + 1, 3 * rpb * row_in + rpb // 5 * row_in + 11, 3 * rpb * row_in + row_in]
    total = sum(lens)
    x = np.stack([synth.samples(synth.line_seed(40 + l), 0, total * C).reshape(total, C) for l in range(lines)]).astype(dtype)
    refs = [O.Resampler(proto, T, up, down, C) for _ in range(lines)]
    cap = -(-max(lens) * up // down) + 1
    with P.Resampler(proto, T, up, down, 4096, C, dtype=dtype, lines=lines, max_batch=max(lens) // 4096 + 1) as p:
        p.start()
        names = _stream(p, refs, x, lens, cap, lines, C, dtype)
    # float32 streams of up to 24 taps per phase take the form (float64: the rows' float64 samples do not fit a CU's LDS;
    # 32 taps: the window alone is 128 registers): everybody else's values were checked just the same
    takes = dtype == np.float32 and T <= 24
    assert names[0].startswith("resample_rows_kernel") == takes, names
    assert not names[1].startswith("resample_rows_kernel"), names   # half a row: not even one block of rows
    assert names[2].startswith("resample_rows_kernel") == takes, names
    assert names[3].startswith("resample_rows_kernel") == takes, names


def test_rows_form_is_the_default_for_long_streams_of_four_channels_and_more():
    """Default thresholds: 6 channels and more take the rows from the first block (one 4096-frame pipe buffer a call: 11.7 us
    against the tiled kernel's 16.7 -- since round 6, profiles/r06_dispatch_audit.txt); 4 channels keep one pipe buffer
    on the tiled kernel (11.8 against 13.0 us) and take the rows from 64 blocks; a stereo stream keeps the wave kernel
    (level with the rows: not switched)."""
    up, down, T, F = 160, 147, 24, 4096
    proto = synth.resampler_proto(up, down, T)
    # (64 blocks of 16 rows of 8 channels need 64 x 16 x 147 = 150 528 frames; of 32 rows of 4 channels twice that)
    for C, K, short_kernel, long_kernel in ((8, 40, "resample_rows_kernel", "resample_rows_kernel"), (4, 80, "resample_tiled_kernel", "resample_rows_kernel"),
                                            (2, 40, "resample_wave_kernel", "resample_wave_kernel")):
        n = K * F
        x = synth.samples(synth.line_seed(51), 0, n * C).reshape(n, C).astype(np.float32)
        cap = -(-n * up // down) + 1
        ref = O.Resampler(proto, T, up, down, C)
        with P.Resampler(proto, T, up, down, F, C, dtype=np.float32, max_batch=K) as p:
            p.start()
            got = p.process(x[:F], out_cap_frames=-(-F * up // down) + 1)
            assert p.kernel_name().startswith(short_kernel), (C, p.kernel_name())
            assert np.array_equal(got, ref.process(x[:F].astype(np.float64)).reshape(-1, C).astype(np.float32))
            d_in = torch.from_numpy(x).cuda()
            d_out = torch.full((cap * C,), float("nan"), dtype=torch.float32, device="cuda")
            n_out = p.resample_batch(d_in, n, d_out, cap)
            torch.cuda.synchronize()
            assert p.kernel_name().startswith(long_kernel), (C, p.kernel_name())
            want = ref.process(x.astype(np.float64)).reshape(-1, C).astype(np.float32)
            assert n_out == want.shape[0]
            assert np.array_equal(d_out.cpu().numpy()[: n_out * C].reshape(n_out, C), want)


@pytest.mark.parametrize("C", [2, 8])
def test_rows_form_samples_that_are_not_finite(monkeypatch, C):
    """Samples that are not finite: NaNs and infinities come out exactly where the oracle puts them (the form multiplies
    nothing by a zero it was not given), everything else bit for bit."""
    monkeypatch.setenv("PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS", "1")
    up, down, T = 160, 147, 24
    proto = synth.resampler_proto(up, down, T)
    rpb = 64 // (C // 2)
    n = 5 * rpb * down + 17
    x = synth.samples(synth.line_seed(60), 0, n * C).reshape(n, C).astype(np.float32)
    x[100, 0] = np.inf               # first block
    x[2 * rpb * down + 5, C - 1] = np.nan      # third block; the blocks between them are finite
    x[n - 3, 0] = -np.inf            # last block
    cap = -(-n * up // down) + 1
    with P.Resampler(proto, T, up, down, 4096, C, dtype=np.float32, max_batch=n // 4096 + 1) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.full((cap * C,), 7.0, dtype=torch.float32, device="cuda")
        n_out = p.resample_batch(d_in, n, d_out, cap)
        torch.cuda.synchronize()
        assert p.kernel_name().startswith("resample_rows_kernel")
    with np.errstate(invalid="ignore"):
        want = O.Resampler(proto, T, up, down, C).process(x.astype(np.float64)).reshape(-1, C).astype(np.float32)
    got = d_out.cpu().numpy()[: n_out * C].reshape(n_out, C)
    assert n_out == want.shape[0]
    assert np.isnan(want).any() and np.isinf(want).any()
    assert np.array_equal(got, want, equal_nan=True)


def test_rows_form_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS", "-1")
    up, down, T, C, F, K = 160, 147, 24, 8, 4096, 40
    proto = synth.resampler_proto(up, down, T)
    n = K * F
    d_in = torch.zeros(n * C, dtype=torch.float32, device="cuda")
    cap = -(-n * up // down) + 1
    d_out = torch.empty(cap * C, dtype=torch.float32, device="cuda")
    with P.Resampler(proto, T, up, down, F, C, dtype=np.float32, max_batch=K) as p:
        p.start()
        p.resample_batch(d_in, n, d_out, cap)
        torch.cuda.synchronize()
        assert p.kernel_name().startswith("resample_tiled_kernel")


def test_stereo_streams_the_wave_kernel_does_not_take_come_to_the_rows_by_default():
    """161 / 147: the wave kernel's group would be 161 waves wide; a long stereo stream of it takes the row form without
    any knob (a short call keeps the workgroup-tiled pair kernel)."""
    up, down, T, C, F, K = 161, 147, 24, 2, 4096, 160
    proto = synth.resampler_proto(up, down, T)
    n = K * F
    x = synth.samples(synth.line_seed(52), 0, n * C).reshape(n, C).astype(np.float32)
    cap = -(-n * up // down) + 1
    ref = O.Resampler(proto, T, up, down, C)
    with P.Resampler(proto, T, up, down, F, C, dtype=np.float32, max_batch=K) as p:
        p.start()
        got = p.process(x[:F], out_cap_frames=-(-F * up // down) + 1)
        assert p.kernel_name().startswith("resample_pair_kernel"), p.kernel_name()
        assert np.array_equal(got, ref.process(x[:F].astype(np.float64)).reshape(-1, C).astype(np.float32))
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.full((cap * C,), float("nan"), dtype=torch.float32, device="cuda")
        n_out = p.resample_batch(d_in, n, d_out, cap)
        torch.cuda.synchronize()
        assert p.kernel_name().startswith("resample_rows_kernel"), p.kernel_name()
        want = ref.process(x.astype(np.float64)).reshape(-1, C).astype(np.float32)
        assert n_out == want.shape[0]
        assert np.array_equal(d_out.cpu().numpy()[: n_out * C].reshape(n_out, C), want)


@pytest.mark.parametrize("C", [2, 8])
@pytest.mark.parametrize("off_in,off_out", [(1, 0), (0, 1), (3, 5)])
def test_rows_form_buffers_aligned_to_an_element_only(monkeypatch, C, off_in, off_out):
    """Device buffers that start at an odd element of their allocation (4-byte aligned, not 8 or 16)."""
    monkeypatch.setenv("PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS", "1")
    up, down, T = 160, 147, 24
    proto = synth.resampler_proto(up, down, T)
    rpb = 64 // (C // 2)
    n = 3 * rpb * down + 29
    x = synth.samples(synth.line_seed(61), 0, n * C).reshape(n, C).astype(np.float32)
    cap = -(-n * up // down) + 1
    with P.Resampler(proto, T, up, down, 4096, C, dtype=np.float32, max_batch=n // 4096 + 1) as p:
        p.start()
        big_in = torch.zeros(n * C + 8, dtype=torch.float32, device="cuda")
        big_in[off_in:off_in + n * C] = torch.from_numpy(x.ravel()).cuda()
        big_out = torch.full((cap * C + 16,), float("nan"), dtype=torch.float32, device="cuda")
        d_in = big_in[off_in:off_in + n * C]
        d_out = big_out[off_out:off_out + cap * C]
        n_out = p.resample_batch(d_in, n, d_out, cap)
        torch.cuda.synchronize()
        assert p.kernel_name().startswith("resample_rows_kernel")
    want = O.Resampler(proto, T, up, down, C).process(x.astype(np.float64)).reshape(-1, C).astype(np.float32)
    assert n_out == want.shape[0]
    got = big_out.cpu().numpy()
    assert np.array_equal(got[off_out:off_out + n_out * C].reshape(n_out, C), want)
    assert np.isnan(got[:off_out]).all() and np.isnan(got[off_out + n_out * C:]).all()   # nothing outside the call's outputs
