"""Every kernel family the DEFAULT build of the library can dispatch is launched here, on the default build, and its
pipe_hip_kernel_name asserted -- so the 75 tests that force a variant through an A/B-only switch (and skip against the
library that ships) are never the only cover of a form a user can reach.  tests/test_abi_surface.py checks, on CPU,
that FAMILIES names every family the sources can report and that nothing in this file depends on an A/B switch."""
import numpy as np
import pytest

from pipe_amd import synth

F = 4096
Q1 = synth.biquad_rbj_lowpass()
Q3 = np.vstack([synth.biquad_rbj_lowpass(fc=f) for f in (500.0, 1500.0, 4000.0)])
TAPS = synth.fir_lowpass_taps(256, f32_rounded=True)


def _batch(p, lines, frames, channels, dtype):
    import torch
    x = torch.zeros(lines * frames * channels, dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
    y = torch.empty_like(x)
    p.process_batch(x, y, frames)
    torch.cuda.synchronize()
    return p.kernel_name()


def _per_buffer(p, frames, channels, dtype):
    p.process(np.zeros((frames, channels), dtype))
    return p.kernel_name()


def _gain(P):
    with P.Gain(0.5, F, 2, dtype=np.float32) as p:
        p.start()
        return _per_buffer(p, F, 2, np.float32)


def _mix(P):
    with P.Mix(2, F, 2, dtype=np.float32) as p:
        p.start()
        p.process([np.zeros((F, 2), np.float32)] * 2)
        return p.kernel_name()


def _fir_direct(P):
    with P.Fir(TAPS, F, 2, dtype=np.float32) as p:
        p.start()
        return _per_buffer(p, F, 2, np.float32)


def _fir_mfma(P):
    with P.Fir(TAPS, F, 2, dtype=np.float64, max_batch=16) as p:   # float64 results: the ordered form, 16 buffers a call
        p.start()
        return _batch(p, 1, 16 * F, 2, np.float64)


def _fir_ols(P):
    with P.Fir(TAPS, F, 2, dtype=np.float32, lines=8, max_batch=256) as p:
        p.start()
        return _batch(p, 8, 256 * F, 2, np.float32)


def _chain_fused(P):
    kw = dict(dtype=np.float32, lines=96)
    with P.Chain([P.Fir(TAPS, F, 8, **kw), P.Biquad(Q1, F, 8, **kw), P.Gain(0.5, F, 8, **kw)]) as p:
        p.start()
        return _batch(p, 96, F, 8, np.float32)


def _biquad_register(P):
    with P.Biquad(Q1, 512, 2, dtype=np.float64, lines=300) as p:
        p.start()
        return _batch(p, 300, 512, 2, np.float64)


def _biquad_lds(P):
    with P.Biquad(Q1, F, 2, dtype=np.float64) as p:
        p.start()
        return _per_buffer(p, F, 2, np.float64)


def _biquad_lds_sp(P):
    with P.Biquad(Q3, F, 2, dtype=np.float64) as p:
        p.start()
        return _per_buffer(p, F, 2, np.float64)


def _biquad_tile(P):
    with P.Biquad(Q1, F, 2, dtype=np.float32) as p:
        p.start()
        return _per_buffer(p, F, 2, np.float32)


def _biquad_lane_walk(P):
    with P.Biquad(Q1, F, 16, dtype=np.float32, lines=64, max_batch=8) as p:
        p.start()
        return _batch(p, 64, 8 * F, 16, np.float32)


def _resampler(channels, taps_per_phase):
    def run(P):
        proto = synth.resampler_proto(160, 147, taps_per_phase)
        with P.Resampler(proto, taps_per_phase, 160, 147, F, channels, dtype=np.float32) as p:
            p.start()
            return _per_buffer(p, 3000, channels, np.float32)
    return run


def _resampler_pair(P):
    # 161 / 147: a group of the wave kernel would be 161 waves (161 is odd) -- the workgroup-tiled pair kernel takes it
    proto = synth.resampler_proto(161, 147, 24)
    with P.Resampler(proto, 24, 161, 147, F, 2, dtype=np.float32) as p:
        p.start()
        return _per_buffer(p, 3000, 2, np.float32)


def _resampler_rows(P):
    # a resident 8-channel stream of 40 pipe buffers: 64 blocks of 16 rows (a period of the phase pattern each) and more
    import torch
    proto = synth.resampler_proto(160, 147, 24)
    n = 40 * F
    with P.Resampler(proto, 24, 160, 147, F, 8, dtype=np.float32, max_batch=40) as p:
        p.start()
        d_in = torch.zeros(n * 8, dtype=torch.float32, device="cuda")
        cap = -(-n * 160 // 147) + 1
        d_out = torch.empty(cap * 8, dtype=torch.float32, device="cuda")
        p.resample_batch(d_in, n, d_out, cap)
        torch.cuda.synchronize()
        return p.kernel_name()


# family (the kernel name up to its template arguments) -> (how to reach it on the default build, what the name must start with)
FAMILIES = {
    "gain_kernel": (_gain, "gain_kernel<f32,f32>"),
    "mix_kernel": (_mix, "mix_kernel<f32>"),
    "fir_direct_kernel": (_fir_direct, "fir_direct_kernel<"),
    "fir_mfma_kernel": (_fir_mfma, "fir_mfma_kernel<f64,f64>"),
    "fir_ols_kernel": (_fir_ols, "fir_ols_kernel<f32,f32,32x32>"),
    "chain_fused_kernel": (_chain_fused, "chain_fused_kernel<f32,f32,fir+biquad1+gain>"),
    "biquad_kernel": (_biquad_register, "biquad_kernel<f64,f64>"),
    "biquad_lds_kernel": (_biquad_lds, "biquad_lds_kernel<f64,f64>"),
    "biquad_lds_sp_kernel": (_biquad_lds_sp, "biquad_lds_sp_kernel<f64,f64>"),
    "biquad_tile_kernel": (_biquad_tile, "biquad_tile_kernel<f32,f32,segmented>"),
    "biquad_kernel<segmented>": (_biquad_lane_walk, "biquad_kernel<f32,f32,segmented>"),
    "resample_wave_kernel": (_resampler(2, 24), "resample_wave_kernel<f32,f32>"),
    "resample_rows_kernel": (_resampler_rows, "resample_rows_kernel<f32,f32>"),
    "resample_pair_kernel": (_resampler_pair, "resample_pair_kernel<f32,f32>"),
    "resample_tiled_kernel": (_resampler(8, 24), "resample_tiled_kernel<f32,f32"),
    "resample_kernel": (_resampler(64, 48), "resample_kernel<f32,f32>"),
}


@pytest.mark.gpu
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_the_default_build_dispatches_the_family(family):
    from pipe_amd import processors as P
    run, prefix = FAMILIES[family]
    name = run(P)
    assert name.startswith(prefix), (family, name)
