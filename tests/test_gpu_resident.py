"""PIPE_HIP_PARAM_RESIDENT: pipe_hip_process with the next buffer's work queued on the device ahead of its call
(behind a doorbell word the host rings; run.go:198-224's loop moved next to the data, depth per link still
fitting.go:56-60's one buffer).  Everything the plain per-buffer path guarantees must hold through this entry
too: bit for bit the oracle, mutations seen by the next buffer, short buffers, restarts."""
import numpy as np
import pytest

from oracle import oracle as O
from pipe_amd import _lib as L
from pipe_amd import processors as P
from pipe_amd import synth

pytestmark = pytest.mark.gpu

F, C, N = 4096, 2, 256


def stream(seed, buffers, frames=F, channels=C):
    return synth.samples(synth.line_seed(seed), 0, buffers * frames * channels).reshape(buffers, frames, channels)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fir_stream_is_the_oracles_bit_for_bit(dtype):
    taps = synth.fir_lowpass_taps(N)
    x = stream(3, 12)
    with P.Fir(taps, F, C, dtype=dtype) as fir, P.Fir(taps, F, C, dtype=dtype) as plain:
        fir.start()
        plain.start()
        assert fir.set_resident(True)
        ref = O.Fir(taps, C)
        for k in range(12):
            frames = F if k != 7 and k != 11 else (1000 if k == 7 else 17)  # short buffers in the middle and at the end
            xin = x[k, :frames].astype(dtype)
            got = fir.process(xin)
            want = ref.process(x[k, :frames].astype(dtype).astype(np.float64)).reshape(frames, C).astype(dtype)
            assert got.shape == want.shape and np.array_equal(got, want), f"buffer {k}"
            assert np.array_equal(got, plain.process(xin))
        fir.flush()


def test_a_taps_mutation_reaches_the_next_buffer_and_no_other():
    taps = synth.fir_lowpass_taps(N)
    taps2 = taps[::-1].copy() * 0.5
    x = stream(4, 8).astype(np.float32)
    with P.Fir(taps, F, C, dtype=np.float32) as fir:
        fir.start()
        assert fir.set_resident(True)
        ref = O.Fir(taps, C)
        for k in range(8):
            if k == 3:  # the queued launch of buffer 3 was made with the old taps: it is run and dropped
                fir.set_taps(taps2)
                ref.set_taps(taps2)
            got = fir.process(x[k])
            want = ref.process(x[k].astype(np.float64)).reshape(F, C).astype(np.float32)
            assert np.array_equal(got, want), f"buffer {k}"


def test_gain_and_its_mutation():
    x = stream(5, 6).astype(np.float32)
    with P.Gain(0.25, F, C, dtype=np.float32) as g:
        g.start()
        assert g.set_resident(True)
        for k in range(6):
            if k == 4:
                g.set_gain(-3.0)
            want = (x[k].astype(np.float64) * (0.25 if k < 4 else -3.0)).astype(np.float32)
            assert np.array_equal(g.process(x[k]), want)


def test_chain_of_fir_and_gain_restart_and_other_entries():
    import torch
    taps = synth.fir_lowpass_taps(64)
    x = stream(6, 10).astype(np.float32)
    with P.Chain([P.Fir(taps, F, C, dtype=np.float32), P.Gain(0.5, F, C, dtype=np.float32)]) as ch:
        ch.start()
        assert ch.set_resident(True)
        ref = O.Fir(taps, C)
        for k in range(10):
            if k == 4:  # StartFunc in the middle: the queued launch is taken back, state is silence again
                ch.start()
                ref = O.Fir(taps, C)
            if k == 6:  # a device-resident batch call on the same handle between two buffers
                d_in = torch.from_numpy(x[k].reshape(-1)).cuda()
                d_out = torch.empty_like(d_in)
                ch.process_batch(d_in, d_out, F)
                torch.cuda.synchronize()
                got = d_out.cpu().numpy().reshape(F, C)
            else:
                got = ch.process(x[k])
            want = (ref.process(x[k].astype(np.float64)).reshape(F, C) * 0.5).astype(np.float32)
            assert np.array_equal(got, want), f"buffer {k}"
        ch.set_resident(False)
        got = ch.process(x[0])
        want = (ref.process(x[0].astype(np.float64)).reshape(F, C) * 0.5).astype(np.float32)
        assert np.array_equal(got, want)


def test_a_biquad_is_queued_ahead_for_the_calls_that_take_its_tile_form():
    """The tile form writes a series' new state out of place (the halves of a double buffer trade places), so a queued
    launch can be taken back; the ordered recurrence updates its state in place and cannot.  float32 buffers of >= 1024
    frames: queued ahead, bit for bit the plain path's tile form; a short buffer in between and float64 buffers run the
    plain path (ordered, bit for bit the oracle) on the same handle."""
    q = synth.biquad_rbj_lowpass()
    taps = synth.fir_lowpass_taps(64)
    x = stream(9, 8)
    for mk in (lambda dt: P.Biquad(q, F, C, dtype=dt),
               lambda dt: P.Chain([P.Fir(taps, F, C, dtype=dt), P.Biquad(q, F, C, dtype=dt), P.Gain(0.5, F, C, dtype=dt)])):
        with mk(np.float32) as res, mk(np.float32) as plain:
            res.start()
            plain.start()
            assert res.set_resident(True)
            for k in range(8):
                frames = 300 if k == 3 else (2000 if k == 5 else F)   # (300: the ordered form; 2000: another tile call)
                if k == 6:
                    for h in (res, plain):
                        h.set_stage_param(1, L.PARAM_COEFFS, synth.biquad_rbj_lowpass(fc=3000.0)) if hasattr(h, "stages") \
                            else h.set_coeffs(synth.biquad_rbj_lowpass(fc=3000.0))
                xin = x[k, :frames].astype(np.float32)
                assert np.array_equal(res.process(xin), plain.process(xin)), k
        with mk(np.float64) as r64:
            r64.start()
            assert r64.set_resident(True)
            refs = [O.Biquad(q, C)] if not hasattr(r64, "stages") else [O.Fir(taps, C), O.Biquad(q, C)]
            for k in range(3):
                y = x[k]
                for r in refs:
                    y = np.asarray(r.process(y)).reshape(F, C)
                if hasattr(r64, "stages"):
                    y = y * 0.5
                assert np.array_equal(r64.process(x[k]), y), k


def test_stages_with_in_place_state_and_many_lines_are_turned_away():
    three = np.vstack([synth.biquad_rbj_lowpass(fc=f) for f in (500.0, 1500.0, 4000.0)])
    with P.Biquad(three, F, C, dtype=np.float32) as bq:   # (the tile form holds two sections)
        bq.start()
        with pytest.raises(L.PipeHipError):
            bq.set_resident(True)
    proto = synth.resampler_proto(160, 147, 24)
    with P.Resampler(proto, 24, 160, 147, F, C, dtype=np.float32) as rs:
        rs.start()
        with pytest.raises(L.PipeHipError):
            rs.set_resident(True)
    with P.Gain(0.5, 4096, 8, dtype=np.float32, lines=512) as g:  # 64 MiB a buffer: not the zero-copy path
        g.start()
        with pytest.raises(L.PipeHipError):
            g.set_resident(True)


def test_a_handle_destroyed_with_work_queued_leaves_nothing_waiting():
    x = stream(7, 2).astype(np.float32)
    for _ in range(3):
        g = P.Gain(2.0, F, C, dtype=np.float32)
        g.start()
        assert g.set_resident(True)
        assert np.array_equal(g.process(x[0]), (x[0] * 2.0).astype(np.float32))
        g.close()  # a launch for the next buffer is queued behind the doorbell right now
    with P.Gain(2.0, F, C, dtype=np.float32) as g2:  # the device still answers
        g2.start()
        assert np.array_equal(g2.process(x[1]), (x[1] * 2.0).astype(np.float32))


def test_queued_work_does_not_hold_a_device_wide_synchronisation_for_ever():
    """A queue waiting for a doorbell holds up hipDeviceSynchronize; the library's watchdog takes queued work back
    when no call has come for the idle limit (here 60 ms), and the stream goes on bit for bit afterwards."""
    import time
    import torch
    taps = synth.fir_lowpass_taps(N)
    x = stream(8, 4).astype(np.float32)
    with P.Fir(taps, F, C, dtype=np.float32) as fir:
        fir.start()
        fir._set_param(L.PARAM_RESIDENT, [60.0])
        ref = O.Fir(taps, C)
        for k in range(4):
            got = fir.process(x[k])
            want = ref.process(x[k].astype(np.float64)).reshape(F, C).astype(np.float32)
            assert np.array_equal(got, want), f"buffer {k}"
            if k == 1:
                t0 = time.perf_counter()
                torch.cuda.synchronize()  # the next buffer's work is queued behind the doorbell right now
                assert time.perf_counter() - t0 < 2.0
            if k == 2:
                time.sleep(0.2)  # the watchdog has taken the queued work back: the next call queues afresh



def test_one_doorbell_per_device_and_the_second_handle_stays_on_the_plain_path():
    """A queue that waits for a doorbell costs every other waiting queue of the process tens of microseconds
    (scripts/micro/queue_independence.hip), so ONE handle per device holds the doorbell; the next one that asks is
    answered PIPE_HIP_EBUSY and runs the plain path -- same bits -- until the holder gives it back."""
    x = stream(11, 6).astype(np.float32)
    with P.Gain(0.5, F, C, dtype=np.float32) as a, P.Gain(0.25, F, C, dtype=np.float32) as b:
        a.start()
        b.start()
        assert a.set_resident(True) is True
        assert a.set_resident(True) is True          # asking twice is fine
        assert b.set_resident(True) is False         # PIPE_HIP_EBUSY
        assert a.resident_info()[0] and not b.resident_info()[0]
        for k in range(3):
            assert np.array_equal(a.process(x[k]), (x[k] * 0.5).astype(np.float32))
            assert np.array_equal(b.process(x[k]), (x[k] * 0.25).astype(np.float32))
        assert a.set_resident(False) is True         # the doorbell is free again
        assert b.set_resident(True) is True
        for k in range(3, 6):
            assert np.array_equal(a.process(x[k]), (x[k] * 0.5).astype(np.float32))
            assert np.array_equal(b.process(x[k]), (x[k] * 0.25).astype(np.float32))
    with P.Gain(2.0, F, C, dtype=np.float32) as c:   # ... and after its holder was destroyed
        c.start()
        assert c.set_resident(True) is True


def test_dropped_launches_are_counted():
    import time
    x = stream(12, 4).astype(np.float32)
    with P.Gain(0.5, F, C, dtype=np.float32) as g:
        g.start()
        assert g.set_resident(True, idle_ms=40)
        g.process(x[0])
        time.sleep(0.25)                              # no call for 40 ms: the watchdog rings, the work runs on stale input
        held, by_watchdog, by_entry = g.resident_info()
        assert held and by_watchdog == 1 and by_entry == 0
        assert np.array_equal(g.process(x[1]), (x[1] * 0.5).astype(np.float32))
        g.set_gain(2.0)                               # a mutation finds queued work: one more dropped launch
        assert g.resident_info()[2] == 1
        assert np.array_equal(g.process(x[2]), (x[2] * 2.0).astype(np.float32))


def test_a_stage_that_grows_its_scratch_while_work_is_being_queued():
    """The tile biquad sizes its look-back area by the call: 2000 frames first, then full buffers -- the larger area is
    allocated while the wait for the next doorbell is already queued; the old one must not be freed there (a free
    waits for every queue of the device, this one included: ADVICE r4)."""
    q = synth.biquad_rbj_lowpass()
    x = stream(13, 6).astype(np.float32)
    with P.Biquad(q, 4 * F, C, dtype=np.float32) as res, P.Biquad(q, 4 * F, C, dtype=np.float32) as plain:
        res.start()
        plain.start()
        assert res.set_resident(True)
        big = np.concatenate([x[2], x[3], x[4], x[5]])
        for xin in (x[0][:2000], x[1][:2000], x[1], big, big, x[0][:1024]):
            assert np.array_equal(res.process(xin), plain.process(xin))


def test_a_queued_launch_that_fails_on_the_device_ends_the_run_loudly():
    """PIPE_HIP_PARAM_DEBUG makes the next tile launch give up on a predecessor tile.  On the plain path the call is
    run again through the ordered recurrence; queued ahead, its successor is already queued on its state, so the call
    answers PIPE_HIP_EHIP (pipe.go:438-440: a ProcessFunc error ends the run), further calls PIPE_HIP_ESTATE, and the
    next StartFunc begins from silence -- bit for bit a fresh handle."""
    q = synth.biquad_rbj_lowpass()
    FB = 4 * F                                          # four tiles of 4096 frames a call: tile 1 has a predecessor
    x = stream(14, 4, frames=FB).astype(np.float32)
    with P.Biquad(q, FB, C, dtype=np.float32) as res, P.Biquad(q, FB, C, dtype=np.float32) as fresh:
        res.start()
        fresh.start()
        assert res.set_resident(True)
        res.process(x[0])
        assert res.kernel_name().startswith("biquad_tile_kernel"), res.kernel_name()
        res._set_param(L.PARAM_DEBUG, [1.0, 20000.0])   # (the queued launch is dropped; the next one will fail)
        with pytest.raises(L.PipeHipError) as e:
            res.process(x[1])
        assert e.value.status == L.EHIP
        with pytest.raises(L.PipeHipError) as e:
            res.process(x[2])
        assert e.value.status == L.ESTATE
        res.start()
        for k in range(4):
            assert np.array_equal(res.process(x[k]), fresh.process(x[k])), k
