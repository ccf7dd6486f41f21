"""The C++ host-side mirror of pipe.Run / pipe.New+Start+Wait (pipe_amd/csrc/host)
against the reference's own known answers -- the same cases as
tests/test_oracle_pipe.py, so product loop and oracle loop are both pinned to
pipe_test.go / mock_test.go -- and, on the GPU, HIP Processors inside the loop
against the oracle's loop."""
import numpy as np
import pytest

from oracle import oracle as O
from pipe_amd import host as H
from pipe_amd import synth

BUF = 512  # pipe_test.go:17
MODES = [H.MODE_RUN, H.MODE_ASYNC]


def mock_line(limit, channels=1, procs=1, discard=True, value=0.0, **kw):
    return H.Line(limit=limit, channels=channels, value=value, discard=discard,
                  procs=[H.Proc(H.PROC_MOCK) for _ in range(procs)], **kw)


def assert_line(res, messages, samples):
    for c in [res.source, *res.procs, res.sink]:
        assert (c.messages, c.samples) == (messages, samples)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("limit,messages", [(1040, 3), (1640, 4), (3048, 6), (4096, 8)])
def test_short_last_buffer_counts(mode, limit, messages):
    err, res = H.run(BUF, [mock_line(limit)], mode)  # pipe_test.go:337,363,394,399,404
    assert not err.failed, err.message
    assert_line(res[0], messages, limit)
    assert res[0].source.flushed and res[0].procs[0].flushed and res[0].sink.flushed


@pytest.mark.parametrize("mode", MODES)
def test_three_lines(mode):
    err, res = H.run(BUF, [mock_line(3048), mock_line(1640), mock_line(4096)], mode)  # pipe_test.go:385-436
    assert not err.failed
    assert_line(res[0], 6, 3048)
    assert_line(res[1], 4, 1640)
    assert_line(res[2], 8, 4096)


@pytest.mark.parametrize("mode", MODES)
def test_simple_pipe_862_buffers(mode):
    err, res = H.run(BUF, [mock_line(862 * BUF, channels=2)], mode)  # TestSimplePipe pipe_test.go:82-106
    assert not err.failed
    assert_line(res[0], 862, 862 * BUF)


@pytest.mark.parametrize("mode", MODES)
def test_restart_doubles_sink_counts(mode):
    err, res = H.run(BUF, [mock_line(862 * BUF, channels=2, procs=0)], mode, runs=2)  # TestReset pipe_test.go:108-131
    assert not err.failed
    assert (res[0].sink.messages, res[0].sink.samples) == (2 * 862, 2 * 862 * BUF)


@pytest.mark.parametrize("limit,value,calls", [(11, 1.0, 3), (2500, 2.0, 500)])
def test_mock_source_calls_and_constant_fill(limit, value, calls):
    err, res = H.run(5, [mock_line(limit, channels=2, procs=1, value=value, discard=False)])  # mock_test.go:69-92
    assert not err.failed
    assert (res[0].source.messages, res[0].source.samples) == (calls, limit)
    assert res[0].values.size == limit * 2 and np.all(res[0].values == value)


@pytest.mark.parametrize("data", [[1, 1, 1, 1], [1, 1, 1, 1, 2, 2, 2, 2]])
def test_mock_processor_identity(data):
    x = np.array(data, dtype=np.float64)  # mock_test.go:133-146
    line = H.Line(limit=x.size, channels=1, src_kind=H.SRC_ARRAY, data=x, procs=[H.Proc(H.PROC_MOCK)], discard=False)
    err, res = H.run(x.size, [line])
    assert not err.failed and np.array_equal(res[0].values, x)


def test_hook_order_two_lines_processor_start_error():
    l1 = mock_line(1040, discard=False)  # pipe_test.go:265-309
    l2 = H.Line(limit=1040, channels=1, procs=[H.Proc(H.PROC_MOCK, err_on_start=True)], discard=False)
    err, res = H.run(BUF, [l1, l2])
    assert err.failed and err.is_mock_error and "error starting" in err.message
    a, b = res
    assert (a.source.started, a.procs[0].started, a.sink.started) == (True, True, True)
    assert (a.source.flushed, a.procs[0].flushed, a.sink.flushed) == (True, True, True)
    assert (b.source.started, b.procs[0].started, b.sink.started) == (True, True, False)
    assert (b.source.flushed, b.procs[0].flushed, b.sink.flushed) == (True, False, False)


@pytest.mark.parametrize("mode", MODES)
def test_processor_error_propagates_and_everything_is_flushed(mode):
    line = H.Line(limit=1040, channels=1, procs=[H.Proc(H.PROC_MOCK, err_on_call=True)])  # pipe_test.go:437-457
    err, res = H.run(BUF, [line], mode)
    assert err.failed and err.is_mock_error and "error running" in err.message
    r = res[0]
    assert r.source.flushed and r.procs[0].flushed and r.sink.flushed


@pytest.mark.parametrize("which", ["source", "processor", "sink"])
def test_line_binding_fail(which):
    line = mock_line(1040)  # TestLineBindingFail pipe_test.go:21-80
    if which == "source":
        line.src_err_on_make = True
    elif which == "processor":
        line.procs[0].err_on_make = True
    else:
        line.sink_err_on_make = True
    err, _ = H.run(BUF, [line], H.MODE_ASYNC)
    assert err.failed and err.is_mock_error and err.is_bind_error and err.message.startswith(which)


def test_host_loop_equals_oracle_loop_on_counts_and_values():
    # product loop (C++) and oracle loop (C) on the same synthetic 2-channel stream
    frames = 7 * BUF + 77
    hl = H.Line(limit=frames, channels=2, src_kind=H.SRC_SYNTH, seed=synth.line_seed(1),
                procs=[H.Proc(H.PROC_MOCK), H.Proc(H.PROC_MOCK)], discard=False)
    ol = O.Line(limit=frames, channels=2, src_kind=O.SRC_SYNTH, seed=synth.line_seed(1),
                procs=[O.Proc(O.PROC_COPY), O.Proc(O.PROC_COPY)], discard=False)
    herr, hres = H.run(BUF, [hl])
    oerr, ores = O.run_lines(BUF, [ol])
    assert not herr.failed and oerr.ok
    assert np.array_equal(hres[0].values, ores[0].values)
    assert (hres[0].sink.messages, hres[0].sink.samples) == (ores[0].sink.messages, ores[0].sink.samples)


# ------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_hip_copy_in_the_loop_config1(mode):
    # BASELINE configs[0]: mock.Source -> Processor(gain 1.0) -> mock.Sink, 512-frame buffers
    err, res = H.run(BUF, [H.Line(limit=862 * BUF, channels=2, value=1.0, discard=True,
                                   procs=[H.Proc(H.PROC_HIP_COPY)])], mode)
    assert not err.failed, err.message
    assert_line(res[0], 862, 862 * BUF)
    err, res = H.run(BUF, [H.Line(limit=1040, channels=2, value=2.0, discard=False,
                                   procs=[H.Proc(H.PROC_HIP_COPY)])], mode)
    assert not err.failed
    assert_line(res[0], 3, 1040)
    assert res[0].values.size == 2080 and np.all(res[0].values == 2.0)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_hip_fir_biquad_gain_lines_equal_oracle_loop(mode):
    C_, frames = 2, 9 * BUF + 200
    taps = synth.fir_lowpass_taps(256)
    sos = synth.biquad_rbj_lowpass()
    hlines, olines = [], []
    for i in range(3):  # three Lines round-robin / concurrently, different lengths
        n = frames - i * 300
        hlines.append(H.Line(limit=n, channels=C_, src_kind=H.SRC_SYNTH, seed=synth.line_seed(i), discard=False,
                             procs=[H.Proc(H.PROC_HIP_FIR, taps), H.Proc(H.PROC_HIP_BIQUAD, sos),
                                    H.Proc(H.PROC_HIP_GAIN, [0.5])]))
        olines.append(O.Line(limit=n, channels=C_, src_kind=O.SRC_SYNTH, seed=synth.line_seed(i), discard=False,
                             procs=[O.Proc(O.PROC_FIR, taps), O.Proc(O.PROC_BIQUAD, sos), O.Proc(O.PROC_GAIN, [0.5])]))
    herr, hres = H.run(BUF, hlines, mode)
    oerr, ores = O.run_lines(BUF, olines)
    assert not herr.failed, herr.message
    for h, o in zip(hres, ores):
        assert (h.sink.messages, h.sink.samples) == (o.sink.messages, o.sink.samples)
        assert np.array_equal(h.values, o.values)  # float64 buffers: bit-exact


@pytest.mark.gpu
def test_hip_fused_chain_equals_separate_stages_and_oracle():
    C_, frames = 8, 3 * BUF + 5
    taps = synth.fir_lowpass_taps(64)
    sos = synth.biquad_rbj_lowpass()
    src = dict(limit=frames, channels=C_, src_kind=H.SRC_SYNTH, seed=synth.line_seed(4), discard=False)
    fused = H.Line(procs=[H.Proc(H.PROC_HIP_CHAIN, H.chain_params(taps, sos, 0.25))], **src)
    herr, hres = H.run(BUF, [fused])
    ol = O.Line(limit=frames, channels=C_, src_kind=O.SRC_SYNTH, seed=synth.line_seed(4), discard=False,
                procs=[O.Proc(O.PROC_FIR, taps), O.Proc(O.PROC_BIQUAD, sos), O.Proc(O.PROC_GAIN, [0.25])])
    oerr, ores = O.run_lines(BUF, [ol])
    assert not herr.failed, herr.message
    assert np.array_equal(hres[0].values, ores[0].values)


@pytest.mark.gpu
def test_mutation_reaches_hip_handle_through_the_message():
    # an initializer mutation (pipe.go:205-206) rides the first Message down the Line
    # and is applied in the processing thread before ProcessFunc (pipe.go:433)
    line = H.Line(limit=1040, channels=2, value=1.0, discard=False,
                  procs=[H.Proc(H.PROC_HIP_GAIN, [1.0], mutate_gain=0.25)])
    err, res = H.run(BUF, [line], H.MODE_ASYNC)
    assert not err.failed, err.message
    assert np.all(res[0].values == 0.25)


@pytest.mark.gpu
def test_hip_processor_error_surfaces_as_run_error():
    line = H.Line(limit=1040, channels=1, procs=[H.Proc(H.PROC_HIP_GAIN, [1.0], err_on_call=True)])
    err, res = H.run(BUF, [line])
    assert err.failed and err.is_mock_error
    assert res[0].source.flushed and res[0].procs[0].flushed and res[0].sink.flushed


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_the_host_loop_with_every_stage_sharing_the_doorbell_queue(mode):
    """The same host-loop tests with every HIP stage in the device's SHARED doorbell queue (PIPE_HIP_PARAM_RESIDENT_SHARED)
    -- in the synchronous mode it is made for, and in the asynchronous one (a thread per component: calls in any order
    and concurrent; they serialise on the queue's lock and ring each other's work: slower, and the same bits)."""
    from tests._child import run_child
    run_child(f"""
        import tests.test_host_pipe as T
        T.test_hip_copy_in_the_loop_config1({mode})
        for _ in range(12):  # (round 6: one run in three hung in the async mode -- a StartFunc's hipStreamSynchronize on the
            # SHARED stream behind another handle's parked doorbell, profiles/r06_shared_queue_async_hang.txt)
            T.test_hip_fir_biquad_gain_lines_equal_oracle_loop({mode})
        T.test_hip_fused_chain_equals_separate_stages_and_oracle()
        T.test_mutation_reaches_hip_handle_through_the_message()
        T.test_hip_processor_error_surfaces_as_run_error()
    """, timeout_s=100, env={"PIPE_HOST_RESIDENT_SHARED": "1", "PIPE_HIP_STALL_DUMP_MS": "5000"})


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_the_host_loop_through_the_resident_path(mode):
    """run.go:198-224's loop with PIPE_HIP_PARAM_RESIDENT asked for by EVERY HIP stage (the harness's
    PIPE_HOST_RESIDENT is what hip.Stage.SetResident is in the Go shim): the first stage of the device that can take a
    queued launch back holds the doorbell, the others are answered PIPE_HIP_EBUSY / EINVAL and stay on the plain
    path -- the same known answers, the same bits as without it: short last buffers, restarts between runs, an
    in-band mutation, an error that ends the run.  In a child process with a limit of its own (async mode: a thread
    per component; this is the test that never returned in round 4)."""
    from tests._child import run_child
    run_child(f"""
        import tests.test_host_pipe as T
        T.test_hip_copy_in_the_loop_config1({mode})
        T.test_hip_fir_biquad_gain_lines_equal_oracle_loop({mode})
        T.test_hip_fused_chain_equals_separate_stages_and_oracle()
        T.test_mutation_reaches_hip_handle_through_the_message()
        T.test_hip_processor_error_surfaces_as_run_error()
    """, timeout_s=90, env={"PIPE_HOST_RESIDENT": "1"})


@pytest.mark.gpu
def test_many_lines_of_resident_stages_async_stress():
    """8 Lines x (FIR, gain), every stage asking for the doorbell, async mode (17+ threads), 2000 buffers a Line, 20
    times over: every run equal to the oracle's loop and done within 10 s.  Components that run concurrently must not
    hold each other up (merger.go:25-30, run.go:171-196, fitting.go:56-60): one stage holds the device's doorbell on a
    hardware queue of its own, nobody else parks anything."""
    from tests._child import run_child
    out = run_child("""
        import time
        import numpy as np
        from oracle import oracle as O
        from pipe_amd import host as H
        from pipe_amd import synth
        BUF, LINES, BUFFERS, REPS = 512, 8, 2000, 20
        taps = synth.fir_lowpass_taps(32)
        n = BUFFERS * BUF - 77   # (a short last buffer)
        mk = lambda i: dict(limit=n, channels=2, src_kind=H.SRC_SYNTH, seed=synth.line_seed(i), discard=False)
        olines = [O.Line(limit=n, channels=2, src_kind=O.SRC_SYNTH, seed=synth.line_seed(i), discard=False,
                         procs=[O.Proc(O.PROC_FIR, taps), O.Proc(O.PROC_GAIN, [0.5])]) for i in range(LINES)]
        oerr, ores = O.run_lines(BUF, olines)
        worst = 0.0
        for rep in range(REPS):
            hlines = [H.Line(procs=[H.Proc(H.PROC_HIP_FIR, taps), H.Proc(H.PROC_HIP_GAIN, [0.5])], **mk(i)) for i in range(LINES)]
            t0 = time.perf_counter()
            herr, hres = H.run(BUF, hlines, H.MODE_ASYNC)
            dt = time.perf_counter() - t0
            worst = max(worst, dt)
            assert not herr.failed, herr.message
            assert dt < 10.0, (rep, dt)
            for h, o in zip(hres, ores):
                assert (h.sink.messages, h.sink.samples) == (o.sink.messages, o.sink.samples)
                assert np.array_equal(h.values, o.values), rep
        print(f"worst run {worst:.2f} s", flush=True)
    """, timeout_s=280, env={"PIPE_HOST_RESIDENT": "1"})
    assert "worst run" in out


@pytest.mark.gpu
def test_the_synchronous_host_loop_with_every_stage_in_the_shared_doorbell_queue():
    """pipe.Run's synchronous executor (run.go:37-52, 112-132: all Lines in one thread, round-robin, a Line's stages in
    order) with EVERY HIP stage of 8 Lines in the device's shared doorbell queue (PIPE_HIP_PARAM_RESIDENT_SHARED; the
    harness's PIPE_HOST_RESIDENT_SHARED): the oracle's loop bit for bit, Lines that end at different times (EOF removal
    changes the call order: the queue follows), a short last buffer, a restart."""
    from tests._child import run_child
    out = run_child("""
        import time
        import numpy as np
        from oracle import oracle as O
        from pipe_amd import host as H
        from pipe_amd import synth
        BUF, LINES = 512, 8
        taps = synth.fir_lowpass_taps(32)
        lens = [600 * BUF - 77, 400 * BUF, 600 * BUF + 5, 100 * BUF, 601 * BUF, 7, 300 * BUF - 1, 600 * BUF]
        mk = lambda i: dict(limit=lens[i], channels=2, src_kind=H.SRC_SYNTH, seed=synth.line_seed(i), discard=False)
        olines = [O.Line(limit=lens[i], channels=2, src_kind=O.SRC_SYNTH, seed=synth.line_seed(i), discard=False,
                         procs=[O.Proc(O.PROC_FIR, taps), O.Proc(O.PROC_GAIN, [0.5])]) for i in range(LINES)]
        oerr, ores = O.run_lines(BUF, olines)
        for rep in range(3):
            hlines = [H.Line(procs=[H.Proc(H.PROC_HIP_FIR, taps), H.Proc(H.PROC_HIP_GAIN, [0.5])], **mk(i)) for i in range(LINES)]
            t0 = time.perf_counter()
            herr, hres = H.run(BUF, hlines, H.MODE_RUN)
            dt = time.perf_counter() - t0
            assert not herr.failed, herr.message
            for h, o in zip(hres, ores):
                assert (h.sink.messages, h.sink.samples) == (o.sink.messages, o.sink.samples)
                assert np.array_equal(h.values, o.values), rep
            calls = sum(r.sink.messages for r in hres) * 2
            print(f"run {rep}: {dt:.2f} s, {dt / calls * 1e6:.1f} us per stage call", flush=True)
    """, timeout_s=200, env={"PIPE_HOST_RESIDENT_SHARED": "1"})
    assert "us per stage call" in out


# ---------------------------------------------------------- stage-major (batched) Run
def test_run_batched_equals_run_without_groups():
    # no BatchGroup anywhere: the stage-major pass must give every Line what pipe.Run gives it
    lens = [7 * BUF + 77, BUF, 3, 4 * BUF + 1]
    mk = lambda n, i: dict(limit=n, channels=2, src_kind=H.SRC_SYNTH, seed=synth.line_seed(i), discard=False)
    a = [H.Line(procs=[H.Proc(H.PROC_MOCK)] * (1 + i % 2), **mk(n, i)) for i, n in enumerate(lens)]
    b = [H.Line(procs=[H.Proc(H.PROC_MOCK)] * (1 + i % 2), **mk(n, i)) for i, n in enumerate(lens)]
    e1, r1 = H.run(BUF, a, H.MODE_RUN)
    e2, r2 = H.run(BUF, b, H.MODE_RUN_BATCHED)
    assert not e1.failed and not e2.failed, e2.message
    for x, y in zip(r1, r2):
        assert np.array_equal(x.values, y.values)
        for cx, cy in ((x.source, y.source), (x.sink, y.sink), (x.procs[0], y.procs[0])):
            assert (cx.messages, cx.samples, cx.started, cx.flushed) == (cy.messages, cy.samples, cy.started, cy.flushed)


def test_run_batched_error_and_restart_rules():
    line = H.Line(limit=4 * BUF, channels=1, procs=[H.Proc(H.PROC_MOCK, err_on_call=True)])
    ok = H.Line(limit=4 * BUF, channels=1, procs=[H.Proc(H.PROC_MOCK)])
    err, res = H.run(BUF, [ok, line], H.MODE_RUN_BATCHED)
    assert err.failed and err.is_mock_error
    for r in res:  # the deferred flush reaches every started component (run.go:207-214)
        assert r.source.flushed and r.procs[0].flushed and r.sink.flushed
    err, res = H.run(BUF, [H.Line(limit=862 * BUF, channels=1, procs=[H.Proc(H.PROC_MOCK)])], H.MODE_RUN_BATCHED,
                     runs=2)
    assert not err.failed
    assert (res[0].sink.messages, res[0].sink.samples) == (2 * 862, 2 * 862 * BUF)


@pytest.mark.gpu
@pytest.mark.parametrize("runs", [1, 2])
def test_batched_chain_lines_equal_oracle_loop(runs):
    # five Lines of different lengths behind ONE device handle, one launch per pass: each Line
    # must still get exactly the oracle loop's float64 samples, including its short last buffer
    # and the Lines that end early (whose slots then idle)
    C_ = 2
    taps = synth.fir_lowpass_taps(256)
    sos = synth.biquad_rbj_lowpass()
    lens = [9 * BUF + 200, 9 * BUF + 200, 5 * BUF, 77, 6 * BUF + 511]
    hlines, olines = [], []
    for i, n in enumerate(lens):
        hlines.append(H.Line(limit=n, channels=C_, src_kind=H.SRC_SYNTH, seed=synth.line_seed(i), discard=False,
                             procs=[H.Proc(H.PROC_HIP_CHAIN, H.chain_params(taps, sos, 0.5)), H.Proc(H.PROC_MOCK)]))
        olines.append(O.Line(limit=n, channels=C_, src_kind=O.SRC_SYNTH, seed=synth.line_seed(i), discard=False,
                             procs=[O.Proc(O.PROC_FIR, taps), O.Proc(O.PROC_BIQUAD, sos), O.Proc(O.PROC_GAIN, [0.5]),
                                    O.Proc(O.PROC_COPY)]))
    herr, hres = H.run(BUF, hlines, H.MODE_RUN_BATCHED, runs=runs)
    oerr, ores = O.run_lines(BUF, olines)
    assert not herr.failed, herr.message
    for h, o, n in zip(hres, ores, lens):
        assert h.sink.samples == runs * o.sink.samples and h.sink.messages == runs * o.sink.messages
        # the sink keeps the LAST run's samples: after a restart they are the same samples again,
        # i.e. StartFunc zeroed the device state of every slot
        assert np.array_equal(h.values, o.values)


# ---------------------------------------------------------------- live edits of a running batched pipe
def test_live_add_line_and_insert_processor_follow_the_reference_rules():
    """Pipe.AddLine / Pipe.InsertProcessor on the stage-major executor: both are
    mutations the executor applies BETWEEN two passes (multiLineExecutor.addRoute run.go:134-145,
    startSyncProcessor run.go:147-169).  A Line that joins before pass k runs exactly as it would
    alone; an inserted Processor sees the buffers from that pass on, is started when it is inserted
    and flushed with the Line; nobody else notices."""
    mk = lambda n, i, **kw: H.Line(limit=n, channels=2, src_kind=H.SRC_SYNTH, seed=synth.line_seed(i), discard=False, **kw)
    lens = [7 * BUF + 77, 6 * BUF, 4 * BUF + 5]
    lines = [
        mk(lens[0], 0, procs=[H.Proc(H.PROC_MOCK), H.Proc(H.PROC_MOCK, insert_before_pass=2)]),
        mk(lens[1], 1, procs=[H.Proc(H.PROC_MOCK, insert_before_pass=3), H.Proc(H.PROC_MOCK)]),
        mk(lens[2], 2, procs=[H.Proc(H.PROC_MOCK)], join_before_pass=3),
    ]
    err, res = H.run(BUF, lines, H.MODE_RUN_BATCHED)
    assert not err.failed, err.message
    for i, n in enumerate(lens):
        alone_err, alone = H.run(BUF, [mk(n, i, procs=[H.Proc(H.PROC_MOCK)])], H.MODE_RUN)
        assert not alone_err.failed
        assert np.array_equal(res[i].values, alone[0].values)          # mock Processors are copies
        total = alone[0].sink.messages
        assert (res[i].sink.messages, res[i].sink.samples) == (total, n)
        assert res[i].source.started and res[i].source.flushed and res[i].sink.flushed
    total0, total1 = -(-lens[0] // BUF), -(-lens[1] // BUF)
    assert res[0].procs[0].messages == total0
    assert res[0].procs[1].messages == total0 - 2                       # inserted before pass 2
    assert res[1].procs[0].messages == total1 - 3 and res[1].procs[1].messages == total1
    for p in (res[0].procs[1], res[1].procs[0]):
        assert p.started and p.flushed
    assert res[2].procs[0].messages == -(-lens[2] // BUF)               # the late Line, whole


@pytest.mark.gpu
def test_live_add_line_joins_a_running_batch_handle_and_gain_is_inserted_mid_run():
    # Lines 0, 1 run a batched HIP chain (one device handle, slots 0..2); Line 2 joins before pass 4 and
    # takes slot 2: its state starts from silence (pipe_hip_start_lines), the others keep theirs.
    # A HIP gain is inserted behind Line 0's chain before pass 3: buffers 0..2 leave unscaled,
    # the rest scaled.  Everything bit for bit against the oracle loop.
    C_ = 2
    taps = synth.fir_lowpass_taps(64)
    sos = synth.biquad_rbj_lowpass()
    lens = [8 * BUF + 100, 9 * BUF, 5 * BUF + 17]
    chain = H.Proc(H.PROC_HIP_CHAIN, H.chain_params(taps, sos, 0.5))
    hlines = [
        H.Line(limit=lens[0], channels=C_, src_kind=H.SRC_SYNTH, seed=synth.line_seed(40), discard=False,
               procs=[chain, H.Proc(H.PROC_HIP_GAIN, [0.25], insert_before_pass=3)]),
        H.Line(limit=lens[1], channels=C_, src_kind=H.SRC_SYNTH, seed=synth.line_seed(41), discard=False, procs=[chain]),
        H.Line(limit=lens[2], channels=C_, src_kind=H.SRC_SYNTH, seed=synth.line_seed(42), discard=False, procs=[chain],
               join_before_pass=4),
    ]
    herr, hres = H.run(BUF, hlines, H.MODE_RUN_BATCHED)
    assert not herr.failed, herr.message
    for i, n in enumerate(lens):
        ol = O.Line(limit=n, channels=C_, src_kind=O.SRC_SYNTH, seed=synth.line_seed(40 + i), discard=False,
                    procs=[O.Proc(O.PROC_FIR, taps), O.Proc(O.PROC_BIQUAD, sos), O.Proc(O.PROC_GAIN, [0.5])])
        _, ores = O.run_lines(BUF, [ol])
        want = ores[0].values.copy()
        if i == 0:
            want[3 * BUF * C_:] *= 0.25     # exact: a power of two
        assert hres[i].sink.samples == n
        assert np.array_equal(hres[i].values, want), f"line {i}"
    assert hres[0].procs[1].messages == -(-lens[0] // BUF) - 3 and hres[0].procs[1].started and hres[0].procs[1].flushed


# ---------------------------------------------------------------- the pools recycle: no allocation in the steady state
@pytest.mark.parametrize("mode", [H.MODE_RUN, H.MODE_ASYNC, H.MODE_RUN_BATCHED])
def test_pool_allocators_create_no_buffers_in_the_steady_state(mode):
    """PoolAllocator (pipe.go:490-492; Source.execute pipe.go:394, Processor.execute :437 draw from it,
    `defer m.Signal.Free` :426,:458 hand the buffer back): the number of buffers a pipe ever creates is
    set by its topology -- a 100x longer stream creates exactly as many as a short one."""
    def created(buffers):
        mk = lambda i: H.Line(limit=buffers * BUF, channels=2, src_kind=H.SRC_SYNTH, seed=synth.line_seed(i),
                              procs=[H.Proc(H.PROC_MOCK), H.Proc(H.PROC_MOCK)])
        before = H.pool_buffers_created()
        err, res = H.run(BUF, [mk(0), mk(1)], mode)
        assert not err.failed and res[0].sink.messages == buffers
        return H.pool_buffers_created() - before
    short, long_ = created(8), created(800)
    if mode == H.MODE_ASYNC:
        # one thread per stage: a pool's buffers are the one being filled, the one waiting in the fitting
        # (capacity 1) and the one the next stage reads: at most 3 per pool, 3 pools per Line, 2 Lines --
        # set by the pipeline's depth, not by the stream's length
        assert short <= 18 and long_ <= 18
    else:
        assert long_ == short == 2 * 3   # one buffer per pool: the input is freed before the next pass

