"""The oracle's restatement of the reference loop against the reference's OWN
known answers.  Each test cites the reference test that
holds the expected value."""
import numpy as np
import pytest

from oracle import oracle as O

BUF = 512  # pipe_test.go:17
MOCK_ERR = 77


def mock_line(limit, channels=1, procs=1, discard=True, value=0.0, **kw):
    return O.Line(limit=limit, channels=channels, value=value, discard=discard,
                  procs=[O.Proc(O.PROC_COPY) for _ in range(procs)], **kw)


def assert_line(res, messages, samples):
    # assertLine(): source, processor and sink all see the same counts (pipe_test.go:655-663)
    for c in [res.source, *res.procs, res.sink]:
        assert (c.messages, c.samples) == (messages, samples)


@pytest.mark.parametrize("limit,messages", [(1040, 3), (1640, 4), (3048, 6), (4096, 8)])
def test_short_last_buffer_counts(limit, messages):
    # pipe_test.go:337,363,394,399,404 (TestLines, buf 512, C=1)
    err, res = O.run_lines(BUF, [mock_line(limit)])
    assert err.ok
    assert_line(res[0], messages, limit)
    assert res[0].source.flushed and res[0].procs[0].flushed and res[0].sink.flushed


def test_two_and_three_lines_round_robin():
    # "two lines ok" / "three lines ok"  pipe_test.go:356-436
    err, res = O.run_lines(BUF, [mock_line(1040), mock_line(1640)])
    assert err.ok
    assert_line(res[0], 3, 1040)
    assert_line(res[1], 4, 1640)
    err, res = O.run_lines(BUF, [mock_line(3048), mock_line(1640), mock_line(4096)])
    assert err.ok
    assert_line(res[0], 6, 3048)
    assert_line(res[1], 4, 1640)
    assert_line(res[2], 8, 4096)
    for r in res:
        assert r.source.flushed and r.procs[0].flushed and r.sink.flushed


def test_simple_pipe_862_buffers_two_channels():
    # TestSimplePipe pipe_test.go:82-106: 862 messages, 862*512 frames, C=2
    err, res = O.run_lines(BUF, [mock_line(862 * BUF, channels=2)])
    assert err.ok
    assert_line(res[0], 862, 862 * BUF)


def test_source_to_sink_without_processors():
    # line_test.go:11-19
    err, res = O.run_lines(BUF, [mock_line(862 * BUF, channels=2, procs=0)])
    assert err.ok
    assert (res[0].sink.messages, res[0].sink.samples) == (862, 862 * BUF)


@pytest.mark.parametrize("limit,value,calls", [(11, 1.0, 3), (2500, 2.0, 500)])
def test_mock_source_calls(limit, value, calls):
    # TestSource mock_test.go:69-92: buffer 5, C=2
    err, res = O.run_lines(5, [mock_line(limit, channels=2, procs=0, value=value, discard=False)])
    assert err.ok
    assert (res[0].source.messages, res[0].source.samples) == (calls, limit)
    # constant source: every scalar sample == Value (mock.go:100-102)
    assert res[0].values.size == limit * 2
    assert np.all(res[0].values == value)


@pytest.mark.parametrize("data", [[1, 1, 1, 1], [1, 1, 1, 1, 2, 2, 2, 2]])
def test_mock_processor_is_identity_and_sink_appends(data):
    # TestProcessor mock_test.go:133-146, TestSink mock_test.go:185-202 (Channels: 1)
    x = np.array(data, dtype=np.float64)
    line = O.Line(limit=x.size, channels=1, src_kind=O.SRC_ARRAY, data=x,
                  procs=[O.Proc(O.PROC_COPY)], discard=False)
    err, res = O.run_lines(x.size, [line])
    assert err.ok
    assert np.array_equal(res[0].values, x)


def test_constant_source_through_copy_reaches_sink():
    err, res = O.run_lines(BUF, [mock_line(1040, channels=2, value=3.25, discard=False)])
    assert err.ok
    assert res[0].values.size == 1040 * 2 and np.all(res[0].values == 3.25)


def test_restart_doubles_sink_counts():
    # TestReset pipe_test.go:108-131
    p = O.Pipe(BUF, [mock_line(862 * BUF, channels=2, procs=0)])
    assert p.run().ok
    r = p.results()[0]
    assert (r.source.messages, r.source.samples) == (862, 862 * BUF)
    p.reset_source(0)
    assert p.run().ok
    r = p.results()[0]
    assert (r.sink.messages, r.sink.samples) == (2 * 862, 2 * 862 * BUF)


def test_hook_order_single_line_processor_start_error():
    # pipe_test.go:310-329
    line = O.Line(limit=1040, channels=1, procs=[O.Proc(O.PROC_COPY, err_on_start=MOCK_ERR)],
                  discard=False)
    err, res = O.run_lines(BUF, [line])
    assert err.err_start == MOCK_ERR
    r = res[0]
    assert (r.source.started, r.procs[0].started, r.sink.started) == (True, True, False)
    assert (r.source.flushed, r.procs[0].flushed, r.sink.flushed) == (True, False, False)


@pytest.mark.parametrize("src_flush_err", [0, MOCK_ERR])
def test_hook_order_two_lines_processor_start_error(src_flush_err):
    # pipe_test.go:228-309
    l1 = O.Line(limit=1040, channels=1, procs=[O.Proc(O.PROC_COPY)], discard=False,
                src_err_on_flush=src_flush_err)
    l2 = O.Line(limit=1040, channels=1, procs=[O.Proc(O.PROC_COPY, err_on_start=MOCK_ERR)],
                discard=False)
    err, res = O.run_lines(BUF, [l1, l2])
    assert err.err_start == MOCK_ERR
    assert err.err_flush == src_flush_err
    a, b = res
    assert (a.source.started, a.procs[0].started, a.sink.started) == (True, True, True)
    assert (a.source.flushed, a.procs[0].flushed, a.sink.flushed) == (True, True, True)
    assert (b.source.started, b.procs[0].started, b.sink.started) == (True, True, False)
    assert (b.source.flushed, b.procs[0].flushed, b.sink.flushed) == (True, False, False)


def test_processor_error_propagates_and_everything_is_flushed():
    # "single processor error" pipe_test.go:437-457
    line = O.Line(limit=1040, channels=1, procs=[O.Proc(O.PROC_COPY, err_on_call=MOCK_ERR)])
    err, res = O.run_lines(BUF, [line])
    assert err.err_exec == MOCK_ERR
    r = res[0]
    assert r.source.flushed and r.procs[0].flushed and r.sink.flushed


def test_dsp_chain_through_the_loop_equals_array_api():
    # the loop only moves buffers: FIR->biquad->gain through 512-frame buffers with
    # a short last buffer must equal the same bodies applied to the whole stream
    rng = np.random.default_rng(7)
    C, frames = 2, 5 * BUF + 123
    x = rng.uniform(-1, 1, frames * C)
    taps = rng.uniform(-0.2, 0.2, 33)
    sos = np.array([[0.2, 0.3, 0.1, -0.5, 0.25]])
    line = O.Line(limit=frames, channels=C, src_kind=O.SRC_ARRAY, data=x, discard=False,
                  procs=[O.Proc(O.PROC_FIR, taps), O.Proc(O.PROC_BIQUAD, sos), O.Proc(O.PROC_GAIN, [0.5])])
    err, res = O.run_lines(BUF, [line])
    assert err.ok
    assert res[0].sink.messages == 6 and res[0].sink.samples == frames
    want = O.gain(O.Biquad(sos, C).process(O.Fir(taps, C).process(x)), 0.5)
    assert np.array_equal(res[0].values, want)
