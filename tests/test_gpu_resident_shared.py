"""PIPE_HIP_PARAM_RESIDENT_SHARED: SEVERAL handles of one device queue their next buffer's work on the device's ONE
doorbell queue, in the order they are called (pipe.Run's synchronous executor calls all Lines of a context round-robin,
a Line's stages in order: run.go:37-52, 112-132).  Every guarantee of the plain per-buffer path holds through it -- bit
for bit the oracle -- in the predicted order, in any other order (the work ahead is rung and taken back), with short
buffers, mutations, restarts, other entries and handles that come and go.  Each test runs in a child process with a
limit of its own (tests/_child.py): a queue that is never rung must cost one named failure, not the suite."""
import pytest

from tests._child import run_child

pytestmark = pytest.mark.gpu

PRELUDE = """
import time
import numpy as np
from oracle import oracle as O
from pipe_amd import _lib as L
from pipe_amd import processors as P
from pipe_amd import synth
F, C = 4096, 2
taps = synth.fir_lowpass_taps(256)
q = synth.biquad_rbj_lowpass()
def stream(seed, buffers, frames=F, channels=C):
    return synth.samples(synth.line_seed(seed), 0, buffers * frames * channels).reshape(buffers, frames, channels)
def make(kind, dtype):
    if kind == "fir":
        return P.Fir(taps, F, C, dtype=dtype), (lambda r=O.Fir(taps, C): r)()
    if kind == "gain":
        return P.Gain(0.7071067811865476, F, C, dtype=dtype), None
    if kind == "biquad":
        return P.Biquad(q, F, C, dtype=dtype), O.Biquad(q, C)
    return P.Chain([P.Fir(taps, F, C, dtype=dtype), P.Gain(0.5, F, C, dtype=dtype)]), O.Fir(taps, C)
def want_of(kind, ref, x, dtype):
    x64 = x.astype(dtype).astype(np.float64)
    if kind == "gain":
        return O.gain(x64, 0.7071067811865476).reshape(x.shape).astype(dtype)
    y = ref.process(x64).reshape(x.shape)
    return (O.gain(y, 0.5).reshape(x.shape) if kind == "chain" else y).astype(dtype)
"""


def test_handles_called_round_robin_all_hold_the_doorbell_and_equal_the_oracle():
    """Twelve handles (FIR, gain, FIR -> gain chains; float32 and float64) called round-robin, 40 buffers each, a short
    buffer in the middle: every one holds the doorbell, every buffer is the oracle's bit for bit, and NOTHING is dropped
    in the steady state (the prediction of the call order is exact); a call costs no more than the plain path's."""
    out = run_child(PRELUDE + """
kinds = ["fir", "gain", "chain", "fir", "gain", "chain"] * 2
dts = [np.float32] * 6 + [np.float64] * 6
hs = [make(k, d) for k, d in zip(kinds, dts)]
xs = [stream(100 + i, 40) for i in range(len(hs))]
for (h, _) in hs:
    h.start()
    assert h.set_resident_shared(True)
assert all(h.resident_info()[0] for h, _ in hs)
t_calls = []
for k in range(40):
    frames = 1000 if k == 17 else F
    for i, ((h, ref), kind, dt) in enumerate(zip(hs, kinds, dts)):
        xin = xs[i][k, :frames].astype(dt)
        t0 = time.perf_counter()
        got = h.process(xin)
        t_calls.append(time.perf_counter() - t0)
        assert np.array_equal(got, want_of(kind, ref, xs[i][k, :frames], dt)), (k, i, kind)
dropped = sum(h.resident_info()[2] for h, _ in hs)
# the short buffer and the one after it take queued work back (two per handle at most); nothing else does
assert dropped <= 3 * len(hs), dropped
plain = []
with P.Fir(taps, F, C, dtype=np.float32) as pl:
    pl.start()
    for k in range(40):
        t0 = time.perf_counter(); pl.process(xs[0][k].astype(np.float32)); plain.append(time.perf_counter() - t0)
fir_calls = sorted(t_calls[i] for i in range(len(t_calls)) if kinds[i % len(hs)] == "fir" and dts[i % len(hs)] == np.float32)
print(f"shared fir call median {fir_calls[len(fir_calls) // 2] * 1e6:.1f} us, plain {sorted(plain)[20] * 1e6:.1f} us, dropped {dropped}", flush=True)
for h, _ in hs:
    h.flush(); h.close()
""", timeout_s=120)
    assert "shared fir call median" in out


def test_any_order_of_calls_is_still_the_oracles():
    """Calls in RANDOM order (what an asynchronous host would do): the work ahead of a call's own is rung, run on stale
    input and taken back by its owners -- slower, counted, and still bit for bit."""
    out = run_child(PRELUDE + """
rng = np.random.default_rng(5)
kinds = ["fir", "chain", "gain", "fir", "biquad"]
hs = [make(k, np.float32) for k in kinds]
xs = [stream(200 + i, 30).astype(np.float32) for i in range(len(hs))]
pos = [0] * len(hs)
for (h, _) in hs:
    h.start()
    assert h.set_resident_shared(True)
while min(pos) < 30:
    i = int(rng.integers(len(hs)))
    if pos[i] >= 30:
        continue
    h, ref = hs[i]
    got = h.process(xs[i][pos[i]])
    assert np.array_equal(got, want_of(kinds[i], ref, xs[i][pos[i]], np.float32)), (i, pos[i])
    pos[i] += 1
print("dropped", [h.resident_info()[2] for h, _ in hs], flush=True)
assert sum(h.resident_info()[2] for h, _ in hs) > 0
for h, _ in hs:
    h.close()
""", timeout_s=120)
    assert "dropped" in out


def test_mutations_restarts_batch_calls_and_handles_that_come_and_go():
    """Other entries on a sharing handle take back EVERYTHING queued on the device; a taps mutation reaches the next
    buffer and no other; StartFunc begins from silence; a device-resident batch call in between; a handle destroyed
    with work of several handles queued; a handle that joins late; the exclusive doorbell and the shared queue exclude
    each other; a float64 biquad (its ordered recurrence cannot be taken back) is refused."""
    out = run_child(PRELUDE + """
import torch
a, ra = make("fir", np.float32)
b, rb = make("chain", np.float32)
c, _ = make("gain", np.float32)
xs = [stream(300 + i, 24).astype(np.float32) for i in range(4)]
for h in (a, b, c):
    h.start()
    assert h.set_resident_shared(True)
with P.Fir(taps, F, C, dtype=np.float32) as excl:
    assert excl.set_resident(True) is False            # the device's doorbell is shared: no exclusive holder beside it
with P.Biquad(q, F, C, dtype=np.float64) as b64:
    st = L.lib().pipe_hip_set_param(b64._h, L.PARAM_RESIDENT_SHARED, P._dptr(np.array([1.0])), 1)
    assert st == L.EINVAL, st
taps2 = taps[::-1].copy() * 0.5
d = None
for k in range(24):
    if k == 5:
        a.set_taps(taps2); ra.set_taps(taps2)
    if k == 9:
        b.start(); rb.reset()
    if k == 12:     # a device-resident batch call on a sharing handle, between two per-buffer calls
        xin = torch.from_numpy(xs[2][k]).cuda(); y = torch.empty_like(xin)
        c.process_batch(xin, y, F); torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), want_of("gain", None, xs[2][k], np.float32))
    if k == 14:     # a handle joins late
        d, rd = make("fir", np.float32); d.start(); assert d.set_resident_shared(True)
    if k == 20:     # ... and one goes, with everybody's work queued
        c.close(); c = None
    for i, (h, kind, ref) in enumerate(((a, "fir", ra), (b, "chain", rb), (c, "gain", None), (d, "fir", rd if d else None))):
        if h is None:
            continue
        assert np.array_equal(h.process(xs[i][k]), want_of(kind, ref, xs[i][k], np.float32)), (k, i)
# everybody leaves: the exclusive doorbell is free again
for h in (a, b, d):
    h.flush(); h.close()
with P.Fir(taps, F, C, dtype=np.float32) as excl:
    excl.start()
    assert excl.set_resident(True)
    with P.Gain(0.5, F, C, dtype=np.float32) as g2:
        assert g2.set_resident_shared(True) is False   # ... and excludes the shared queue while it is held
print("ok", flush=True)
""", timeout_s=120)
    assert "ok" in out


def test_queued_work_of_many_handles_does_not_hold_a_device_wide_wait_for_ever():
    """hipDeviceSynchronize (and a hipFree of anybody) waits for every queue of the device: with several handles' work
    parked and no call coming, the watchdog rings all of it after the idle limit; the handles carry on afterwards."""
    out = run_child(PRELUDE + """
import torch
hs = [make(k, np.float32) for k in ("fir", "gain", "chain")]
xs = [stream(400 + i, 6).astype(np.float32) for i in range(3)]
for h, _ in hs:
    h.start()
    assert h.set_resident_shared(True, idle_ms=100)
for k in range(3):
    for i, (h, ref) in enumerate(hs):
        assert np.array_equal(h.process(xs[i][k]), want_of(("fir", "gain", "chain")[i], ref, xs[i][k], np.float32))
t0 = time.perf_counter()
torch.cuda.synchronize()          # three handles have work parked: the watchdog rings it
dt = time.perf_counter() - t0
assert dt < 2.0, dt
tmp = torch.empty(1 << 20, device="cuda"); del tmp; torch.cuda.empty_cache()
for k in range(3, 6):
    for i, (h, ref) in enumerate(hs):
        assert np.array_equal(h.process(xs[i][k]), want_of(("fir", "gain", "chain")[i], ref, xs[i][k], np.float32))
assert sum(h.resident_info()[1] for h, _ in hs) >= 1      # dropped by the watchdog
print(f"device-wide wait returned after {dt * 1e3:.0f} ms", flush=True)
for h, _ in hs:
    h.close()
""", timeout_s=120)
    assert "device-wide wait returned" in out
