"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/dsp_oracle.h and oracle/pipe_oracle.h for what is
restated from the reference and what is "parity unpinned").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_pipe.so")
_lib = None

MAX_PROCS = 8
EOF = -1
PROC_COPY, PROC_GAIN, PROC_FIR, PROC_BIQUAD = 0, 1, 2, 3
SRC_CONST, SRC_SYNTH, SRC_ARRAY = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = ["dsp_oracle.c", "pipe_oracle.c", "dsp_oracle.h", "pipe_oracle.h", "cpu_baseline.c"]
    newest = max(os.path.getmtime(os.path.join(_HERE, s)) for s in srcs)
    outs = [_LIB_PATH, os.path.join(_HERE, "cpu_baseline")]
    if force or not all(os.path.exists(o) and os.path.getmtime(o) >= newest for o in outs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class _ProcDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("params", C.POINTER(C.c_double)), ("n_params", C.c_int),
                ("err_on_call", C.c_int), ("err_on_start", C.c_int), ("err_on_flush", C.c_int)]


class _LineDesc(C.Structure):
    _fields_ = [("src_kind", C.c_int), ("src_limit", C.c_int64), ("src_value", C.c_double),
                ("src_channels", C.c_int), ("src_seed", C.c_uint64),
                ("src_data", C.POINTER(C.c_double)),
                ("src_err_on_call", C.c_int), ("src_err_on_start", C.c_int),
                ("src_err_on_flush", C.c_int),
                ("n_procs", C.c_int), ("procs", _ProcDesc * MAX_PROCS),
                ("sink_discard", C.c_int), ("sink_err_on_call", C.c_int),
                ("sink_err_on_start", C.c_int), ("sink_err_on_flush", C.c_int)]


class _Counter(C.Structure):
    _fields_ = [("messages", C.c_int64), ("samples", C.c_int64),
                ("started", C.c_int), ("flushed", C.c_int)]


class _LineResult(C.Structure):
    _fields_ = [("source", _Counter), ("procs", _Counter * MAX_PROCS), ("sink", _Counter),
                ("sink_values", C.POINTER(C.c_double)), ("sink_values_len", C.c_int64)]


class _RunError(C.Structure):
    _fields_ = [("err_start", C.c_int), ("err_exec", C.c_int), ("err_flush", C.c_int)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.odsp_gain.argtypes = [dp, dp, C.c_int64, C.c_double]
        L.odsp_fir_new.restype = C.c_void_p
        L.odsp_fir_new.argtypes = [dp, C.c_int, C.c_int]
        L.odsp_fir_reset.argtypes = [C.c_void_p]
        L.odsp_fir_set_taps.argtypes = [C.c_void_p, dp]
        L.odsp_fir_process.argtypes = [C.c_void_p, dp, dp, C.c_int64]
        L.odsp_fir_free.argtypes = [C.c_void_p]
        L.odsp_biquad_new.restype = C.c_void_p
        L.odsp_biquad_new.argtypes = [dp, C.c_int, C.c_int]
        L.odsp_biquad_reset.argtypes = [C.c_void_p]
        L.odsp_biquad_set_coeffs.argtypes = [C.c_void_p, dp]
        L.odsp_biquad_process.argtypes = [C.c_void_p, dp, dp, C.c_int64]
        L.odsp_biquad_free.argtypes = [C.c_void_p]
        L.odsp_resampler_new.restype = C.c_void_p
        L.odsp_resampler_new.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.odsp_resampler_reset.argtypes = [C.c_void_p]
        L.odsp_resampler_out_frames.restype = C.c_int64
        L.odsp_resampler_out_frames.argtypes = [C.c_void_p, C.c_int64]
        L.odsp_resampler_process.restype = C.c_int64
        L.odsp_resampler_process.argtypes = [C.c_void_p, dp, C.c_int64, dp, C.c_int64]
        L.odsp_resampler_free.argtypes = [C.c_void_p]
        L.odsp_mix.argtypes = [C.POINTER(dp), C.c_int, dp, C.c_int64]
        L.odsp_synth_fill.argtypes = [C.c_uint64, C.c_int64, dp, C.c_int64]
        L.opipe_bind.restype = C.c_void_p
        L.opipe_bind.argtypes = [C.c_int, C.c_int, C.POINTER(_LineDesc)]
        L.opipe_run.argtypes = [C.c_void_p, C.POINTER(_RunError)]
        L.opipe_reset_source.argtypes = [C.c_void_p, C.c_int]
        L.opipe_results.argtypes = [C.c_void_p, C.POINTER(_LineResult)]
        L.opipe_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


# --------------------------------------------------------------------------- DSP
def gain(x, g: float) -> np.ndarray:
    x = _f64(x)
    out = np.empty_like(x)
    lib().odsp_gain(_dptr(x), _dptr(out), x.size, float(g))
    return out


class Fir:
    """Direct-form FIR, float64, fma chain k = 0..N-1 (oracle/dsp_oracle.h)."""

    def __init__(self, taps, channels: int):
        self.taps = _f64(taps)
        self.channels = channels
        self._h = lib().odsp_fir_new(_dptr(self.taps), self.taps.size, channels)
        if not self._h:
            raise ValueError("bad FIR parameters")

    def reset(self):
        lib().odsp_fir_reset(self._h)

    def set_taps(self, taps):
        t = _f64(taps)
        assert t.size == self.taps.size
        self.taps = t
        lib().odsp_fir_set_taps(self._h, _dptr(t))

    def process(self, x) -> np.ndarray:
        """x: (frames, channels) or flat interleaved."""
        x = _f64(x)
        frames = x.size // self.channels
        out = np.empty_like(x)
        lib().odsp_fir_process(self._h, _dptr(x), _dptr(out), frames)
        return out

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.odsp_fir_free(self._h)
            self._h = None


class Biquad:
    """DF2T cascade; coeffs shape (nsections, 5) = b0 b1 b2 a1 a2."""

    def __init__(self, coeffs, channels: int):
        self.coeffs = _f64(coeffs).reshape(-1, 5)
        self.channels = channels
        self._h = lib().odsp_biquad_new(_dptr(self.coeffs), self.coeffs.shape[0], channels)
        if not self._h:
            raise ValueError("bad biquad parameters")

    def reset(self):
        lib().odsp_biquad_reset(self._h)

    def set_coeffs(self, coeffs):
        c = _f64(coeffs).reshape(-1, 5)
        assert c.shape == self.coeffs.shape
        self.coeffs = c
        lib().odsp_biquad_set_coeffs(self._h, _dptr(c))

    def process(self, x) -> np.ndarray:
        x = _f64(x)
        frames = x.size // self.channels
        out = np.empty_like(x)
        lib().odsp_biquad_process(self._h, _dptr(x), _dptr(out), frames)
        return out

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.odsp_biquad_free(self._h)
            self._h = None


class Resampler:
    """Rational polyphase resampler up/down; proto has up*taps_per_phase taps."""

    def __init__(self, proto, taps_per_phase: int, up: int, down: int, channels: int):
        self.proto = _f64(proto)
        assert self.proto.size == up * taps_per_phase
        self.channels = channels
        self._h = lib().odsp_resampler_new(_dptr(self.proto), taps_per_phase, up, down, channels)
        if not self._h:
            raise ValueError("bad resampler parameters")

    def reset(self):
        lib().odsp_resampler_reset(self._h)

    def out_frames(self, in_frames: int) -> int:
        return int(lib().odsp_resampler_out_frames(self._h, in_frames))

    def process(self, x, out_cap_frames: Optional[int] = None) -> np.ndarray:
        x = _f64(x)
        frames = x.size // self.channels
        n = self.out_frames(frames)
        cap = n if out_cap_frames is None else out_cap_frames
        out = np.empty(max(cap, 1) * self.channels, dtype=np.float64)
        w = lib().odsp_resampler_process(self._h, _dptr(x), frames, _dptr(out), cap)
        if w < 0:
            raise OverflowError("output capacity too small")
        return out[: w * self.channels].copy()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.odsp_resampler_free(self._h)
            self._h = None


def mix(inputs: Sequence[np.ndarray]) -> np.ndarray:
    arrs = [_f64(a) for a in inputs]
    n = arrs[0].size
    assert all(a.size == n for a in arrs)
    out = np.empty(n, dtype=np.float64)
    ptrs = (C.POINTER(C.c_double) * len(arrs))(*[_dptr(a) for a in arrs])
    lib().odsp_mix(ptrs, len(arrs), _dptr(out), n)
    return out.reshape(arrs[0].shape)


def synth(seed: int, first_index: int, samples: int) -> np.ndarray:
    out = np.empty(samples, dtype=np.float64)
    lib().odsp_synth_fill(C.c_uint64(seed), first_index, _dptr(out), samples)
    return out


# --------------------------------------------------------------------------- pipe loop
@dataclass
class Proc:
    kind: int = PROC_COPY
    params: Optional[Sequence[float]] = None
    err_on_call: int = 0
    err_on_start: int = 0
    err_on_flush: int = 0


@dataclass
class Line:
    """mock.Source -> processors -> mock.Sink (mock/mock.go)."""
    limit: int = 0
    channels: int = 1
    value: float = 0.0
    src_kind: int = SRC_CONST
    seed: int = 0
    data: Optional[np.ndarray] = None
    src_err_on_call: int = 0
    src_err_on_start: int = 0
    src_err_on_flush: int = 0
    procs: List[Proc] = field(default_factory=list)
    discard: bool = True
    sink_err_on_call: int = 0
    sink_err_on_start: int = 0
    sink_err_on_flush: int = 0


@dataclass
class Counter:
    messages: int
    samples: int
    started: bool
    flushed: bool


@dataclass
class LineResult:
    source: Counter
    procs: List[Counter]
    sink: Counter
    values: Optional[np.ndarray]


@dataclass
class RunError:
    err_start: int
    err_exec: int
    err_flush: int

    @property
    def ok(self) -> bool:
        return not (self.err_start or self.err_exec or self.err_flush)


def _cnt(c: _Counter) -> Counter:
    return Counter(int(c.messages), int(c.samples), bool(c.started), bool(c.flushed))


class Pipe:
    """Bound oracle pipe; run() may be repeated (TestReset, pipe_test.go:108-131)."""

    def __init__(self, buffer_size: int, lines: Sequence[Line]):
        self._keep = []
        self._lines = list(lines)
        descs = (_LineDesc * len(lines))()
        for d, l in zip(descs, lines):
            d.src_kind = l.src_kind
            d.src_limit = l.limit
            d.src_value = l.value
            d.src_channels = l.channels
            d.src_seed = l.seed
            if l.data is not None:
                a = _f64(l.data)
                self._keep.append(a)
                d.src_data = _dptr(a)
            d.src_err_on_call = l.src_err_on_call
            d.src_err_on_start = l.src_err_on_start
            d.src_err_on_flush = l.src_err_on_flush
            d.n_procs = len(l.procs)
            for k, p in enumerate(l.procs):
                d.procs[k].kind = p.kind
                if p.params is not None:
                    a = _f64(p.params).ravel()
                    self._keep.append(a)
                    d.procs[k].params = _dptr(a)
                    d.procs[k].n_params = a.size
                d.procs[k].err_on_call = p.err_on_call
                d.procs[k].err_on_start = p.err_on_start
                d.procs[k].err_on_flush = p.err_on_flush
            d.sink_discard = 1 if l.discard else 0
            d.sink_err_on_call = l.sink_err_on_call
            d.sink_err_on_start = l.sink_err_on_start
            d.sink_err_on_flush = l.sink_err_on_flush
        self._descs = descs
        self._p = lib().opipe_bind(buffer_size, len(lines), descs)

    def run(self) -> RunError:
        e = _RunError()
        lib().opipe_run(self._p, C.byref(e))
        return RunError(e.err_start, e.err_exec, e.err_flush)

    def reset_source(self, line: int):
        lib().opipe_reset_source(self._p, line)

    def results(self) -> List[LineResult]:
        res = (_LineResult * len(self._lines))()
        lib().opipe_results(self._p, res)
        out = []
        for r, l in zip(res, self._lines):
            vals = None
            if not l.discard:
                n = int(r.sink_values_len)
                vals = np.ctypeslib.as_array(r.sink_values, shape=(n,)).copy() if n else np.empty(0)
            out.append(LineResult(_cnt(r.source), [_cnt(r.procs[k]) for k in range(len(l.procs))],
                                  _cnt(r.sink), vals))
        return out

    def __del__(self):
        if getattr(self, "_p", None) and _lib is not None:
            _lib.opipe_free(self._p)
            self._p = None


def run_lines(buffer_size: int, lines: Sequence[Line]):
    """pipe.Run(ctx, bufferSize, lines...) on the oracle (pipe.go:90-103)."""
    p = Pipe(buffer_size, lines)
    err = p.run()
    return err, p.results()


def cpu_baseline(lines: int, channels: int, frames: int, buffers: int, ntaps: int,
                 threads: int = 1) -> dict:
    """Time the oracle's restatement of the reference loop (bench.py cpu_baseline)."""
    import json
    build()
    out = subprocess.check_output([os.path.join(_HERE, "cpu_baseline"), str(lines), str(channels),
                                   str(frames), str(buffers), str(ntaps), str(threads)])
    return json.loads(out.decode())


def cpu_optimized(lines: int, channels: int, frames: int, buffers: int, ntaps: int, threads: int = 1) -> dict:
    """Time the OPTIMISED float32 CPU FIR (oracle/cpu_fir_opt.c: not the reference's algorithm,
    not bit-compatible -- bench.py reports it separately from cpu_baseline).  Compiled with
    -march=native on the box that runs it (a binary built elsewhere may use instructions this
    host lacks), into a temporary file."""
    import json
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "cpu_fir_opt")
        subprocess.check_call(["gcc", "-O3", "-std=c11", "-march=native", "-ffast-math", "-o", exe,
                               os.path.join(_HERE, "cpu_fir_opt.c"), "-lm", "-lpthread"])
        out = subprocess.check_output([exe, str(lines), str(channels), str(frames), str(buffers), str(ntaps),
                                       str(threads)])
    return json.loads(out.decode())
