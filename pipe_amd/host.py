"""ctypes binding of the host-side mirror of pipe.Run / pipe.New+Start+Wait
(include/pipe_host.h, pipe_amd/csrc/host).  Test/bench harness only."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib

MAX_PROCS = 8
PROC_MOCK, PROC_HIP_COPY, PROC_HIP_GAIN, PROC_HIP_FIR, PROC_HIP_BIQUAD, PROC_HIP_CHAIN = range(6)
SRC_CONST, SRC_SYNTH, SRC_ARRAY = range(3)
MODE_RUN, MODE_ASYNC, MODE_RUN_BATCHED = 0, 1, 2


class _ProcDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("params", C.POINTER(C.c_double)), ("n_params", C.c_int32),
                ("err_on_call", C.c_int32), ("err_on_start", C.c_int32), ("err_on_flush", C.c_int32),
                ("err_on_make", C.c_int32), ("mutate_gain", C.c_int32), ("mutated_gain", C.c_double),
                ("insert_before_pass", C.c_int32)]


class _LineDesc(C.Structure):
    _fields_ = [("src_kind", C.c_int32), ("src_limit", C.c_int64), ("src_value", C.c_double),
                ("src_channels", C.c_int32), ("src_seed", C.c_uint64), ("src_data", C.POINTER(C.c_double)),
                ("src_err_on_call", C.c_int32), ("src_err_on_start", C.c_int32),
                ("src_err_on_flush", C.c_int32), ("src_err_on_make", C.c_int32),
                ("n_procs", C.c_int32), ("procs", _ProcDesc * MAX_PROCS),
                ("sink_discard", C.c_int32), ("sink_err_on_call", C.c_int32),
                ("sink_err_on_start", C.c_int32), ("sink_err_on_flush", C.c_int32),
                ("sink_err_on_make", C.c_int32), ("join_before_pass", C.c_int32)]


class _Counter(C.Structure):
    _fields_ = [("messages", C.c_int64), ("samples", C.c_int64), ("started", C.c_int32),
                ("flushed", C.c_int32)]


class _LineResult(C.Structure):
    _fields_ = [("source", _Counter), ("procs", _Counter * MAX_PROCS), ("sink", _Counter),
                ("sink_values", C.POINTER(C.c_double)), ("sink_values_len", C.c_int64)]


class _Error(C.Structure):
    _fields_ = [("failed", C.c_int32), ("is_mock_error", C.c_int32), ("is_bind_error", C.c_int32),
                ("message", C.c_char * 512)]


@dataclass
class Proc:
    kind: int = PROC_MOCK
    params: Optional[Sequence[float]] = None
    err_on_call: bool = False
    err_on_start: bool = False
    err_on_flush: bool = False
    err_on_make: bool = False
    mutate_gain: Optional[float] = None
    insert_before_pass: int = 0   # MODE_RUN_BATCHED: > 0 = inserted live before that pass (Pipe.InsertProcessor)


@dataclass
class Line:
    limit: int = 0
    channels: int = 1
    value: float = 0.0
    src_kind: int = SRC_CONST
    seed: int = 0
    data: Optional[np.ndarray] = None
    src_err_on_call: bool = False
    src_err_on_start: bool = False
    src_err_on_flush: bool = False
    src_err_on_make: bool = False
    procs: List[Proc] = field(default_factory=list)
    discard: bool = True
    sink_err_on_call: bool = False
    sink_err_on_start: bool = False
    sink_err_on_flush: bool = False
    sink_err_on_make: bool = False
    join_before_pass: int = 0     # MODE_RUN_BATCHED: > 0 = added to the running pipe before that pass (Pipe.AddLine)


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
    failed: bool
    is_mock_error: bool
    is_bind_error: bool
    message: str


def chain_params(taps, coeffs, gain) -> np.ndarray:
    t = np.asarray(taps, dtype=np.float64).ravel()
    c = np.asarray(coeffs, dtype=np.float64).reshape(-1, 5)
    return np.concatenate([[t.size], t, [c.shape[0]], c.ravel(), [gain]])


def pool_buffers_created() -> int:
    """Signal buffers created by every PoolAllocator of this process so far."""
    fn = _lib.host_lib().pipe_host_pool_buffers_created
    fn.restype = C.c_int64
    fn.argtypes = []
    return int(fn())


def _cnt(c) -> Counter:
    return Counter(int(c.messages), int(c.samples), bool(c.started), bool(c.flushed))


def run(buffer_size: int, lines: Sequence[Line], mode: int = MODE_RUN, runs: int = 1, device: int = 0):
    L = _lib.host_lib()
    fn = L.pipe_host_run
    fn.restype = C.c_int
    fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(_LineDesc), C.POINTER(_LineResult),
                   C.POINTER(_Error), C.c_int32, C.c_int32]
    L.pipe_host_free_values.argtypes = [C.POINTER(C.c_double)]
    keep = []
    descs = (_LineDesc * len(lines))()
    for d, l in zip(descs, lines):
        d.src_kind, d.src_limit, d.src_value = l.src_kind, l.limit, l.value
        d.src_channels, d.src_seed = l.channels, l.seed
        if l.data is not None:
            a = np.ascontiguousarray(l.data, dtype=np.float64).ravel()
            keep.append(a)
            d.src_data = a.ctypes.data_as(C.POINTER(C.c_double))
        d.src_err_on_call, d.src_err_on_start = int(l.src_err_on_call), int(l.src_err_on_start)
        d.src_err_on_flush, d.src_err_on_make = int(l.src_err_on_flush), int(l.src_err_on_make)
        d.n_procs = len(l.procs)
        for k, p in enumerate(l.procs):
            q = d.procs[k]
            q.kind = p.kind
            if p.params is not None:
                a = np.ascontiguousarray(p.params, dtype=np.float64).ravel()
                keep.append(a)
                q.params = a.ctypes.data_as(C.POINTER(C.c_double))
                q.n_params = a.size
            q.err_on_call, q.err_on_start = int(p.err_on_call), int(p.err_on_start)
            q.err_on_flush, q.err_on_make = int(p.err_on_flush), int(p.err_on_make)
            if p.mutate_gain is not None:
                q.mutate_gain, q.mutated_gain = 1, float(p.mutate_gain)
            q.insert_before_pass = int(p.insert_before_pass)
        d.sink_discard = int(l.discard)
        d.sink_err_on_call, d.sink_err_on_start = int(l.sink_err_on_call), int(l.sink_err_on_start)
        d.sink_err_on_flush, d.sink_err_on_make = int(l.sink_err_on_flush), int(l.sink_err_on_make)
        d.join_before_pass = int(l.join_before_pass)
    res = (_LineResult * len(lines))()
    err = _Error()
    rc = fn(mode, buffer_size, len(lines), descs, res, C.byref(err), runs, device)
    if rc != 0:
        raise ValueError("pipe_host_run: bad scenario description")
    out = []
    for r, l in zip(res, lines):
        vals = None
        if not l.discard:
            n = int(r.sink_values_len)
            vals = np.ctypeslib.as_array(r.sink_values, shape=(n,)).copy() if n else np.empty(0)
        if r.sink_values:
            L.pipe_host_free_values(r.sink_values)
        out.append(LineResult(_cnt(r.source), [_cnt(r.procs[k]) for k in range(len(l.procs))],
                              _cnt(r.sink), vals))
    e = RunError(bool(err.failed), bool(err.is_mock_error), bool(err.is_bind_error),
                 err.message.decode(errors="replace"))
    return e, out
