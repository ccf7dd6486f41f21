#!/usr/bin/env python3
"""PCIe-inclusive rate of pipe_hip_process_lines_pinned at BASELINE configs[3]'s shape: 512 Lines x 4096 x 8 float32
(64 MiB each way per call), every Line's buffer its own pinned allocation (pipe_hip_host_alloc: the pool a Go host
builds on it, pipe.go:490-492) -- no host copy anywhere, the row kernels read and write the caller's buffers over
PCIe.  Against pipe_hip_process_lines (pageable rows, copied by the pool of copy threads)."""
import ctypes as CT
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pipe_amd import _lib as LIB  # noqa: E402
from pipe_amd import processors as P  # noqa: E402
from pipe_amd import synth  # noqa: E402

L_, F, C = int(os.environ.get("PROBE_LINES", "512")), 4096, 8
taps = synth.fir_lowpass_taps(256, f32_rounded=True)
q = synth.biquad_rbj_lowpass()
kw = dict(dtype=np.float32, lines=L_, max_batch=1)
lib = LIB.lib()
row = F * C * 4


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def pinned_rows(n, slab):
    """n row pointers: carved from ONE pinned block (a pool) or one allocation each."""
    if slab:
        p = CT.c_void_p()
        LIB.check(lib.pipe_hip_host_alloc(row * n, CT.byref(p)), "host_alloc")
        return [p.value + i * row for i in range(n)], [p.value]
    ptrs = []
    for _ in range(n):
        p = CT.c_void_p()
        LIB.check(lib.pipe_hip_host_alloc(row, CT.byref(p)), "host_alloc")
        ptrs.append(p.value)
    return ptrs, ptrs


rng = np.random.default_rng(1)
x = rng.uniform(-1, 1, (L_, F, C)).astype(np.float32)
for what, mk in (("gain", lambda: P.Gain(0.5, F, C, **kw)),
                 ("configs[3] chain FIR-256 -> biquad -> gain", lambda: P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(q, F, C, **kw),
                                                                                   P.Gain(0.7071067811865476, F, C, **kw)]))):
    frames = (CT.c_int32 * L_)(*([F] * L_))
    outf = (CT.c_int32 * L_)()
    y = np.empty_like(x)
    pin, pout = (CT.c_void_p * L_)(*[x[l].ctypes.data for l in range(L_)]), (CT.c_void_p * L_)(*[y[l].ctypes.data for l in range(L_)])
    res, allocs, same_all = {}, [], True
    for label, slab in (("pinned, one block carved into rows (DMA engines)", True), ("pinned, one allocation per row (row kernels)", False)):
        (ins_p, a1), (outs_p, a2) = pinned_rows(L_, slab), pinned_rows(L_, slab)
        allocs += a1 + a2
        for l in range(L_):
            CT.memmove(ins_p[l], x[l].ctypes.data, row)
        tin, tout = (CT.c_void_p * L_)(*ins_p), (CT.c_void_p * L_)(*outs_p)
        with mk() as p:
            p.start()
            res[label] = timed(lambda: LIB.check(lib.pipe_hip_process_lines_pinned(p._h, tin, frames, tout, outf), "lines_pinned"))
            got = np.stack([np.ctypeslib.as_array((CT.c_float * (F * C)).from_address(outs_p[l])).reshape(F, C).copy() for l in (0, L_ - 1)])
        if slab:
            got_slab = got
        else:
            same_all = bool(np.array_equal(got, got_slab))
    ins_p, outs_p = [], allocs
    with mk() as p:
        p.start()
        res["pageable"] = timed(lambda: LIB.check(lib.pipe_hip_process_lines(p._h, pin, frames, pout, outf), "lines"))
    # (same stream position for both handles' LAST call: compare it)
    same = bool(same_all and np.array_equal(got[0], y[0]) and np.array_equal(got[1], y[L_ - 1]))
    for k, dt in res.items():
        print(json.dumps({"what": what, "lines": L_, "path": k, "ms_per_call": round(dt * 1e3, 3),
                          "host_gb_s_each_way": round(L_ * row / dt / 1e9, 2), "same_result_as_other_path": same}), flush=True)
    for ptr in ins_p + outs_p:
        lib.pipe_hip_host_free(CT.c_void_p(ptr))
