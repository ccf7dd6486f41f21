"""What the binding costs per ProcessFunc call: one 4096 x 2 float64 pipe buffer moved into the
pinned staging slice and the result back -- one interface call per sample (what integration/go/hip/hip.go did until
round 4: Sample(i) / SetSample(i, v), mock.go:100-102) against bulk copies (signal.ReadFloat64 / signal.WriteFloat64,
mock/mock_test.go:120,128) -- next to what the device round trip it wraps costs (pipe_hip_process of a 256-tap FIR
from pinned staging).  The C++ stand-in for the Go side: pipe_amd/csrc/host/binding_cost.cpp (no Go toolchain here).
    PYTHONPATH=. python scripts/binding_cost.py > profiles/r05_binding_cost.jsonl"""
import ctypes as C
import json
import os
import time

import numpy as np

from pipe_amd import _lib, synth

F, CH, REPS = 4096, 2, 3000
H = _lib.host_lib()
H.pipe_host_binding_cost.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
L = _lib.lib()
n = C.c_int32()
have_gpu = L.pipe_hip_device_count(C.byref(n)) == 0 and n.value > 0
pin_in, pin_out = C.c_void_p(), C.c_void_p()
if have_gpu:
    _lib.check(L.pipe_hip_host_alloc(F * CH * 8, C.byref(pin_in)), "host_alloc")
    _lib.check(L.pipe_hip_host_alloc(F * CH * 8, C.byref(pin_out)), "host_alloc")
out = (C.c_double * 4)()
assert H.pipe_host_binding_cost(F, CH, REPS, pin_in, pin_out, out) == 0
row = {"buffer": f"{F} x {CH} float64", "reps": REPS, "staging": "pinned (pipe_hip_host_alloc)" if have_gpu else "heap (no GPU here)",
       "host": f"{os.cpu_count()} cores", "unit": "us per buffer, in + out, median",
       "per_sample_interface_calls_f64_staging": round(out[0], 2), "bulk_copies_f64_staging": round(out[1], 2),
       "per_sample_interface_calls_f32_staging": round(out[2], 2), "bulk_copy_plus_slice_loops_f32_staging": round(out[3], 2),
       "stand_in": "C++ virtual calls the optimiser cannot see through (binding_cost.cpp); a Go interface call is no cheaper"}
print(json.dumps(row), flush=True)
if have_gpu:
    from pipe_amd import processors as P
    taps = synth.fir_lowpass_taps(256)
    for dt, name in ((np.float64, "f64"), (np.float32, "f32")):
        with P.Fir(taps, F, CH, dtype=dt) as fir:
            fir.start()
            x = np.zeros((F, CH), dt)
            y = np.empty_like(x)
            got = C.c_int32()
            call = lambda: L.pipe_hip_process(fir._h, x.ctypes.data, F, y.ctypes.data, F, C.byref(got))
            for _ in range(200):
                call()
            lat = []
            for _ in range(REPS):
                t0 = time.perf_counter()
                call()
                lat.append((time.perf_counter() - t0) * 1e6)
            print(json.dumps({"pipe_hip_process_fir256": name, "median_us": round(float(np.median(lat)), 2),
                              "note": "through ctypes (+3-4 us over C); the device round trip the binding's copies wrap"}), flush=True)
