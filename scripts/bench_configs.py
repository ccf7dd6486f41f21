#!/usr/bin/env python3
"""Side measurements for the BASELINE.json configs that are not the bench.py headline
(configs[0], [1] per-call form, [2], [3], [4]).  One JSON object per line on stdout.
Run on the GPU box: python scripts/bench_configs.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pipe_amd import host as H  # noqa: E402
from pipe_amd import processors as P  # noqa: E402
from pipe_amd import synth  # noqa: E402

torch.cuda.set_stream(torch.cuda.Stream())  # device-resident calls launch directly on this stream (processors._TorchOrder)


def emit(**kw):
    print(json.dumps(kw), flush=True)


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def config0():
    # mock.Source -> Processor(gain 1.0) -> mock.Sink, 1 Line, 512-frame buffers, 862 buffers (pipe_test.go:82-106)
    line = lambda kind: H.Line(limit=862 * 512, channels=2, value=1.0, discard=True, procs=[H.Proc(kind)])
    for name, kind in (("host mock.Processor (CPU, the reference's own component)", H.PROC_MOCK),
                       ("HIP copy through pipe_hip_process (f64 buffers)", H.PROC_HIP_COPY)):
        for mode, mname in ((H.MODE_RUN, "pipe.Run"), (H.MODE_ASYNC, "pipe.New+Start")):
            H.run(512, [line(kind)], mode)
            t0 = time.perf_counter()
            err, res = H.run(512, [line(kind)], mode)
            dt = time.perf_counter() - t0
            assert not err.failed and res[0].sink.messages == 862
            emit(config=0, what=name, mode=mname, buffers=862, seconds=round(dt, 4),
                 us_per_buffer=round(dt / 862 * 1e6, 2), msamples_per_s=round(862 * 512 * 2 / dt / 1e6, 2))


def config1_per_call():
    F, C, N = 4096, 2, 256
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    # through the compiled host mirror (no Python in the loop): mock.Source -> HIP FIR -> mock.Sink.
    # bind + handle creation are paid once per run, so time two stream lengths and difference them
    for mode, mname in ((H.MODE_RUN, "pipe.Run"), (H.MODE_ASYNC, "pipe.New+Start")):
        def run_n(n):
            line = H.Line(limit=n * F, channels=C, value=0.25, discard=True, procs=[H.Proc(H.PROC_HIP_FIR, taps)])
            t0 = time.perf_counter()
            err, res = H.run(F, [line], mode)
            dt = time.perf_counter() - t0
            assert not err.failed and res[0].sink.messages == n
            return dt
        run_n(50)
        t_small, t_big = run_n(500), run_n(4500)
        per = (t_big - t_small) / 4000
        emit(config=1, what="C++ host mirror: mock.Source -> HIP FIR-256 (f64 buffers, ProcessFunc per buffer) -> mock.Sink, 4096x2",
             mode=mname, us_per_buffer=round(per * 1e6, 2), setup_ms=round((t_small - 500 * per) * 1e3, 2),
             msamples_per_s=round(F * C / per / 1e6, 2), realtime_factor_48k=round(F / 48000 / per, 1))
    x = synth.samples(synth.line_seed(0), 0, F * C, np.float32).reshape(F, C)
    for dtype in (np.float32, np.float64):
        with P.Fir(taps, F, C, dtype=dtype) as p:
            p.start()
            xin = x.astype(dtype)
            dt = timed(lambda: p.process(xin), 300, 20)
            emit(config=1, what="ProcessFunc form: pipe_hip_process, one 4096x2 buffer, H2D+kernel+D2H, synchronous",
                 io=str(np.dtype(dtype)), us_per_buffer=round(dt * 1e6, 2),
                 msamples_per_s=round(F * C / dt / 1e6, 2), realtime_factor_48k=round(F / 48000 / dt, 1))
            # the asynchronous form: buffer k + 1 is submitted while buffer k is on the device
            # (two staging slots per handle), so staging / launch overlap kernels and transfers
            def sc():
                p.submit(xin)
                p.collect()
            dt2 = timed(sc, 300, 20)
            emit(config=1, what="submit + collect, one buffer in flight", io=str(np.dtype(dtype)),
                 us_per_buffer=round(dt2 * 1e6, 2))
            p.submit(xin)

            def pipelined():
                p.submit(xin)
                p.collect()
            dt3 = timed(pipelined, 300, 20)
            p.collect()
            emit(config=1, what="submit(k+1) before collect(k): two buffers in flight", io=str(np.dtype(dtype)),
                 us_per_buffer=round(dt3 * 1e6, 2), msamples_per_s=round(F * C / dt3 / 1e6, 2))


def batch(config, what, proc, d_in, d_out, frames, samples, reps=20):
    # device-resident launches are timed at steady state: the first launches after an idle
    # period run 10-25 % slow (clocks, TLBs), so warm up for ~50 ms worth of work first
    warm = timed(lambda: proc.process_batch(d_in, d_out, frames), 3, 1)
    for _ in range(int(min(400, max(5, 0.05 / max(warm, 1e-6))))):
        proc.process_batch(d_in, d_out, frames)
    reps = int(min(400, max(reps, 0.05 / max(warm, 1e-6))))
    proc.set_profiling(True)
    proc.kernel_time(reset=True)
    dt = timed(lambda: proc.process_batch(d_in, d_out, frames), reps, 3)
    ms, n = proc.kernel_time(reset=True)
    emit(config=config, what=what, kernel=proc.kernel_name(), ms_per_step=round(dt * 1e3, 4),
         kernel_ms=round(ms / max(n, 1), 4), msamples_per_s=round(samples / dt / 1e6, 1),
         algorithmic_gb_s=round(samples * 8 / dt / 1e9, 1))


def pcie_inclusive(config, what, proc, L_, F, C, dtype=np.float32, reps=30):
    # the ProcessFunc form with host buffers: memcpy to pinned -> H2D -> kernel(s) -> D2H -> memcpy out,
    # synchronous, one buffer_size buffer per Line per call, straight through the C ABI
    import ctypes as CT
    from pipe_amd import _lib as LIB
    x = np.full((L_, F, C), 0.25, dtype=dtype)
    y = np.empty_like(x)
    n = CT.c_int32()
    fn = LIB.lib().pipe_hip_process
    dt = timed(lambda: LIB.check(fn(proc._h, x.ctypes.data, F, y.ctypes.data, F, CT.byref(n)), "process"), reps, 3)
    emit(config=config, what=what, ms_per_step=round(dt * 1e3, 4), msamples_per_s=round(x.size / dt / 1e6, 1),
         host_gb_s_each_way=round(x.nbytes / dt / 1e9, 2))


def config2():
    F, C, N, L, K = 4096, 2, 256, 64, 64
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    n = L * K * F * C
    d_in = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty_like(d_in)
    with P.Fir(taps, F, C, dtype=np.float32, lines=L, max_batch=K) as p:
        p.start()
        batch(2, f"{L} Lines x {K} buffers x 4096x2 f32, FIR-256, one launch per step", p, d_in, d_out, K * F, n)
        p.set_exact(True)
        batch(2, "same, bit-exact direct form", p, d_in, d_out, K * F, n, reps=5)
        p.set_exact(False)
        pcie_inclusive(2, f"PCIe-inclusive: pipe_hip_process, {L} Lines x one 4096x2 f32 host buffer each per call", p, L, F, C)


def config3():
    F, C, N, L = 4096, 8, 256, 512
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    q = synth.biquad_rbj_lowpass()
    n = L * F * C
    d_in = torch.empty(n, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    d_out = torch.empty_like(d_in)
    kw = dict(dtype=np.float32, lines=L, max_batch=1)
    with P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(q, F, C, **kw), P.Gain(0.7071067811865476, F, C, **kw)]) as p:
        p.start()
        batch(3, f"{L} Lines x 8 ch x 4096 frames f32, FIR-256 + biquad + gain chain (one fused kernel)",
              p, d_in, d_out, F, n, reps=10)
        pcie_inclusive(3, f"PCIe-inclusive: pipe_hip_process, the same chain from host buffers ({L} Lines x 4096x8 f32)",
                       p, L, F, C, reps=8)
    for name, mk in (("FIR-256 alone", lambda: P.Fir(taps, F, C, **kw)), ("biquad alone", lambda: P.Biquad(q, F, C, **kw)),
                     ("gain alone", lambda: P.Gain(0.5, F, C, **kw))):
        with mk() as p:
            p.start()
            batch(3, name, p, d_in, d_out, F, n, reps=10)


def config4():
    F, C, T, up, down = 4096, 2, 24, 160, 147
    K = 1024
    proto = synth.resampler_proto(up, down, T)
    n_in = K * F
    d_in = torch.empty(n_in * C, dtype=torch.float32, device="cuda")
    P.synth_fill(d_in, synth.line_seed(0))
    cap = -(-n_in * up // down) + 1
    d_out = torch.empty(cap * C, dtype=torch.float32, device="cuda")
    with P.Resampler(proto, T, up, down, F, C, dtype=np.float32, max_batch=K) as p:
        p.start()
        outn = [0]
        def step():
            outn[0] = p.resample_batch(d_in, n_in, d_out, cap)
        dt = timed(step, 20, 3)
        emit(config=4, what=f"polyphase resampler 44.1->48 kHz, {K} buffers of 4096x2 f32 per launch",
             ms_per_step=round(dt * 1e3, 4), in_frames=n_in, out_frames=outn[0],
             msamples_out_per_s=round(outn[0] * C / dt / 1e6, 1),
             algorithmic_gb_s=round((n_in + outn[0]) * C * 4 / dt / 1e9, 1))
    a = torch.empty(n_in * C, dtype=torch.float32, device="cuda")
    P.synth_fill(a, synth.line_seed(1))
    with P.Mix(2, F, C, dtype=np.float32, max_batch=K) as m:
        m.start()
        out = torch.empty_like(a)
        dt = timed(lambda: m.mix_batch([d_in, a], out, n_in), 20, 3)
        emit(config=4, what="2-input mix (the build-defined 'fan-in'), same size", ms_per_step=round(dt * 1e3, 4),
             algorithmic_gb_s=round(3 * n_in * C * 4 / dt / 1e9, 1))
    with P.Gain(0.5, F, C, dtype=np.float32, max_batch=K) as g:
        g.start()
        out = torch.empty_like(a)
        # larger stream so the working set exceeds the 256 MiB Infinity Cache
        big = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
        bout = torch.empty_like(big)
        gg = P.Gain(0.5, F, C, dtype=np.float32, max_batch=16384)
        gg.start()
        dt = timed(lambda: gg.process_batch(big, bout, big.numel() // C), 20, 3)
        emit(config="copy/gain", what="gain kernel, 256 MiB in + 256 MiB out f32 (HBM-bound reference point)",
             ms_per_step=round(dt * 1e3, 4), algorithmic_gb_s=round(big.numel() * 8 / dt / 1e9, 1),
             hbm_frac=round(big.numel() * 8 / dt / 8e12, 4))
        gg.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["0", "1", "2", "3", "4"]
    for w in which:
        {"0": config0, "1": config1_per_call, "2": config2, "3": config3, "4": config4}[w]()
