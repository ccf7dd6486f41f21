"""Python face of the C ABI, used by the tests and bench.py.

Each class is one Processor component as the reference defines it
(pipe.go:49-60): constructed by an allocator that receives bufferSize and the
input SignalProperties (line.go:26-30), with the three hooks

    start()   -> StartFunc    (pipe.go:82-83)
    process() -> ProcessFunc  (pipe.go:62-64)   host buffers, synchronous
    flush()   -> FlushFunc    (pipe.go:84-86)

plus the device-resident `process_batch` used for roofline runs.  Everything is
a thin call into libpipe_hip.so; nothing here computes.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib as L


def _np_dtype(dtype) -> np.dtype:
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise TypeError("dtype must be float32 or float64")
    return dt


def _code(dt: np.dtype) -> int:
    return L.F64 if dt == np.dtype(np.float64) else L.F32


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class _TorchOrder:
    """Runs a device-resident call in torch's stream order.  The tensors come from torch: their memory
    is handed out and taken back in the order of TORCH's current stream, so the kernels that touch
    them must sit in that order too -- on the handle's own (non-blocking) stream a kernel could still
    be reading an input that torch has already given to the next allocation (seen: the fill of a
    later output landing in an earlier call's input).  torch's default stream cannot be named through
    the C ABI (its handle is NULL = "the handle's own stream"), so there every Processor owns a side
    stream: it waits for torch's current stream, the call is launched on it, torch's current stream
    waits for it (two dependency hops, ~30 us per call: benchmarks run under torch.cuda.set_stream of
    a stream of their own, which is named directly).  No host synchronisation anywhere."""

    def __init__(self, proc, stream):
        self.proc, self.explicit, self.cur = proc, stream, None

    def __enter__(self) -> int:
        if self.explicit:
            return self.explicit
        import torch
        p = self.proc
        cur = torch.cuda.current_stream(p.device)
        if cur.cuda_stream:          # a stream with a name: launch right there
            return cur.cuda_stream
        if getattr(p, "_side", None) is None:
            p._side = torch.cuda.Stream(device=p.device)
        self.cur = cur
        p._side.wait_stream(cur)
        return p._side.cuda_stream

    def __exit__(self, *exc):
        if self.cur is not None:
            self.cur.wait_stream(self.proc._side)
        return False


def _devptr(t) -> int:
    """Device pointer of a torch tensor (torch is plumbing only)."""
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("need a contiguous device tensor")
    return t.data_ptr()


class Processor:
    """Base: owns a pipe_hip_processor handle."""

    def __init__(self, buffer_size: int, channels: int, dtype=np.float32, device: int = 0,
                 lines: int = 1, max_batch: int = 1):
        self.dtype = _np_dtype(dtype)
        self.buffer_size = int(buffer_size)
        self.channels = int(channels)
        self.lines = int(lines)
        self.max_batch = int(max_batch)
        self.device = int(device)
        self._cfg = L.Config(self.device, self.buffer_size, self.channels, _code(self.dtype),
                             self.lines, self.max_batch)
        self._h = C.c_void_p()
        self._owned = False

    # -- lifecycle ---------------------------------------------------------------
    def _created(self, status: int, what: str):
        L.check(status, what)
        return self

    def start(self):
        L.check(L.lib().pipe_hip_start(self._h), "start")

    def start_lines(self, first: int, count: int = 1):
        """StartFunc of some Lines of a batch handle only (a Line joining a running group)."""
        L.check(L.lib().pipe_hip_start_lines(self._h, int(first), int(count)), "start_lines")

    def flush(self):
        L.check(L.lib().pipe_hip_flush(self._h), "flush")

    def close(self):
        if self._h and not self._owned:
            L.lib().pipe_hip_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- properties ------------------------------------------------------------------
    def output_properties(self):
        ch, up, down = C.c_int32(), C.c_int32(), C.c_int32()
        L.check(L.lib().pipe_hip_output_properties(self._h, C.byref(ch), C.byref(up), C.byref(down)),
                "output_properties")
        return ch.value, up.value, down.value

    # -- ProcessFunc -----------------------------------------------------------------
    def _shape_in(self, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        per_line = x.size // self.lines
        if per_line * self.lines != x.size or per_line % self.channels:
            raise ValueError("input size does not match lines x frames x channels")
        return x, per_line // self.channels

    def process(self, x: np.ndarray, out_cap_frames: Optional[int] = None) -> np.ndarray:
        x, frames = self._shape_in(x)
        cap = self.buffer_size if out_cap_frames is None else int(out_cap_frames)
        out_ch, _, _ = self.output_properties()
        out = np.empty((self.lines, max(cap, 1), out_ch), dtype=self.dtype)
        n = C.c_int32()
        L.check(L.lib().pipe_hip_process(self._h, x.ctypes.data, frames, out.ctypes.data, cap,
                                         C.byref(n)), "process")
        return self._shape_out(out, cap, n.value)

    def _shape_out(self, out, cap, n):
        # fixed-rate stages pack Lines at n frames; rate changers at cap frames
        out_ch = out.shape[2]
        flat = out.reshape(-1)
        if self._fixed_rate() or self.lines == 1:
            res = flat[: self.lines * n * out_ch].reshape(self.lines, n, out_ch)
        else:
            res = out[:, :n, :]
        return res[0].copy() if self.lines == 1 else res.copy()

    def _fixed_rate(self) -> bool:
        return True

    def process_lines(self, xs: Sequence[Optional[np.ndarray]], pinned: bool = False):
        """One multiLineExecutor pass (run.go:112-132) through a handle with lines = len(xs):
        xs[l] is Line l's buffer (frames x channels, its own length) or None for a Line that
        has ended.  Returns the per-Line outputs (None where the Line has ended)."""
        if len(xs) != self.lines:
            raise ValueError("need one entry per Line")
        out_ch, _, _ = self.output_properties()
        keep, ins, outs, frames = [], [], [], []
        slabs = []
        if pinned == "slab":
            # every row carved from ONE pinned block per direction, back to back in Line order (a pool built on
            # pipe_hip_host_alloc): rows of equal length then move by the DMA engines, a chunk of Lines per copy
            rows = [None if x is None else np.ascontiguousarray(x, dtype=self.dtype).reshape(-1, self.channels) for x in xs]
            nin = sum(r.size for r in rows if r is not None)
            nout = sum(r.shape[0] * out_ch for r in rows if r is not None)
            bi = self._pinned_like(np.empty(max(nin, 1), dtype=self.dtype))
            bo = self._pinned_like(np.empty(max(nout, 1), dtype=self.dtype))
            slabs = [bi, bo]
            pi = po = 0
            for r in rows:
                if r is None:
                    ins.append(None); outs.append(None); frames.append(0); keep.append(None)
                    continue
                a = bi[pi:pi + r.size].reshape(r.shape)
                a[...] = r
                o = bo[po:po + r.shape[0] * out_ch].reshape(r.shape[0], out_ch)
                pi += r.size
                po += r.shape[0] * out_ch
                keep.append((a, o))
                ins.append(a.ctypes.data); outs.append(o.ctypes.data); frames.append(a.shape[0])
            xs = []
        for x in xs:
            if x is None:
                ins.append(None); outs.append(None); frames.append(0); keep.append(None)
                continue
            a = np.ascontiguousarray(x, dtype=self.dtype).reshape(-1, self.channels)
            o = np.empty((a.shape[0], out_ch), dtype=self.dtype)
            if pinned:
                a, o = self._pinned_copy(a), self._pinned_like(o)
            keep.append((a, o))
            ins.append(a.ctypes.data); outs.append(o.ctypes.data); frames.append(a.shape[0])
        n = self.lines
        pin = (C.c_void_p * n)(*ins)
        pout = (C.c_void_p * n)(*outs)
        fr = (C.c_int32 * n)(*frames)
        wr = (C.c_int32 * n)()
        fn = L.lib().pipe_hip_process_lines_pinned if pinned else L.lib().pipe_hip_process_lines
        L.check(fn(self._h, pin, fr, pout, wr), "process_lines")
        res = []
        for l, k in enumerate(keep):
            if k is None:
                res.append(None)
                continue
            assert wr[l] == frames[l]
            res.append(np.array(k[1][: wr[l]], copy=True))
        if slabs:
            for arr in slabs:
                L.lib().pipe_hip_host_free(C.c_void_p(arr.ctypes.data))
        elif pinned:
            for k in keep:
                if k is not None:
                    for arr in k:
                        L.lib().pipe_hip_host_free(C.c_void_p(arr.ctypes.data))
        return res

    def _pinned_like(self, a: np.ndarray) -> np.ndarray:
        ptr = C.c_void_p()
        L.check(L.lib().pipe_hip_host_alloc(max(a.nbytes, 16), C.byref(ptr)), "host_alloc")
        buf = (C.c_char * max(a.nbytes, 16)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)

    def _pinned_copy(self, a: np.ndarray) -> np.ndarray:
        b = self._pinned_like(a)
        b[...] = a
        return b

    def submit(self, x: np.ndarray):
        x, frames = self._shape_in(x)
        L.check(L.lib().pipe_hip_submit(self._h, x.ctypes.data, frames), "submit")

    def collect(self, out_cap_frames: Optional[int] = None) -> np.ndarray:
        cap = self.buffer_size if out_cap_frames is None else int(out_cap_frames)
        out_ch, _, _ = self.output_properties()
        out = np.empty((self.lines, max(cap, 1), out_ch), dtype=self.dtype)
        n = C.c_int32()
        L.check(L.lib().pipe_hip_collect(self._h, out.ctypes.data, cap, C.byref(n)), "collect")
        return self._shape_out(out, cap, n.value)

    # -- device-resident batch ------------------------------------------------------------
    def process_batch(self, d_in, d_out, frames_per_line: Optional[int] = None, stream: int = 0):
        """d_in/d_out: torch device tensors of lines*frames*channels elements."""
        if frames_per_line is None:
            frames_per_line = d_in.numel() // (self.lines * self.channels)
        need = self.lines * int(frames_per_line) * self.channels
        self._check_device_tensor(d_in, need)
        self._check_device_tensor(d_out, need)
        with _TorchOrder(self, stream) as st:
            L.check(L.lib().pipe_hip_process_batch(self._h, _devptr(d_in), _devptr(d_out),
                                                   int(frames_per_line), C.c_void_p(st)),
                    "process_batch")

    def _check_device_tensor(self, t, need: int):
        import torch
        want = torch.float64 if self.dtype == np.dtype(np.float64) else torch.float32
        if t.dtype != want:
            raise TypeError(f"tensor dtype {t.dtype} does not match the handle's {self.dtype}")
        if t.numel() < need:
            raise ValueError(f"tensor holds {t.numel()} elements, the call needs {need}")

    # -- measurement -------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        L.check(L.lib().pipe_hip_set_profiling(self._h, 1 if on else 0), "set_profiling")

    def kernel_time(self, reset: bool = True):
        ms, n = C.c_double(), C.c_int64()
        L.check(L.lib().pipe_hip_kernel_time(self._h, C.byref(ms), C.byref(n), 1 if reset else 0),
                "kernel_time")
        return ms.value, n.value

    def kernel_name(self) -> str:
        return L.lib().pipe_hip_kernel_name(self._h).decode()

    def _set_param(self, param: int, values):
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        L.check(L.lib().pipe_hip_set_param(self._h, param, _dptr(v), v.size), "set_param")

    def set_relaxed_f64(self, on: bool = True):
        """PIPE_HIP_PARAM_RELAXED_F64: float64 buffers may take the biquad's tile form (explicit opt-in; the result
        differs from the oracle's float64 by the recurrence's own rounding noise)."""
        self._set_param(L.PARAM_RELAXED_F64, [1.0 if on else 0.0])

    def set_resident(self, on: bool = True, idle_ms: int = 0) -> bool:
        """PIPE_HIP_PARAM_RESIDENT: keep the next buffer's work queued on the device behind a doorbell, so that
        process() costs no kernel launch and no completion event (stages that can take a queued launch back:
        gain, FIR, the tile biquad, chains of those; one buffer of at most 1 MiB).  ONE handle per device can hold
        the doorbell: False when another handle holds it (PIPE_HIP_EBUSY) -- this handle stays on the plain path."""
        v = np.ascontiguousarray([float(idle_ms) if (on and idle_ms > 1) else (1.0 if on else 0.0)], dtype=np.float64)
        st = L.lib().pipe_hip_set_param(self._h, L.PARAM_RESIDENT, _dptr(v), 1)
        if st == L.EBUSY:
            return False
        L.check(st, "set_param(RESIDENT)")
        return True

    def set_resident_shared(self, on: bool = True, idle_ms: int = 0) -> bool:
        """PIPE_HIP_PARAM_RESIDENT_SHARED: as set_resident, for SEVERAL handles of one device that are called in a fixed
        rotation and never concurrently (pipe.Run's synchronous executor): they share the device's one doorbell queue.
        False when the device's doorbell is held exclusively, or sixteen handles share it already (PIPE_HIP_EBUSY)."""
        v = np.ascontiguousarray([float(idle_ms) if (on and idle_ms > 1) else (1.0 if on else 0.0)], dtype=np.float64)
        st = L.lib().pipe_hip_set_param(self._h, L.PARAM_RESIDENT_SHARED, _dptr(v), 1)
        if st == L.EBUSY:
            return False
        L.check(st, "set_param(RESIDENT_SHARED)")
        return True

    def resident_info(self):
        """(holds the doorbell, queued launches dropped by the watchdog, ... dropped by another entry)"""
        held, wd, en = C.c_int32(), C.c_int64(), C.c_int64()
        L.check(L.lib().pipe_hip_resident_info(self._h, C.byref(held), C.byref(wd), C.byref(en)), "resident_info")
        return bool(held.value), wd.value, en.value


class Gain(Processor):
    def __init__(self, gain: float, buffer_size: int, channels: int, **kw):
        super().__init__(buffer_size, channels, **kw)
        self._created(L.lib().pipe_hip_gain_create(C.byref(self._cfg), float(gain), C.byref(self._h)),
                      "gain_create")

    def set_gain(self, gain: float):
        self._set_param(L.PARAM_GAIN, [gain])


class Copy(Gain):
    """mock.Processor (mock/mock.go:130-157): the pass-through copy."""

    def __init__(self, buffer_size: int, channels: int, **kw):
        super().__init__(1.0, buffer_size, channels, **kw)


class Fir(Processor):
    def __init__(self, taps: Sequence[float], buffer_size: int, channels: int, **kw):
        super().__init__(buffer_size, channels, **kw)
        t = np.ascontiguousarray(taps, dtype=np.float64).ravel()
        self.ntaps = t.size
        self._created(L.lib().pipe_hip_fir_create(C.byref(self._cfg), _dptr(t), t.size,
                                                  C.byref(self._h)), "fir_create")

    def set_taps(self, taps):
        self._set_param(L.PARAM_TAPS, taps)

    def set_exact(self, exact: bool):
        """True pins the ordered-fma direct form (bit-exact); False (default) lets large
        float32 batches use the overlap-save FFT form (<= 1 ulp float32)."""
        self._set_param(L.PARAM_EXACT, [1.0 if exact else 0.0])


class Biquad(Processor):
    def __init__(self, coeffs, buffer_size: int, channels: int, **kw):
        super().__init__(buffer_size, channels, **kw)
        c = np.ascontiguousarray(coeffs, dtype=np.float64).reshape(-1, 5)
        self._created(L.lib().pipe_hip_biquad_create(C.byref(self._cfg), _dptr(c), c.shape[0],
                                                     C.byref(self._h)), "biquad_create")

    def set_coeffs(self, coeffs):
        self._set_param(L.PARAM_COEFFS, coeffs)

    def set_exact(self, exact: bool):
        """True pins the one-lane-per-series ordered recurrence (bit-exact); False (default) lets
        float32 results use the time-segmented form (<= 1 ulp float32)."""
        self._set_param(L.PARAM_EXACT, [1.0 if exact else 0.0])


class Resampler(Processor):
    def __init__(self, proto, taps_per_phase: int, up: int, down: int, buffer_size: int,
                 channels: int, **kw):
        super().__init__(buffer_size, channels, **kw)
        p = np.ascontiguousarray(proto, dtype=np.float64).ravel()
        if p.size != up * taps_per_phase:
            raise ValueError("proto must have up*taps_per_phase taps")
        self.up, self.down = up, down
        self._created(L.lib().pipe_hip_resampler_create(C.byref(self._cfg), _dptr(p), taps_per_phase,
                                                        up, down, C.byref(self._h)),
                      "resampler_create")

    def _fixed_rate(self) -> bool:
        return False

    def resample_batch(self, d_in, in_frames: int, d_out, out_cap_frames: int, stream: int = 0) -> int:
        n = C.c_int64()
        with _TorchOrder(self, stream) as st:
            L.check(L.lib().pipe_hip_resample_batch(self._h, _devptr(d_in), int(in_frames), _devptr(d_out),
                                                    int(out_cap_frames), C.byref(n), C.c_void_p(st)),
                    "resample_batch")
        return n.value


class Mix(Processor):
    def __init__(self, inputs: int, buffer_size: int, channels: int, **kw):
        super().__init__(buffer_size, channels, **kw)
        self.inputs = inputs
        self._created(L.lib().pipe_hip_mix_create(C.byref(self._cfg), inputs, C.byref(self._h)),
                      "mix_create")

    def process(self, xs: Sequence[np.ndarray], out_cap_frames=None) -> np.ndarray:  # type: ignore[override]
        arrs = [np.ascontiguousarray(x, dtype=self.dtype) for x in xs]
        if len(arrs) != self.inputs or any(a.size != arrs[0].size for a in arrs):
            raise ValueError("mix: need `inputs` arrays of equal size")
        if arrs[0].size % (self.lines * self.channels):
            raise ValueError("input size does not match lines x frames x channels")
        frames = arrs[0].size // (self.lines * self.channels)
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        out = np.empty_like(arrs[0])
        L.check(L.lib().pipe_hip_mix_process(self._h, ptrs, len(arrs), frames, out.ctypes.data),
                "mix_process")
        return out

    def mix_batch(self, d_ins, d_out, frames_per_line: int, stream: int = 0):
        need = self.lines * int(frames_per_line) * self.channels
        for t in list(d_ins) + [d_out]:
            self._check_device_tensor(t, need)
        ptrs = (C.c_void_p * len(d_ins))(*[_devptr(t) for t in d_ins])
        with _TorchOrder(self, stream) as st:
            L.check(L.lib().pipe_hip_mix_batch(self._h, ptrs, len(d_ins), _devptr(d_out),
                                               int(frames_per_line), C.c_void_p(st)),
                    "mix_batch")


class Chain(Processor):
    """A Line's Processors slice (line.go:17) fused on the device; takes ownership."""

    def __init__(self, stages: Sequence[Processor]):
        s0 = stages[0]
        super().__init__(s0.buffer_size, s0.channels, dtype=s0.dtype, device=s0.device,
                         lines=s0.lines, max_batch=s0.max_batch)
        arr = (C.c_void_p * len(stages))(*[s._h.value for s in stages])
        self._created(L.lib().pipe_hip_chain_create(arr, len(stages), C.byref(self._h)),
                      "chain_create")
        for s in stages:
            s._owned = True
        self.stages = list(stages)

    def set_exact(self, exact: bool):
        """PIPE_HIP_PARAM_EXACT on every stage that has a relaxed form: the staged chain of ordered forms."""
        self._set_param(L.PARAM_EXACT, [1.0 if exact else 0.0])

    def set_stage_param(self, stage: int, param: int, values):
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        L.check(L.lib().pipe_hip_chain_set_param(self._h, int(stage), param, _dptr(v), v.size),
                "chain_set_param")


def device_count() -> int:
    n = C.c_int32()
    L.lib().pipe_hip_device_count(C.byref(n))
    return n.value


def synth_fill(d_out, seed: int, first_index: int = 0, device: Optional[int] = None, stream: int = 0):
    """Fill a torch device tensor with the SplitMix64 synthetic stream."""
    import torch
    dt = L.F64 if d_out.dtype == torch.float64 else L.F32
    dev = d_out.device.index if device is None else device
    L.check(L.lib().pipe_hip_synth_fill(dev or 0, _devptr(d_out), dt, C.c_uint64(seed),
                                        int(first_index), d_out.numel(), C.c_void_p(stream or None)),
            "synth_fill")
