"""ctypes binding of libpipe_hip.so (include/pipe_hip.h).

There is no CPU fallback: if the HIP library has not been built, or cannot be
loaded, importing a Processor fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PIPE_HIP_LIB: an alternative build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("PIPE_HIP_LIB") or os.path.join(_HERE, "lib", "libpipe_hip.so")

OK, EINVAL, ENODEV, EHIP, ENOMEM, ECAP, ESTATE, EBUSY = range(8)
F32, F64 = 0, 1
PARAM_GAIN, PARAM_TAPS, PARAM_COEFFS, PARAM_EXACT, PARAM_RESIDENT, PARAM_DEBUG, PARAM_RELAXED_F64, PARAM_RESIDENT_SHARED = 0, 1, 2, 3, 4, 5, 6, 7


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("buffer_size", C.c_int32), ("channels", C.c_int32),
                ("dtype", C.c_int32), ("lines", C.c_int32), ("max_batch", C.c_int32)]


class PipeHipError(RuntimeError):
    def __init__(self, status: int, what: str):
        self.status = status
        msg = lib().pipe_hip_strerror(status).decode()
        hip = lib().pipe_hip_last_hip_error() if status == EHIP else 0
        super().__init__(f"{what}: {msg} (status {status}" + (f", hipError {hip})" if hip else ")"))


HOST_LIB_PATH = os.path.join(_HERE, "lib", "libpipe_host.so")

_lib = None
_host_lib = None


def host_lib():
    """libpipe_host.so: the C++ mirror of the reference's host side (include/pipe_host.h), TEST
    HARNESS -- it drives libpipe_hip.so through the C ABI the way the Go pipe would."""
    global _host_lib
    if _host_lib is None:
        lib()  # the product library first: one HIP runtime, and libpipe_host.so links against it
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} is missing: build it with __graft_entry__.build()")
        _host_lib = C.CDLL(HOST_LIB_PATH, mode=C.RTLD_GLOBAL)
    return _host_lib



def _share_torch_hip_runtime():
    """One process, one HIP runtime.  PyTorch-ROCm ships its own libamdhip64.so; if libpipe_hip.so
    is loaded first it binds /opt/rocm's copy, torch later brings the other one, and the runtime
    that initialises second finds no device.  Loading torch's copy first (when torch is installed
    and not imported yet) makes both resolve to the same library, whichever import order the
    caller uses -- e.g. build() followed by smoke() in one interpreter."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass  # fall back to the system runtime; a GPU-less container lands here too


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). pipe_amd has no CPU fallback.")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    dp = C.POINTER(C.c_double)
    hp = C.POINTER(vp)
    cfgp = C.POINTER(Config)
    protos = {
        "pipe_hip_abi_version": (C.c_int, []),
        "pipe_hip_build_flags": (C.c_int, []),
        "pipe_hip_strerror": (C.c_char_p, [C.c_int]),
        "pipe_hip_last_hip_error": (C.c_int, []),
        "pipe_hip_device_count": (C.c_int, [C.POINTER(i32)]),
        "pipe_hip_gain_create": (C.c_int, [cfgp, dbl, hp]),
        "pipe_hip_fir_create": (C.c_int, [cfgp, dp, i32, hp]),
        "pipe_hip_biquad_create": (C.c_int, [cfgp, dp, i32, hp]),
        "pipe_hip_resampler_create": (C.c_int, [cfgp, dp, i32, i32, i32, hp]),
        "pipe_hip_mix_create": (C.c_int, [cfgp, i32, hp]),
        "pipe_hip_chain_create": (C.c_int, [hp, i32, hp]),
        "pipe_hip_output_properties": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "pipe_hip_start": (C.c_int, [vp]),
        "pipe_hip_flush": (C.c_int, [vp]),
        "pipe_hip_start_lines": (C.c_int, [vp, i32, i32]),
        "pipe_hip_destroy": (C.c_int, [vp]),
        "pipe_hip_process": (C.c_int, [vp, vp, i32, vp, i32, C.POINTER(i32)]),
        "pipe_hip_mix_process": (C.c_int, [vp, hp, i32, i32, vp]),
        "pipe_hip_process_lines": (C.c_int, [vp, hp, C.POINTER(i32), hp, C.POINTER(i32)]),
        "pipe_hip_process_lines_pinned": (C.c_int, [vp, hp, C.POINTER(i32), hp, C.POINTER(i32)]),
        "pipe_hip_submit": (C.c_int, [vp, vp, i32]),
        "pipe_hip_collect": (C.c_int, [vp, vp, i32, C.POINTER(i32)]),
        "pipe_hip_set_param": (C.c_int, [vp, i32, dp, i32]),
        "pipe_hip_chain_set_param": (C.c_int, [vp, i32, i32, dp, i32]),
        "pipe_hip_process_batch": (C.c_int, [vp, vp, vp, i64, vp]),
        "pipe_hip_resample_batch": (C.c_int, [vp, vp, i64, vp, i64, C.POINTER(i64), vp]),
        "pipe_hip_mix_batch": (C.c_int, [vp, hp, i32, vp, i64, vp]),
        "pipe_hip_set_profiling": (C.c_int, [vp, i32]),
        "pipe_hip_kernel_time": (C.c_int, [vp, C.POINTER(dbl), C.POINTER(i64), i32]),
        "pipe_hip_kernel_name": (C.c_char_p, [vp]),
        "pipe_hip_resident_info": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)]),
        "pipe_hip_host_alloc": (C.c_int, [i64, hp]),
        "pipe_hip_host_free": (C.c_int, [vp]),
        "pipe_hip_synth_fill": (C.c_int, [i32, vp, i32, C.c_uint64, i64, i64, vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(L, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status: int, what: str):
    if status != OK:
        raise PipeHipError(status, what)
