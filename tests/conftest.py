import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# Environment switches that exist only in the A/B build of the library (`make AB=1`, pipe_amd/lib/libpipe_hip_ab.so,
# selected with PIPE_HIP_LIB): a test that forces a kernel variant through one of them is skipped against the
# default build, which does not read them (pipe_amd/csrc/common.hpp PH_ENV_AB; DESIGN.md lists what ships).
AB_ONLY = (
    "PIPE_HIP_BIQUAD_LDS", "PIPE_HIP_BIQUAD_NO_SP", "PIPE_HIP_BIQUAD_NO_SPLIT", "PIPE_HIP_BIQUAD_NO_TILE",
    "PIPE_HIP_BIQUAD_NO_WAVE_SCAN", "PIPE_HIP_BIQUAD_SEG_ONE_PER_BLOCK", "PIPE_HIP_BIQUAD_SPLIT_COPIES",
    "PIPE_HIP_BIQUAD_SPLIT_MAX_SERIES", "PIPE_HIP_BIQUAD_TILE_SEG", "PIPE_HIP_BIQUAD_TILE_WALK_LINES",
    "PIPE_HIP_BIQUAD_TWO_PASS", "PIPE_HIP_CHAIN_GENERAL", "PIPE_HIP_CHAIN_LOCAL", "PIPE_HIP_CHAIN_NO_TAIL",
    "PIPE_HIP_CHAIN_ONE_SECTION", "PIPE_HIP_CHAIN_STAGGER", "PIPE_HIP_FIR_MFMA_TF", "PIPE_HIP_FIR_NO_LT",
    "PIPE_HIP_FIR_NO_MFMA", "PIPE_HIP_FIR_NO_PARTITION", "PIPE_HIP_FIR_PARTITION_SUM", "PIPE_HIP_FIR_R\"",
    "PIPE_HIP_FIR_RUN_FLOOR", "PIPE_HIP_FIR_WGS_PER_CU", "PIPE_HIP_OLS_MONO_ALONE", "PIPE_HIP_OLS_VARIANT",
    "PIPE_HIP_OVERLAP_TRACE", "PIPE_HIP_RESAMPLE_F64_PLANES", "PIPE_HIP_RESAMPLE_GATHER", "PIPE_HIP_RESAMPLE_LDS_TAPS",
    "PIPE_HIP_RESAMPLE_NO_PAIR", "PIPE_HIP_RESAMPLE_PLANES",
)


def _skip_ab_only_tests(items):
    import inspect
    try:
        from pipe_amd import _lib
        if _lib.lib().pipe_hip_build_flags() & 1:
            return
    except Exception:  # noqa: BLE001 -- no library here: the gpu tests are skipped anyway
        return
    skip = pytest.mark.skip(reason="forces a kernel variant through an A/B-only switch: run against `make AB=1` "
                                   "(PIPE_HIP_LIB=pipe_amd/lib/libpipe_hip_ab.so)")
    for item in items:
        if "gpu" not in item.keywords:
            continue
        try:
            src = inspect.getsource(item.function)
        except (OSError, TypeError, AttributeError):
            continue
        params = " ".join(repr(v) for v in getattr(getattr(item, "callspec", None), "params", {}).values())
        if any(name in src or name in params for name in AB_ONLY):
            item.add_marker(skip)


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip silently; plain
    # runs on a CPU box skip the gpu tests.
    if _have_gpu():
        _skip_ab_only_tests(items)
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
