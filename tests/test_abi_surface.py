"""CPU-side checks of the drop-in boundary: libpipe_hip.so loads without a GPU and
exports every symbol include/pipe_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pipe_hip.h")


def declared_functions(path):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pipe_hip_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from pipe_amd import _lib
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_declares_the_expected_surface():
    names = declared_functions(HEADER)
    for must in ["pipe_hip_fir_create", "pipe_hip_gain_create", "pipe_hip_biquad_create",
                 "pipe_hip_resampler_create", "pipe_hip_mix_create", "pipe_hip_chain_create",
                 "pipe_hip_start", "pipe_hip_process", "pipe_hip_flush", "pipe_hip_destroy",
                 "pipe_hip_set_param", "pipe_hip_process_batch", "pipe_hip_submit", "pipe_hip_collect"]:
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_functions(HEADER) if not hasattr(lib, n)]
    assert not missing, f"declared in pipe_hip.h but not exported: {missing}"


def test_python_binding_prototypes_cover_the_header(lib):
    from pipe_amd import _lib
    L = _lib.lib()
    for n in declared_functions(HEADER):
        assert getattr(L, n).argtypes is not None, n


def test_abi_version_and_strerror(lib):
    assert lib.pipe_hip_abi_version() == 1
    lib.pipe_hip_strerror.restype = ctypes.c_char_p
    assert lib.pipe_hip_strerror(0) == b"ok"
    assert b"capacity" in lib.pipe_hip_strerror(5)


def test_create_without_gpu_fails_loudly_not_silently(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pipe_amd import processors as P
    from pipe_amd._lib import ENODEV, PipeHipError
    with pytest.raises(PipeHipError) as e:
        P.Gain(0.5, 512, 2)
    assert e.value.status == ENODEV  # no CPU fallback exists


def test_library_has_no_vgpr_spills():
    """hipcc (ROCm 7.2) may place a spill store inside an exec-masked region of a kernel; the lanes
    masked off at the spill then reload garbage (seen on a two-section form of the fused chain
    kernel: wrong channel offsets at the store).  So no kernel of the library may spill VGPRs:
    scripts/check_spills.sh recompiles every .hip with -Rpass-analysis=kernel-resource-usage."""
    import shutil
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc") and not shutil.which("hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "check_spills.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_kernel_family_the_sources_can_report_is_asserted_by_a_test_that_never_skips():
    """75 of the GPU tests force a variant through an A/B-only switch and skip against the library
    that ships.  Mechanically: every kernel name the sources can hand to pipe_hip_kernel_name belongs to a family that
    tests/test_gpu_kernel_families.py launches ON THE DEFAULT BUILD and asserts, and nothing in that file names an
    A/B-only switch (conftest.AB_ONLY_RE: the rule by which tests are skipped)."""
    import glob
    import re
    from tests import conftest
    from tests import test_gpu_kernel_families as T
    names = set()
    for f in glob.glob(os.path.join(ROOT, "pipe_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "pipe_amd", "csrc", "*.hpp")):
        names.update(re.findall(r'"([a-z0-9_]+_kernel<[^"]*)"', open(f).read()))
    assert len(names) > 40, names
    def family(name):
        base = name.split("<", 1)[0]
        if base == "biquad_kernel" and "segmented" in name:
            return "biquad_kernel<segmented>"          # the lane-walk time-segmented form: its own launch path
        return base
    families = {family(n) for n in names}
    assert families == set(T.FAMILIES), (families ^ set(T.FAMILIES))
    for fam, (_, prefix) in T.FAMILIES.items():
        assert prefix.startswith(fam.split("<", 1)[0]), (fam, prefix)
    src = open(T.__file__).read()
    assert not conftest.AB_ONLY_RE.search(src)
    assert "monkeypatch" not in src and "environ" not in src
