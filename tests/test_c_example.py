"""The C ABI from plain C: examples/fir_stream.c compiled with gcc against include/pipe_hip.h and
libpipe_hip.so, no Python in the data path.  CPU part: the headers are valid C99 and the example
builds and links; GPU part: its output stream equals the oracle's bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from pipe_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "pipe_amd", "lib")


def build(tmp_path):
    exe = str(tmp_path / "fir_stream")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "fir_stream.c"), "-L", LIBDIR, "-lpipe_hip",
                           f"-Wl,-rpath,{LIBDIR}", "-o", exe])
    return exe


@pytest.mark.parametrize("header", ["pipe_hip.h", "pipe_host.h"])
def test_headers_are_c99(header):
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", header)])


def test_c_example_builds_and_links(tmp_path):
    assert os.path.exists(build(tmp_path))


@pytest.mark.gpu
def test_c_example_stream_equals_oracle(tmp_path):
    exe = build(tmp_path)
    out = tmp_path / "out.f64"
    res = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert "channels=2 rate=1/1 frames=5320" in res.stdout
    got = np.fromfile(out, dtype=np.float64)
    F, C, N = 512, 2, 64
    x = synth.samples(synth.line_seed(0), 0, 5320 * C).reshape(5320, C)
    want = O.Fir(np.full(N, 1.0 / N), C).process(x)
    assert np.array_equal(got, want.ravel())
