"""pipe_amd -- MI355X-native Processor stage execution for pipelined.dev/pipe.

The product is libpipe_hip.so (pipe_amd/csrc, C ABI in include/pipe_hip.h) and
its host-side mirror of the reference's Line/Source/Processor/Sink API
(pipe_amd/csrc/host).  This Python package is the test/bench harness around the
C ABI; it performs no computation of its own and has no CPU fallback.
"""
from . import synth  # noqa: F401  (pure definitions, importable without the library)

__all__ = ["synth", "load"]


def load():
    """Load libpipe_hip.so (raises ImportError if it has not been built)."""
    from . import _lib
    return _lib.lib()
