"""Run a piece of test code in a child interpreter with a wall-clock limit of its own.

Tests that drive the library from several threads (the async host loop: a thread per component, run.go:171-196) or
that hammer one path for seconds run here: if native code never returns, the child's process GROUP is killed and the
test fails with the child's output -- one named failure, the rest of the suite goes on (one hang in the
parent erased 396 results)."""
import os
import signal
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_child(code: str, timeout_s: float = 90.0, env: dict | None = None) -> str:
    """`code` runs with the repo root on sys.path; it must print CHILD-OK as its last act."""
    e = dict(os.environ)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    e["PYTHONFAULTHANDLER"] = "1"
    # (a hang in native code names itself: the library's watchdog prints where every thread stands, abi.hip)
    e.setdefault("PIPE_HIP_STALL_DUMP_MS", "8000")
    if env:
        e.update(env)
    prog = "import faulthandler, sys; faulthandler.enable()\n" + textwrap.dedent(code) + "\nprint('CHILD-OK', flush=True)\n"
    p = subprocess.Popen([sys.executable, "-c", prog], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGABRT)  # (faulthandler prints every thread's stack on SIGABRT)
            out, _ = p.communicate(timeout=10)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            out, _ = p.communicate()
        raise AssertionError(f"child did not finish within {timeout_s} s; its output:\n{out[-6000:]}")
    if p.returncode != 0 or "CHILD-OK" not in out:
        raise AssertionError(f"child failed (rc {p.returncode}); its output:\n{out[-6000:]}")
    return out
