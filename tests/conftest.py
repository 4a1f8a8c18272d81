import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # glibc's own fatal messages (heap-consistency aborts: "free(): invalid pointer" ...) go to /dev/tty unless told
    # otherwise -- nowhere, on a box without a terminal
    os.environ.setdefault("LIBC_FATAL_STDERR_", "1")


@pytest.fixture(scope="session", autouse=True)
def _native_crash_evidence(request):
    """A native abort inside a test (a GPU memory fault ends in abort() on a runtime thread) must leave its evidence
    in the log: pytest captures fd 2 per test, so the runtime's one line about the fault dies with the process.  The
    library's handler (cals_hip_debug_install_crash_trace) copies the captured stderr of the running test and a native
    backtrace to the REAL stderr, saved here with the capture suspended.  CALS_CRASH_TRACE=0 switches it off."""
    if os.environ.get("CALS_CRASH_TRACE", "1") == "0":
        yield
        return
    try:
        import cp_cals_amd
        lib = cp_cals_amd.load_library()
    except Exception:  # library not built: the tests that need it say so themselves
        yield
        return
    capman = request.config.pluginmanager.getplugin("capturemanager")
    if capman is not None:
        with capman.global_and_fixture_disabled():
            fd = os.dup(2)
    else:
        fd = os.dup(2)
    lib.cals_hip_debug_install_crash_trace(fd)
    yield


@pytest.fixture(scope="session")
def oracle():
    import oracle as O  # oracle/oracle.py (test infrastructure)
    O.lib()
    return O


@pytest.fixture(scope="session")
def cc():
    import cp_cals_amd
    return cp_cals_amd


@pytest.fixture(scope="session")
def inputs():
    from cp_cals_amd import inputs as I
    return I
