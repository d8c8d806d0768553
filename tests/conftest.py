import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(REPO, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """With TM_AMD_LIB pointing at a -DTM_GUARD build (guard zones around every device buffer), the whole GPU suite doubles
    as an out-of-bounds-write detector: report what the guard zones caught once every test has run."""
    if "guard" not in os.path.basename(os.environ.get("TM_AMD_LIB", "")):
        return
    import ctypes

    from timemachine_amd.lib import custom_ops

    n = ctypes.c_int(0)
    custom_ops._check(custom_ops._lib.tm_debug_check_guards(ctypes.byref(n)))
    print(f"\n[guard build] device-buffer guard violations over the session: {n.value}")
    if n.value:
        session.exitstatus = 1
