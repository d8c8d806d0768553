import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(REPO, "tests", "golden")


def load_binding(name):
    """The custom_ops module of one binding, whatever TM_AMD_BINDING selected as `timemachine_amd.lib.custom_ops`:
    "pybind11" = the compiled module (csrc/wrap_custom_ops.cpp), "ctypes" = its ctypes twin.  Both sit on the same C ABI."""
    import importlib
    import importlib.util
    import sysconfig

    import timemachine_amd.lib as lib

    if name == "ctypes":
        return importlib.import_module("timemachine_amd.lib.custom_ops_ctypes")
    assert name == "pybind11", name
    if getattr(lib.custom_ops, "BINDING", None) == "pybind11":
        return lib.custom_ops
    path = os.path.join(os.path.dirname(lib.__file__), "custom_ops" + sysconfig.get_config_var("EXT_SUFFIX"))
    spec = importlib.util.spec_from_file_location("timemachine_amd.lib.custom_ops", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session", params=["pybind11", "ctypes"])
def any_binding(request):
    """CPU tests of the boundary run against both bindings"""
    return load_binding(request.param)


NEVER = 2**31 - 1  # (int: both debug knobs take a C int)
# which list and which force kernel a Nonbonded potential runs on: (static complete list up to K atoms, row-block kernel from K atoms)
NB_PATHS = {"static+items": (4608, NEVER), "listed+items": (0, NEVER), "listed+rowblocks": (0, 0), "static+rowblocks": (4608, 0)}


@pytest.fixture(params=list(NB_PATHS))
def nb_path(request, co):
    """Small systems get a static complete list by default, large ones a built neighbor list; forces-only launches can run the
    wave-per-item kernel (the product) or the row-block kernel (csrc/kernels_nonbonded_rowblock.hip.hpp, an independent second
    implementation): the golden comparisons below run on every combination, whatever the size.  The row-block kernel lives in the
    variant library libtimemachine_amd_rowblock.so only: its combinations are skipped under the product library and run when
    tests/test_gpu_second_binding.py re-runs these tests in a child interpreter that loads the variant (TM_AMD_LIB)."""
    static_k, rowblock_k = NB_PATHS[request.param]
    if rowblock_k != NEVER and not co.debug_rowblock_available():
        pytest.skip("the row-block kernel is in the variant library only (run through tests/test_gpu_second_binding.py)")
    before = co.debug_set_static_list_max_k(static_k), co.debug_set_rowblock_min_k(rowblock_k)
    yield request.param
    co.debug_set_static_list_max_k(before[0])
    co.debug_set_rowblock_min_k(before[1])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """With TM_AMD_LIB pointing at a -DTM_GUARD build (guard zones around every device buffer), the whole GPU suite doubles
    as an out-of-bounds-write detector: report what the guard zones caught once every test has run."""
    if "guard" not in os.path.basename(os.environ.get("TM_AMD_LIB", "")):
        return
    from timemachine_amd.lib import custom_ops

    n = custom_ops.debug_check_guards()
    print(f"\n[guard build] device-buffer guard violations over the session: {n}")
    if n:
        session.exitstatus = 1
