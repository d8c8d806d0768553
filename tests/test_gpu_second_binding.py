"""The ctypes twin of the boundary (timemachine_amd/lib/custom_ops_ctypes.py) on the GPU.

The product binding is the compiled pybind11 module; the whole `-m gpu` suite runs through it.  The ctypes module mirrors the same
surface over the same C ABI and is what A/B measurements of variant libraries use (TM_AMD_LIB=<variant .so> implies it: the
compiled module is linked against the product library).  So that it is more than a CPU-tested mirror, a subset of the GPU parity
tests -- golden vectors of the nonbonded family, config 2 with every term, BoundPotential + batches, a deterministic Context run
against the oracle -- runs through it here, in a child interpreter started with TM_AMD_BINDING=ctypes (the binding is chosen at
import time, once per process)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUBSET = "test_nonbonded_golden or test_config2_all_terms or test_context_deterministic_steps_match_oracle or test_bound_potential_and_batches"


@pytest.mark.gpu
def test_gpu_parity_subset_through_the_ctypes_binding():
    env = dict(os.environ, TM_AMD_BINDING="ctypes")
    env.pop("TM_AMD_LIB", None)
    which = subprocess.run(
        [sys.executable, "-c", "from timemachine_amd.lib import custom_ops as co; print(getattr(co, 'BINDING', '?'), co.device_count())"],
        cwd=REPO, env=env, capture_output=True, text=True, timeout=300,
    )
    assert which.returncode == 0, which.stderr[-2000:]
    binding, devices = which.stdout.split()
    assert binding == "ctypes" and int(devices) >= 1
    r = subprocess.run(
        [sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", SUBSET],
        cwd=REPO, env=env, capture_output=True, text=True, timeout=900,
    )
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "failed" not in r.stdout
    # ... and the randomized tests: every entry point a window's caller has (Context with movers, local MD, setters, batches dense
    # and sparse, set_atom_idxs, grouped stepping) and every potential class, through the ctypes marshalling
    r = subprocess.run(
        [sys.executable, "-m", "pytest", "tests/test_gpu_interleavings.py", "tests/test_gpu_random_parity.py", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"],
        cwd=REPO, env=env, capture_output=True, text=True, timeout=900,
    )
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "failed" not in r.stdout


ROWBLOCK_LIB = os.path.join(REPO, "timemachine_amd", "csrc", "libtimemachine_amd_rowblock.so")
ROWBLOCK_SUBSET = "rowblocks or test_dhfr_shaped_box_all_terms or test_md_on_the_row_block_kernel"


@pytest.mark.gpu
def test_row_block_kernel_of_the_variant_library_gives_the_product_kernels_bits():
    """The row-block kernel (csrc/kernels_nonbonded_rowblock.hip.hpp) is a second, independent implementation of the tile kernel;
    it is NOT in the product library.  csrc/build.py builds libtimemachine_amd_rowblock.so (the product objects + nonbonded.hip with
    -DTM_ROWBLOCK); a child interpreter loads it (TM_AMD_LIB implies the ctypes binding) and runs every golden comparison of the
    `nb_path` matrix that selects the kernel (static and listed neighbor lists), the DHFR-shaped box (forces-only == full call on
    BOTH kernels, sampled oracle forces) and MD runs with the kernel on every forces-only launch against the same runs on the
    product kernel, bit for bit."""
    assert os.path.exists(ROWBLOCK_LIB), "csrc/build.py builds the variant library next to the product"
    env = dict(os.environ, TM_AMD_LIB=ROWBLOCK_LIB, TM_AMD_ROWBLOCK_MIN_K="2147483647")
    env.pop("TM_AMD_BINDING", None)
    which = subprocess.run(
        [sys.executable, "-c", "from timemachine_amd.lib import custom_ops as co; print(getattr(co, 'BINDING', '?'), co.debug_rowblock_available())"],
        cwd=REPO, env=env, capture_output=True, text=True, timeout=300,
    )
    assert which.returncode == 0, which.stderr[-2000:]
    assert which.stdout.split() == ["ctypes", "True"], which.stdout
    r = subprocess.run(
        [sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_edge_cases.py", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", ROWBLOCK_SUBSET],
        cwd=REPO, env=env, capture_output=True, text=True, timeout=1500,
    )
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "failed" not in r.stdout
