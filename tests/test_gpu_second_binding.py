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
