"""CPU oracle for the timemachine force-evaluation + Langevin-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``timemachine_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.  The product
path is HIP-only and fails loudly when the extension is missing.

What is restated here (each function cites the reference file:line it follows; paths are relative to
the reference checkout):

* ``ref_potentials``  -- the reference's JAX potentials (``timemachine/potentials/nonbonded.py``,
  ``bonded.py``) restated with torch-f64 so that du/dx and du/dp come from autograd
  (the reference uses ``jax.grad`` of the same energy functions, ``tests/common.py:234-334``).
* ``fixed_point``     -- the integer model of ``cpp/src/kernels/k_fixed_point.cuh`` /
  ``cpp/src/fixed_point.hpp`` (bit-exact).
* ``hilbert``         -- the 128^3 Hilbert LUT, key and stable permutation of
  ``cpp/src/hilbert_sort.cu`` + ``cpp/src/kernels/k_hilbert.cu`` (bit-exact).
* ``nblist``          -- block bounds and brute-force per-row-block interaction sets
  (``tests/test_nblist.py:28-56,117-140``).
* ``integrator``      -- BAOAB Langevin step (``timemachine/integrator.py:124-150``) and the mixed
  precision model of ``cpp/src/kernels/k_integrator.cuh:5-62``.

Pinning: the reference holds no known-answer vectors for this path ("Golden vectors:
none").  The oracle is therefore pinned against the reference itself run in the build container:
``tests/golden/generate_golden.py`` imports the reference's Python potentials / integrator from
/root/reference (under a numpy shim for the absent ``jax`` package), checks this restatement against
them, and writes ``tests/golden/*.npz``.  The vendored C Hilbert code is compiled by
``oracle/Makefile`` into ``oracle/_ref/`` and pins ``hilbert`` bit-for-bit.
"""
