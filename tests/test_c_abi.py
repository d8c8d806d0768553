"""CPU tests of the drop-in boundary: the shared library loads without a GPU, exports every symbol that
include/timemachine_amd.h declares, and the host-only entry points (validation, LUT, fixed-point helpers) behave."""
import ctypes
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "timemachine_amd.h")
LIB = os.path.join(REPO, "timemachine_amd", "csrc", "libtimemachine_amd.so")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build the extension first: python -m timemachine_amd.csrc.build"
    lib = ctypes.CDLL(LIB)
    syms = declared_symbols()
    assert len(syms) > 60
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_header_cites_reference_interfaces():
    text = open(HEADER).read()
    assert text.count("wrap_kernels.cpp") >= 20  # every entry point names the binding it replaces


def test_host_only_entry_points(any_binding):
    assert any_binding.FIXED_EXPONENT == 0x1000000000

    class custom_ops:  # the C ABI itself, through ctypes
        _lib = ctypes.CDLL(LIB)
        _I128 = np.dtype([("lo", np.uint64), ("hi", np.int64)])

    custom_ops._lib.tm_fixed_to_float.restype = ctypes.c_double
    custom_ops._lib.tm_fixed_to_float.argtypes = [ctypes.c_uint64]
    custom_ops._lib.tm_energy_to_float.restype = ctypes.c_double
    assert custom_ops._lib.tm_fixed_to_float(ctypes.c_uint64(1 << 36)) == 1.0
    assert custom_ops._lib.tm_fixed_to_float(ctypes.c_uint64((1 << 64) - (1 << 35))) == -0.5
    rec = np.zeros(1, dtype=custom_ops._I128)
    rec["lo"], rec["hi"] = (1 << 63) - 1, 0  # == LLONG_MAX -> overflowed -> NaN
    assert custom_ops._lib.tm_energy_overflowed(rec.ctypes.data_as(ctypes.c_void_p)) == 1
    assert np.isnan(custom_ops._lib.tm_energy_to_float(rec.ctypes.data_as(ctypes.c_void_p)))
    rec["lo"], rec["hi"] = 3 << 35, 0
    assert custom_ops._lib.tm_energy_to_float(rec.ctypes.data_as(ctypes.c_void_p)) == 1.5
    # 128-bit negative value
    rec["lo"], rec["hi"] = (1 << 64) - (1 << 36), -1
    assert custom_ops._lib.tm_energy_to_float(rec.ctypes.data_as(ctypes.c_void_p)) == -1.0


def test_hilbert_lut_of_the_library_is_bit_exact(any_binding):
    from oracle import hilbert as ohilbert

    custom_ops = any_binding

    np.testing.assert_array_equal(custom_ops.hilbert_lut(), ohilbert.lut())


def test_constructor_validation_messages_match_reference(any_binding):
    """Messages are part of the contract (reference tests regex-match them); all raised before any GPU work."""
    custom_ops = any_binding

    with pytest.raises(RuntimeError, match="Neighborlist N must be at least 1"):  # tests/test_nblist.py:22-25
        custom_ops.Neighborlist_f32(0)
    with pytest.raises(RuntimeError, match="src == dst"):
        custom_ops.HarmonicBond_f64(np.array([[3, 3]], dtype=np.int32))
    with pytest.raises(RuntimeError, match="angle triplets must be unique"):
        custom_ops.HarmonicAngle_f32(np.array([[0, 1, 0]], dtype=np.int32))
    with pytest.raises(RuntimeError, match="torsion quads must be unique"):
        custom_ops.PeriodicTorsion_f32(np.array([[0, 1, 2, 0]], dtype=np.int32))
    with pytest.raises(RuntimeError, match="illegal pair with src == dst: 1, 1"):
        custom_ops.NonbondedExclusions_f32(np.array([[1, 1]], dtype=np.int32), np.ones((1, 2)), 2.0, 1.2)
    with pytest.raises(RuntimeError, match="expected same number of pairs and scale tuples, but got 1 != 2"):
        custom_ops.NonbondedPairList_f64(np.array([[0, 1]], dtype=np.int32), np.ones((2, 2)), 2.0, 1.2)
    with pytest.raises(RuntimeError, match="number of potentials != number of parameter sizes"):
        custom_ops.SummedPotential([], [3])
    with pytest.raises(TypeError):  # unsafe cast, as pybind's py::array_t<int, c_style> overload resolution
        custom_ops.HarmonicBond_f32(np.array([[0.5, 1.5]]))
    with pytest.raises(TypeError):
        custom_ops.Potential()
    with pytest.raises(NotImplementedError):
        custom_ops.BDExchangeMove_f32()  # entries of the reference module that are not built fail loudly, by name


def test_dataclass_field_order_is_constructor_order():
    """to_gpu() calls custom_ops.<Name>_<prec>(*astuple(self)) (reference potentials/potential.py:28-37)."""
    from dataclasses import fields

    from timemachine_amd import potentials as P
    from timemachine_amd.lib import LangevinIntegrator

    assert [f.name for f in fields(P.Nonbonded)] == ["num_atoms", "exclusion_idxs", "scale_factors", "beta", "cutoff", "atom_idxs", "disable_hilbert_sort", "nblist_padding"]
    assert [f.name for f in fields(P.NonbondedAllPairs)] == ["num_atoms", "beta", "cutoff", "atom_idxs", "disable_hilbert_sort", "nblist_padding"]
    assert [f.name for f in fields(P.NonbondedExclusions)] == ["idxs", "rescale_mask", "beta", "cutoff"]
    assert [f.name for f in fields(P.NonbondedPairList)] == ["idxs", "rescale_mask", "beta", "cutoff"]
    assert [f.name for f in fields(P.HarmonicBond)] == ["idxs"]
    assert [f.name for f in fields(P.SummedPotential)] == ["potentials", "params_init", "parallel"]
    assert [f.name for f in fields(LangevinIntegrator)] == ["temperature", "dt", "friction", "masses", "seed"]
    assert P.HarmonicBond._custom_ops_class_name(np.float32) == "HarmonicBond_f32"
    assert P.NonbondedExclusions._custom_ops_class_name(np.float64) == "NonbondedExclusions_f64"
    with pytest.raises(ValueError, match="invalid precision"):
        P.HarmonicBond._custom_ops_class_name(np.float16)


def test_filter_exclusions_matches_reference_fixture():
    """filter_exclusions against the outputs of the reference's own function (timemachine/potentials/nonbonded.py:176-218),
    recorded by tests/golden/generate_golden_next.py: three atom sets (a shuffled subset, everything, a single atom =>
    empty result with the reference's shapes) x update_idxs in {False, True}; and against the oracle's restatement."""
    import os

    from oracle import ref_potentials as rp
    from timemachine_amd.potentials import filter_exclusions

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "filter_exclusions.npz"))
    excl, scales = g["exclusion_idxs"], g["scale_factors"]
    for k in range(3):
        atom_idxs = g[f"atom_idxs_{k}"]
        for upd in (False, True):
            idxs, sc = filter_exclusions(atom_idxs, excl, scales, update_idxs=upd)
            assert idxs.dtype == np.int32 and idxs.shape == g[f"idxs_{k}_{int(upd)}"].shape
            np.testing.assert_array_equal(idxs, g[f"idxs_{k}_{int(upd)}"])
            np.testing.assert_array_equal(sc, g[f"scales_{k}_{int(upd)}"])
        c, d = rp.filter_exclusions(atom_idxs, excl, scales)
        np.testing.assert_array_equal(c, g[f"idxs_{k}_0"].reshape(-1, 2))
        np.testing.assert_array_equal(d, g[f"scales_{k}_0"])
    e, f = filter_exclusions(g["atom_idxs_2"], excl, scales)
    assert e.size == 0 and f.shape == (0, 2)


def test_committed_bench_lines_follow_the_contract():
    """The bench lines kept under profiles/ carry every field the driver and the judge read (metric, whole-job value,
    roofline with live kernel time + PMC traffic, CPU baseline with its sample description)."""
    import glob
    import json

    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r0*_v*_bench.json")))
    assert paths, "no committed bench line"
    d = json.load(open(paths[-1]))
    hrex = sorted(glob.glob(os.path.join(os.path.dirname(paths[-1]), "r0*_bench_hrex.json")))
    if hrex:  # the replica-exchange mode reports the same top-level fields
        h = json.load(open(hrex[-1]))
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
            assert key in h, key
        assert "workload" in h["config"] and h["scaling"] == "strong"
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "ns/day" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port")
    # value == steps / time: ns/day from ms per step at 2.5 fs
    assert abs(d["value"] - d["n_gpus"] * 86400.0 * 2.5e-6 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    if os.path.basename(paths[-1]) >= "r04":  # (round 4 on) the line tracks the whole step and what the host spends on it
        names = {k["name"] for k in d["kernels"]}
        assert {"nonbonded_tiles", "nblist_build", "integrator_update"} <= names
        for k in d["kernels"]:
            for key in ("us_per_step", "launches_per_step", "us_per_launch", "algorithmic_bytes_per_step", "frac_of_hbm_peak", "share_of_step"):
                assert key in k and k[key] >= 0, (k["name"], key)
        # three kernels make the step.  Up to round 4 the shares were us of the PROFILED run over the TIMED step (they summed to
        # 1.12); from round 5 on both come from the profiled run -- one clock -- and the rest of that step is the profiler's own
        # event records between the launches: the shares sum to less than one
        total = sum(k["share_of_step"] for k in d["kernels"])
        if os.path.basename(paths[-1]) >= "r05":
            assert 0.7 < total <= 1.0 and d["kernels_profiled_us_per_step"] > 1e3 * d["ms_per_step"]
        else:
            assert 0.9 < total < 1.3
        for key in ("host_cpu_us_per_step", "cpu_quota", "host_cpu_load"):
            assert key in d and d[key] > 0, key
        for key in ("traffic_build_stamp", "library_build_stamp", "traffic_stale"):
            assert key in r, key


def test_es_force_table_matches_the_analytic_function():
    """The f64 kernels' tabulated electrostatic force factor F(d^2) = (D'(d)/d - D(d)/d^2)/d, D = erfc(beta d) S(d)
    (csrc/nb_es_table.hip.hpp) against the analytic form in numpy/scipy: <= 1e-11 relative over d in [0.0884, 1.2) nm, exactly
    the interval layout the device indexes with the bits of d^2."""
    from scipy.special import erfc

    from timemachine_amd.lib import custom_ops

    assert custom_ops.BINDING == os.environ.get("TM_AMD_BINDING", "pybind11").lower()  # the product binding is the compiled module
    for beta in (2.0, 2.6):
        tab = custom_ops.es_force_table(beta)
        assert tab.shape == (256, 6) and np.all(np.isfinite(tab))
        rng = np.random.default_rng(0)
        s = np.exp(rng.uniform(np.log(2.0**-7), np.log(1.44), 200000))
        s = np.concatenate([s, [2.0**-7, 0.01, 0.25, 1.0, np.nextafter(1.44, 0)]])
        bits = s.view(np.uint64)
        idx = (bits >> np.uint64(47)).astype(np.int64) - ((1023 - 7) << 5)
        frac = ((bits & np.uint64((1 << 47) - 1)) << np.uint64(5)) | np.uint64(0x3FF0000000000000)
        t = frac.view(np.float64) - 1.0
        assert idx.min() >= 0 and idx.max() < 256 and t.min() >= 0 and t.max() < 1
        c = tab[idx]
        p = c[:, 5]
        for k in (4, 3, 2, 1, 0):
            p = p * t + c[:, k]
        d = np.sqrt(s)
        q = (d / 1.2) ** 8
        S = np.cos(0.5 * np.pi * q) ** 3
        dS = -12 * np.pi / 1.2**8 * d**7 * np.sin(0.5 * np.pi * q) * np.cos(0.5 * np.pi * q) ** 2
        e = erfc(beta * d)
        de = -2 * beta / np.sqrt(np.pi) * np.exp(-((beta * d) ** 2))
        F = ((e * dS + de * S) / d - e * S / d**2) / d
        # F -> 0 at the end of the switch: an absolute floor there (1e-10 in F = 6e-9 kJ/mol/nm on a water O-H pair)
        assert np.all(np.abs(p - F) < 3e-11 * np.abs(F) + 1e-10)
        assert np.max(np.abs(p - F)[s < 1.0] / np.abs(F)[s < 1.0]) < (6e-12 if beta == 2.0 else 3e-11)
        # the energy factor G(d^2) = erfc(beta d) S(d) / d of the calls that ask for energies / du/dp: same layout, same bound
        gtab = custom_ops.es_energy_table(beta)
        assert gtab.shape == (256, 6) and np.all(np.isfinite(gtab))
        c = gtab[idx]
        p = c[:, 5]
        for k in (4, 3, 2, 1, 0):
            p = p * t + c[:, k]
        G = e * S / d
        assert np.all(np.abs(p - G) < 3e-11 * np.abs(G) + 1e-11)
        assert np.max(np.abs(p - G)[s < 1.0] / np.abs(G)[s < 1.0]) < (6e-12 if beta == 2.0 else 3e-11)


def test_loading_the_library_exports_the_hardware_queue_count():
    """Streams of contexts stepped together (tm_context_multiple_steps_group) need hardware queues of their own; the HIP runtime reads
    GPU_MAX_HW_QUEUES when it first touches the device.  The LIBRARY exports 8 from a constructor when it is loaded -- a C-ABI consumer
    that never imports the Python package gets it too (include/timemachine_amd.h) -- and leaves a value the user chose alone."""
    import subprocess
    import sys

    probe = (
        "import ctypes, sys\n"
        "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
        "before = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
        f"ctypes.CDLL({LIB!r})\n"
        "after = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
        "print(before, after)\n"
    )
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["None", "b'8'"], out.stdout
    out = subprocess.run([sys.executable, "-c", probe], env=dict(env, GPU_MAX_HW_QUEUES="3"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split() == ["b'3'", "b'3'"], (out.stdout, out.stderr[-500:])
    # ... and the Python package itself does not touch the environment any more
    text = open(os.path.join(REPO, "timemachine_amd", "__init__.py")).read()
    assert "environ" not in text


def test_api_lock_is_per_device_not_per_process():
    """The C ABI's threading contract (include/timemachine_amd.h, "Threading"): an entry point holds the lock of the calling thread's
    current DEVICE.  Two host threads on two devices are inside the ABI at the same time; two on one device never are.  No GPU here:
    tm_debug_set_thread_lock_device stands in for hipSetDevice, tm_debug_hold_api_lock for a long call (reference contract:
    cpp/src/potential.hpp:7 -- not thread-safe per OBJECT, which lives on one device)."""
    import ctypes
    import threading

    from timemachine_amd.csrc import build

    lib = ctypes.CDLL(build.LIB)

    def run(devices):
        reset = ctypes.c_int(0)
        assert lib.tm_debug_hold_api_lock(ctypes.c_int(-1), ctypes.byref(reset)) == 0  # (a negative duration resets the high-water mark)
        seen = [0] * len(devices)
        barrier = threading.Barrier(len(devices))

        def worker(k):
            assert lib.tm_debug_set_thread_lock_device(ctypes.c_int(devices[k])) == 0
            barrier.wait()
            out = ctypes.c_int(0)
            assert lib.tm_debug_hold_api_lock(ctypes.c_int(150), ctypes.byref(out)) == 0
            seen[k] = out.value
            lib.tm_debug_set_thread_lock_device(ctypes.c_int(-1))

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(devices))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        return max(seen)

    assert run([0, 1]) == 2  # two devices: both threads inside at once
    assert run([3, 3]) == 1  # one device: one after the other
    assert run([0, 1, 1]) == 2
