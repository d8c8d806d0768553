"""``timemachine_amd.lib.custom_ops_ctypes`` -- the ctypes twin of the compiled ``custom_ops`` module (selected with
TM_AMD_BINDING=ctypes, or implied by TM_AMD_LIB): the reference's ``timemachine.lib.custom_ops`` surface for the force-evaluation +
Langevin-step hot path, served by ``libtimemachine_amd.so`` (hand-written HIP for gfx950) through its C ABI
(``include/timemachine_amd.h``).

Same class names, constructor argument order, method names, defaults, return shapes and error messages as the pybind11
module the reference builds from ``timemachine/cpp/src/wrap_kernels.cpp`` (the lines are cited per class below), so
reference-style callers (``potentials.*.to_gpu``, ``lib.LangevinIntegrator.impl``, ``Context(...).multiple_steps``)
work by changing only the top-level import.  This module is deliberately thin: validation that the pybind lambdas do,
ctypes marshalling, fixed-point -> float conversion of the returned accumulators.  There is NO CPU fallback: importing
this module without the compiled library raises ImportError, and every method runs on the GPU.
"""
import ctypes
import inspect
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("TM_AMD_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libtimemachine_amd.so")

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"{_LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
        "or python -m timemachine_amd.csrc.build). timemachine_amd has no CPU fallback."
    )
_lib = ctypes.CDLL(_LIB_PATH)

FIXED_EXPONENT = 0x1000000000  # wrap_kernels.cpp:2144
BINDING = "ctypes"

TM_OK, TM_ERR_RUNTIME, TM_ERR_INVALID_HARDWARE = 0, 1, 2
_F32, _F64 = 0, 1


class InvalidHardware(Exception):
    """No usable GPU / driver (reference: custom_ops.InvalidHardware, wrap_kernels.cpp:2311)."""


_lib.tm_last_error.restype = ctypes.c_char_p
_lib.tm_version.restype = ctypes.c_char_p
_lib.tm_fixed_to_float.restype = ctypes.c_double
_lib.tm_fixed_to_float.argtypes = [ctypes.c_uint64]
_lib.tm_energy_to_float.restype = ctypes.c_double

_vp = ctypes.c_void_p
_c_int = ctypes.c_int
_c_double = ctypes.c_double


def _check(code):
    if code == TM_OK:
        return
    msg = _lib.tm_last_error().decode()
    if code == TM_ERR_INVALID_HARDWARE:
        raise InvalidHardware(msg)
    raise RuntimeError(msg)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _as(a, dtype, what):
    """py::array_t<T, c_style> semantics: safe casts are converted, unsafe ones are a TypeError.

    Python sequences and scalars (not ndarrays) are converted element by element, as numpy does
    for the binding layer of the reference: a list of Python ints becomes int32/uint32 when every
    value fits (the reference's own tests pass `[0]` and `[(0, 0)]` as index arrays), a list holding
    a float does not become an integer array."""
    if not isinstance(a, np.ndarray):
        probe = np.asarray(a)
        target = np.dtype(dtype)
        if probe.dtype.kind in "iub" and target.kind in "iu":
            # explicit range check: numpy 1.x wraps out-of-range Python ints silently ([-1] -> 4294967295 for uint32)
            info = np.iinfo(target)
            if probe.dtype == object or (probe.size and (int(probe.min()) < info.min or int(probe.max()) > info.max)):
                raise TypeError(f"{what}: value out of range for {target}")
            return np.ascontiguousarray(probe.astype(target))
        arr = probe
    else:
        arr = a
    if arr.dtype != dtype and not np.can_cast(arr.dtype, dtype, "safe"):
        raise TypeError(f"{what}: incompatible array dtype {arr.dtype}, expected {np.dtype(dtype)}")
    return np.ascontiguousarray(arr, dtype=dtype)


def _f64(a, what="array"):
    return _as(a, np.float64, what)


def _i32(a, what="index array"):
    return _as(a, np.int32, what)


def _u32(a, what="index array"):
    return _as(a, np.uint32, what)


def _verify_coords(coords):
    # wrap_kernels.cpp:51-59
    if coords.ndim != 2:
        raise RuntimeError("coords dimensions must be 2")
    if coords.shape[-1] != 3:
        raise RuntimeError("coords must have a shape that is 3 dimensional")


def _verify_coords_and_box(coords, box):
    # wrap_kernels.cpp:62-78
    _verify_coords(coords)
    if box.ndim != 2 or box.shape[0] != 3 or box.shape[1] != 3:
        raise RuntimeError("box must be 3x3")
    flat = box.reshape(-1)
    for i in range(9):
        if i in (0, 4, 8):
            if flat[i] <= 0.0:
                raise RuntimeError("box must have positive values along diagonal")
        elif flat[i] != 0.0:
            raise RuntimeError("box must be ortholinear")


def _fixed_to_float(u64_arr):
    """FIXED_TO_FLOAT<double>, cpp/src/fixed_point.hpp:18-20."""
    return u64_arr.view(np.int64).astype(np.float64) / float(FIXED_EXPONENT)


_I128 = np.dtype([("lo", np.uint64), ("hi", np.int64)])


def _i128_to_int(rec):
    return (int(rec["hi"]) << 64) | int(rec["lo"])


_LLONG_MAX = (1 << 63) - 1
_LLONG_MIN = -(1 << 63)


def _energy_to_float(rec):
    """convert_energy_to_fp, wrap_kernels.cpp:83-89: NaN when the 128-bit sum left the int64 range."""
    v = _i128_to_int(rec)
    if v >= _LLONG_MAX or v <= _LLONG_MIN:
        return float("nan")
    return float(v) / float(FIXED_EXPONENT)


def cuda_device_reset():
    """wrap_kernels.cpp:2222-2225 (name kept for drop-in compatibility; resets the HIP device)."""
    _check(_lib.tm_device_reset())


def device_count():
    n = _c_int(0)
    _check(_lib.tm_device_count(ctypes.byref(n)))
    return n.value


def set_device(idx):
    _check(_lib.tm_set_device(int(idx)))


def device_synchronize():
    _check(_lib.tm_device_synchronize())


def device_name():
    buf = ctypes.create_string_buffer(256)
    _check(_lib.tm_device_name(buf, ctypes.c_size_t(256)))
    return buf.value.decode()


# ---------------------------------------------------------------------------------------------------------------
class Potential:
    """Base of every potential (wrap_kernels.cpp:731-1131).  Not constructible."""

    _h = None
    _keep = ()

    def __init__(self, *a, **k):
        raise TypeError("Potential: No constructor defined!")

    @classmethod
    def _wrap(cls, handle, keep=()):
        obj = object.__new__(cls)
        obj._h = handle
        obj._keep = tuple(keep)
        return obj

    def __del__(self):
        h, self._h = self._h, None
        if h is not None and _lib is not None:
            _lib.tm_potential_destroy(h)

    def _du_dp_to_float(self, N, P, fixed, shape):
        out = np.empty(shape, dtype=np.float64)
        _check(_lib.tm_potential_du_dp_fixed_to_float(self._h, _c_int(N), _c_int(P), _ptr(fixed), _ptr(out)))
        return out

    def execute(self, coords, params, box, compute_du_dx=True, compute_du_dp=True, compute_u=True):
        """-> (du_dx[N,3] | None, du_dp[params.shape] | None, u | None); wrap_kernels.cpp:1039-1105."""
        coords, params, box = _f64(coords, "coords"), _f64(params, "params"), _f64(box, "box")
        N, P = coords.shape[0], params.size
        _verify_coords_and_box(coords, box)
        du_dx = np.full(N * 3, 9999, dtype=np.uint64) if compute_du_dx else None
        du_dp = np.full(P, 9999, dtype=np.uint64) if compute_du_dp else None
        u = np.zeros(1, dtype=_I128) if compute_u else None
        _check(_lib.tm_potential_execute(self._h, _c_int(N), _c_int(P), _ptr(coords), _ptr(params), _ptr(box), _ptr(du_dx), _ptr(du_dp), _ptr(u)))
        r_dx = _fixed_to_float(du_dx).reshape(N, 3) if compute_du_dx else None
        r_dp = self._du_dp_to_float(N, P, du_dp, params.shape) if compute_du_dp else None
        r_u = _energy_to_float(u[0]) if compute_u else None
        return r_dx, r_dp, r_u

    def execute_raw(self, coords, params, box, compute_du_dx=True, compute_du_dp=True, compute_u=True):
        """The un-converted accumulators: (uint64[N,3] | None, uint64[P] | None, python int | None).  Not part of the
        reference surface; used by the parity tests for bit-exact (integer) comparisons."""
        coords, params, box = _f64(coords, "coords"), _f64(params, "params"), _f64(box, "box")
        N, P = coords.shape[0], params.size
        _verify_coords_and_box(coords, box)
        du_dx = np.zeros(N * 3, dtype=np.uint64) if compute_du_dx else None
        du_dp = np.zeros(P, dtype=np.uint64) if compute_du_dp else None
        u = np.zeros(1, dtype=_I128) if compute_u else None
        _check(_lib.tm_potential_execute(self._h, _c_int(N), _c_int(P), _ptr(coords), _ptr(params), _ptr(box), _ptr(du_dx), _ptr(du_dp), _ptr(u)))
        return (du_dx.reshape(N, 3) if compute_du_dx else None, du_dp, _i128_to_int(u[0]) if compute_u else None)

    def execute_du_dx(self, coords, params, box):
        """wrap_kernels.cpp:1106-1130."""
        return self.execute(coords, params, box, True, False, False)[0]

    def execute_batch(self, coords, params, boxes, compute_du_dx, compute_du_dp, compute_u):
        """-> (du_dx[C,Pb,N,3], du_dp[C,Pb,*params.shape[1:]], u[C,Pb]); wrap_kernels.cpp:738-862."""
        coords, params, boxes = _f64(coords, "coords"), _f64(params, "params"), _f64(boxes, "boxes")
        if coords.ndim != 3 or boxes.ndim != 3:
            raise RuntimeError("coords and boxes must have 3 dimensions")
        if coords.shape[0] != boxes.shape[0]:
            raise RuntimeError("number of batches of coords and boxes don't match")
        if params.ndim < 2:
            raise RuntimeError("parameters must have at least 2 dimensions")
        C, N = coords.shape[0], coords.shape[1]
        Pb = params.shape[0]
        P = params.size // Pb if Pb else 0
        total = C * Pb
        du_dx = np.full(total * N * 3, 9999, dtype=np.uint64) if compute_du_dx else None
        du_dp = np.full(total * P, 9999, dtype=np.uint64) if compute_du_dp else None
        u = np.zeros(total, dtype=_I128) if compute_u else None
        _check(_lib.tm_potential_execute_batch(
            self._h, _c_int(C), _c_int(N), _c_int(Pb), _c_int(P), _ptr(coords), _ptr(params), _ptr(boxes), _ptr(du_dx), _ptr(du_dp), _ptr(u)))
        r_dx = _fixed_to_float(du_dx).reshape(C, Pb, N, 3) if compute_du_dx else None
        r_dp = None
        if compute_du_dp:
            r_dp = np.empty((C, Pb) + params.shape[1:], dtype=np.float64)
            flat = r_dp.reshape(total, P)
            for i in range(total):
                flat[i] = self._du_dp_to_float(N, P, du_dp[i * P : (i + 1) * P], (P,))
        r_u = np.array([_energy_to_float(x) for x in u], dtype=np.float64).reshape(C, Pb) if compute_u else None
        return r_dx, r_dp, r_u

    def execute_batch_sparse(self, coords, params, boxes, coords_batch_idxs, params_batch_idxs, compute_du_dx, compute_du_dp, compute_u):
        """-> (du_dx[B,N,3], du_dp[B,*params.shape[1:]], u[B]); wrap_kernels.cpp:863-1038."""
        coords, params, boxes = _f64(coords, "coords"), _f64(params, "params"), _f64(boxes, "boxes")
        cidx, pidx = _u32(coords_batch_idxs, "coords_batch_idxs"), _u32(params_batch_idxs, "params_batch_idxs")
        if coords.ndim != 3 or boxes.ndim != 3:
            raise RuntimeError("coords and boxes must have 3 dimensions")
        if coords.shape[0] != boxes.shape[0]:
            raise RuntimeError("number of coord arrays and boxes don't match")
        if params.ndim < 2:
            raise RuntimeError("parameters must have at least 2 dimensions")
        if cidx.ndim != 1 or pidx.ndim != 1:
            raise RuntimeError("coords_batch_idxs and params_batch_idxs must be one-dimensional arrays")
        if cidx.size != pidx.size:
            raise RuntimeError("coords_batch_idxs and params_batch_idxs must have the same length")
        B = cidx.size
        if B and cidx.max() >= coords.shape[0]:
            raise RuntimeError("coords_batch_idxs contains an index that is out of bounds")
        if B and pidx.max() >= params.shape[0]:
            raise RuntimeError("params_batch_idxs contains an index that is out of bounds")
        Cs, N = coords.shape[0], coords.shape[1]
        Ps = params.shape[0]
        P = params.size // Ps if Ps else 0
        du_dx = np.full(B * N * 3, 9999, dtype=np.uint64) if compute_du_dx else None
        du_dp = np.full(B * P, 9999, dtype=np.uint64) if compute_du_dp else None
        u = np.zeros(B, dtype=_I128) if compute_u else None
        _check(_lib.tm_potential_execute_batch_sparse(
            self._h, _c_int(Cs), _c_int(N), _c_int(Ps), _c_int(P), _c_int(B), _ptr(cidx), _ptr(pidx), _ptr(coords), _ptr(params),
            _ptr(boxes), _ptr(du_dx), _ptr(du_dp), _ptr(u)))
        r_dx = _fixed_to_float(du_dx).reshape(B, N, 3) if compute_du_dx else None
        r_dp = None
        if compute_du_dp:
            r_dp = np.empty((B,) + params.shape[1:], dtype=np.float64)
            flat = r_dp.reshape(B, P)
            for i in range(B):
                flat[i] = self._du_dp_to_float(N, P, du_dp[i * P : (i + 1) * P], (P,))
        r_u = np.array([_energy_to_float(x) for x in u], dtype=np.float64) if compute_u else None
        return r_dx, r_dp, r_u


def _new_potential(cls, create_fn, *args, keep=()):
    h = _vp()
    _check(create_fn(*args, ctypes.byref(h)))
    return cls._wrap(h, keep)


def _declare_precision_classes(base_name, ctor):
    """Creates <base_name>_f32 / _f64 (the reference declares each template twice, wrap_kernels.cpp:2186-2216)."""
    out = []
    # the object is made in __new__ (a wrapped C handle); __init__ only carries the constructor's signature, so that
    # inspect.signature(cls.__init__) reads like the reference's stubs (tests/test_api_conformance.py)
    ctor_params = list(inspect.signature(ctor).parameters.values())[2:]  # drop (cls, prec)
    init_signature = inspect.Signature([inspect.Parameter("self", inspect.Parameter.POSITIONAL_OR_KEYWORD)] + ctor_params)
    for suffix, prec in (("f32", _F32), ("f64", _F64)):
        def __new__(cls, *args, _prec=prec, **kwargs):
            return ctor(cls, _prec, *args, **kwargs)

        def __init__(self, *a, **k):
            pass

        __init__.__signature__ = init_signature
        klass = type(f"{base_name}_{suffix}", (Potential,), {"__new__": __new__, "__init__": __init__})
        out.append(klass)
    return out


def _harmonic_bond_ctor(cls, prec, bond_idxs):
    """HarmonicBond_*(bond_idxs int32[B,2]); wrap_kernels.cpp:1311-1322."""
    idx = _i32(bond_idxs, "bond_idxs")
    if idx.size % 2 != 0:
        raise RuntimeError("bond_idxs.size() must be exactly 2*k!")
    return _new_potential(cls, _lib.tm_harmonic_bond_create, _c_int(prec), _ptr(idx), _c_int(idx.size // 2))


def _harmonic_angle_ctor(cls, prec, angle_idxs):
    """HarmonicAngle_*(angle_idxs int32[A,3]); wrap_kernels.cpp:1396-1408."""
    idx = _i32(angle_idxs, "angle_idxs")
    if idx.size % 3 != 0:
        raise RuntimeError("angle_idxs.size() must be exactly 3*A")
    return _new_potential(cls, _lib.tm_harmonic_angle_create, _c_int(prec), _ptr(idx), _c_int(idx.size // 3))


def _periodic_torsion_ctor(cls, prec, angle_idxs):
    """PeriodicTorsion_*(angle_idxs int32[T,4]) -- the kwarg really is ``angle_idxs``; wrap_kernels.cpp:1432-1444."""
    idx = _i32(angle_idxs, "angle_idxs")
    if idx.size % 4 != 0:
        raise RuntimeError("torsion_idxs.size() must be exactly 4*k")
    return _new_potential(cls, _lib.tm_periodic_torsion_create, _c_int(prec), _ptr(idx), _c_int(idx.size // 4))


def _nonbonded_all_pairs_ctor(cls, prec, num_atoms, beta, cutoff, atom_idxs_i=None, disable_hilbert_sort=False, nblist_padding=0.1):
    """NonbondedAllPairs_*(num_atoms, beta, cutoff, atom_idxs_i=None, disable_hilbert_sort=False, nblist_padding=0.1);
    wrap_kernels.cpp:1446-1478."""
    idx = None if atom_idxs_i is None else _i32(atom_idxs_i, "atom_idxs_i")
    return _new_potential(
        cls, _lib.tm_nonbonded_all_pairs_create, _c_int(prec), _c_int(int(num_atoms)), _c_double(beta), _c_double(cutoff), _ptr(idx),
        _c_int(0 if idx is None else idx.size), _c_int(1 if disable_hilbert_sort else 0), _c_double(nblist_padding))


def _pair_list_ctor(negated):
    def ctor(cls, prec, pair_idxs_i, scales_i, beta, cutoff):
        """NonbondedPairList_* / NonbondedExclusions_*(pair_idxs_i int32[M,2], scales_i f64[M,2], beta, cutoff);
        wrap_kernels.cpp:1563-1589."""
        idx = _i32(pair_idxs_i, "pair_idxs_i")
        sc = _f64(scales_i, "scales_i")
        if idx.size % 2 != 0:
            raise RuntimeError(f"pair_idxs.size() must be even, but got {idx.size}")
        return _new_potential(
            cls, _lib.tm_nonbonded_pair_list_create, _c_int(prec), _c_int(negated), _ptr(idx), _c_int(idx.size // 2), _ptr(sc),
            _c_int(sc.size // 2), _c_double(beta), _c_double(cutoff))

    return ctor


def _interaction_group_ctor(
    cls, prec, num_atoms, row_atom_idxs_i, beta, cutoff, col_atom_idxs_i=None, disable_hilbert_sort=False, nblist_padding=0.1
):
    """NonbondedInteractionGroup_*(num_atoms, row_atom_idxs_i, beta, cutoff, col_atom_idxs_i=None,
    disable_hilbert_sort=False, nblist_padding=0.1); wrap_kernels.cpp:1481-1561."""
    rows = _i32(row_atom_idxs_i, "row_atom_idxs_i")
    cols = None if col_atom_idxs_i is None else _i32(col_atom_idxs_i, "col_atom_idxs_i")
    return _new_potential(
        cls, _lib.tm_nonbonded_interaction_group_create, _c_int(prec), _c_int(int(num_atoms)), _ptr(rows), _c_int(rows.size),
        _ptr(cols), _c_int(0 if cols is None else cols.size), _c_double(beta), _c_double(cutoff),
        _c_int(1 if disable_hilbert_sort else 0), _c_double(nblist_padding))


def _pair_list_precomputed_ctor(cls, prec, pair_idxs, beta, cutoff):
    """NonbondedPairListPrecomputed_*(pair_idxs int32[B,2], beta, cutoff); wrap_kernels.cpp:1353-1364."""
    idx = _i32(pair_idxs, "pair_idxs")
    if idx.size % 2 != 0:
        raise RuntimeError("idxs.size() must be exactly 2*B!")
    return _new_potential(
        cls, _lib.tm_nonbonded_pair_list_precomputed_create, _c_int(prec), _ptr(idx), _c_int(idx.size // 2), _c_double(beta), _c_double(cutoff))


def _chiral_atom_ctor(cls, prec, idxs):
    """ChiralAtomRestraint_*(idxs int32[R,4]); wrap_kernels.cpp:1366-1378."""
    idx = _i32(idxs, "idxs")
    if idx.size % 4 != 0:
        raise RuntimeError("idxs.size() must be exactly 4*k!")
    return _new_potential(cls, _lib.tm_chiral_atom_restraint_create, _c_int(prec), _ptr(idx), _c_int(idx.size // 4))


def _chiral_bond_ctor(cls, prec, idxs, signs):
    """ChiralBondRestraint_*(idxs int32[R,4], signs int32[R]); wrap_kernels.cpp:1380-1394."""
    idx = _i32(idxs, "idxs")
    sg = _i32(signs, "signs")
    if idx.size % 4 != 0:
        raise RuntimeError("idxs.size() must be exactly 4*R!")
    return _new_potential(
        cls, _lib.tm_chiral_bond_restraint_create, _c_int(prec), _ptr(idx), _c_int(idx.size // 4), _ptr(sg), _c_int(sg.size))


def _flat_bottom_bond_ctor(cls, prec, bond_idxs):
    """FlatBottomBond_*(bond_idxs int32[B,2]); wrap_kernels.cpp:1324-1336."""
    idx = _i32(bond_idxs, "bond_idxs")
    if idx.size % 2 != 0:
        raise RuntimeError("bond_idxs.size() must be exactly 2*k!")
    return _new_potential(cls, _lib.tm_flat_bottom_bond_create, _c_int(prec), _c_int(0), _ptr(idx), _c_int(idx.size // 2), _c_double(0.0))


def _log_flat_bottom_bond_ctor(cls, prec, bond_idxs, beta):
    """LogFlatBottomBond_*(bond_idxs int32[B,2], beta); wrap_kernels.cpp:1338-1351."""
    idx = _i32(bond_idxs, "bond_idxs")
    if idx.size % 2 != 0:
        raise RuntimeError("bond_idxs.size() must be exactly 2*k!")
    return _new_potential(cls, _lib.tm_flat_bottom_bond_create, _c_int(prec), _c_int(1), _ptr(idx), _c_int(idx.size // 2), _c_double(beta))


def _centroid_restraint_ctor(cls, prec, group_a_idxs, group_b_idxs, kb, b0):
    """CentroidRestraint_*(group_a_idxs, group_b_idxs, kb, b0); wrap_kernels.cpp:1410-1430."""
    a, b = _i32(group_a_idxs, "group_a_idxs"), _i32(group_b_idxs, "group_b_idxs")
    return _new_potential(
        cls, _lib.tm_centroid_restraint_create, _c_int(prec), _ptr(a), _c_int(a.size), _ptr(b), _c_int(b.size), _c_double(kb), _c_double(b0))


HarmonicBond_f32, HarmonicBond_f64 = _declare_precision_classes("HarmonicBond", _harmonic_bond_ctor)
FlatBottomBond_f32, FlatBottomBond_f64 = _declare_precision_classes("FlatBottomBond", _flat_bottom_bond_ctor)
LogFlatBottomBond_f32, LogFlatBottomBond_f64 = _declare_precision_classes("LogFlatBottomBond", _log_flat_bottom_bond_ctor)
CentroidRestraint_f32, CentroidRestraint_f64 = _declare_precision_classes("CentroidRestraint", _centroid_restraint_ctor)
NonbondedInteractionGroup_f32, NonbondedInteractionGroup_f64 = _declare_precision_classes("NonbondedInteractionGroup", _interaction_group_ctor)
NonbondedPairListPrecomputed_f32, NonbondedPairListPrecomputed_f64 = _declare_precision_classes(
    "NonbondedPairListPrecomputed", _pair_list_precomputed_ctor)
ChiralAtomRestraint_f32, ChiralAtomRestraint_f64 = _declare_precision_classes("ChiralAtomRestraint", _chiral_atom_ctor)
ChiralBondRestraint_f32, ChiralBondRestraint_f64 = _declare_precision_classes("ChiralBondRestraint", _chiral_bond_ctor)


def _interaction_group_set_atom_idxs(self, row_atom_idxs, col_atom_idxs):
    """wrap_kernels.cpp:1485-1504"""
    rows = _i32(np.asarray(row_atom_idxs, dtype=np.int32))
    cols = _i32(np.asarray(col_atom_idxs, dtype=np.int32))
    _check(_lib.tm_nonbonded_interaction_group_set_atom_idxs(self._h, _ptr(rows), _c_int(rows.size), _ptr(cols), _c_int(cols.size)))


for _k in (NonbondedInteractionGroup_f32, NonbondedInteractionGroup_f64):
    _k.set_atom_idxs = _interaction_group_set_atom_idxs

HarmonicAngle_f32, HarmonicAngle_f64 = _declare_precision_classes("HarmonicAngle", _harmonic_angle_ctor)
PeriodicTorsion_f32, PeriodicTorsion_f64 = _declare_precision_classes("PeriodicTorsion", _periodic_torsion_ctor)
NonbondedAllPairs_f32, NonbondedAllPairs_f64 = _declare_precision_classes("NonbondedAllPairs", _nonbonded_all_pairs_ctor)
NonbondedPairList_f32, NonbondedPairList_f64 = _declare_precision_classes("NonbondedPairList", _pair_list_ctor(0))
NonbondedExclusions_f32, NonbondedExclusions_f64 = _declare_precision_classes("NonbondedExclusions", _pair_list_ctor(1))


def _all_pairs_set_atom_idxs(self, atom_idxs):
    idx = _i32(np.asarray(atom_idxs, dtype=np.int32))
    _check(_lib.tm_nonbonded_all_pairs_set_atom_idxs(self._h, _ptr(idx), _c_int(idx.size)))


def _all_pairs_get_num_atom_idxs(self):
    n = _c_int(0)
    _check(_lib.tm_nonbonded_all_pairs_get_num_atom_idxs(self._h, ctypes.byref(n)))
    return n.value


def _all_pairs_get_atom_idxs(self):
    n = _all_pairs_get_num_atom_idxs(self)
    out = np.zeros(n, dtype=np.int32)
    _check(_lib.tm_nonbonded_all_pairs_get_atom_idxs(self._h, _ptr(out), _c_int(n)))
    return out.tolist()


def _all_pairs_get_tile_count(self):
    n = ctypes.c_uint(0)
    _check(_lib.tm_nonbonded_all_pairs_get_tile_count(self._h, ctypes.byref(n)))
    return n.value


def _all_pairs_debug_timing(self, max_waves=8192):
    """per-wave cycle counters of the last tile launch (-DTM_TIMING builds): (int64[max_waves, 8], waves)"""
    buf = np.zeros((max_waves, 8), dtype=np.int64)
    cnt = _c_int(0)
    _check(_lib.tm_nonbonded_all_pairs_debug_timing(self._h, _ptr(buf), _c_int(buf.size), ctypes.byref(cnt)))
    return buf, cnt.value


def _all_pairs_get_build_count(self):
    n = ctypes.c_uint(0)
    _check(_lib.tm_nonbonded_all_pairs_get_build_count(self._h, ctypes.byref(n)))
    return n.value


def _all_pairs_get_same_frame_skips(self):
    """diagnostic: evaluations that launched no list kernel on the batch entry point's word (same frame as the call before)"""
    a = ctypes.c_longlong(0)
    _check(_lib.tm_nonbonded_all_pairs_get_same_frame_skips(self._h, ctypes.byref(a)))
    return a.value


def _all_pairs_get_memo_stats(self):
    """diagnostic: (energy-only evaluations remembered on the device, of which the all-pairs launch was empty)"""
    a, b = ctypes.c_longlong(0), ctypes.c_longlong(0)
    _check(_lib.tm_nonbonded_all_pairs_get_memo_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def _all_pairs_get_merged_stats(self):
    """diagnostic: (evaluations made as the carrier of an interaction group, its list's tiles, its list's builds)"""
    calls, tiles, builds = ctypes.c_longlong(0), ctypes.c_uint(0), ctypes.c_uint(0)
    _check(_lib.tm_nonbonded_all_pairs_get_merged_stats(self._h, ctypes.byref(calls), ctypes.byref(tiles), ctypes.byref(builds)))
    return calls.value, tiles.value, builds.value


for _k in (NonbondedAllPairs_f32, NonbondedAllPairs_f64):
    _k.get_merged_stats = _all_pairs_get_merged_stats  # diagnostic (not in the reference surface)
    _k.get_memo_stats = _all_pairs_get_memo_stats  # diagnostic (not in the reference surface)
    _k.get_same_frame_skips = _all_pairs_get_same_frame_skips
    _k.get_build_count = _all_pairs_get_build_count  # diagnostic (not in the reference surface)
    _k.debug_timing = _all_pairs_debug_timing
    _k.set_atom_idxs = _all_pairs_set_atom_idxs
    _k.get_atom_idxs = _all_pairs_get_atom_idxs
    _k.get_num_atom_idxs = _all_pairs_get_num_atom_idxs
    _k.get_tile_ixn_count = _all_pairs_get_tile_count  # diagnostic (not in the reference surface)


def _handles(potentials):
    for p in potentials:
        if not isinstance(p, Potential):
            raise TypeError("potentials must be custom_ops.Potential instances")
    arr = (_vp * len(potentials))(*[p._h.value for p in potentials])
    return arr


class SummedPotential(Potential):
    """SummedPotential(potentials, params_sizes, parallel=True); wrap_kernels.cpp:1661-1676."""

    def __new__(cls, potentials, params_sizes, parallel=True):
        potentials = list(potentials)
        sizes = np.ascontiguousarray(np.asarray(list(params_sizes), dtype=np.int32))
        obj = _new_potential(
            cls, _lib.tm_summed_potential_create, _handles(potentials), _c_int(len(potentials)), _ptr(sizes), _c_int(sizes.size),
            _c_int(1 if parallel else 0), keep=potentials)
        return obj

    def __init__(self, potentials, params_sizes, parallel=True):
        pass

    def get_potentials(self):
        return list(self._keep)


class FanoutSummedPotential(Potential):
    """FanoutSummedPotential(potentials, parallel=True); wrap_kernels.cpp:1678-1691."""

    def __new__(cls, potentials, parallel=True):
        potentials = list(potentials)
        return _new_potential(
            cls, _lib.tm_fanout_summed_potential_create, _handles(potentials), _c_int(len(potentials)), _c_int(1 if parallel else 0),
            keep=potentials)

    def __init__(self, potentials, parallel=True):
        pass

    def get_potentials(self):
        return list(self._keep)


# ---------------------------------------------------------------------------------------------------------------
class BoundPotential:
    """BoundPotential(potential, params); wrap_kernels.cpp:1133-1309."""

    def __init__(self, potential, params):
        if not isinstance(potential, Potential):
            raise TypeError("potential must be a custom_ops.Potential")
        p = _f64(params, "params")
        self._potential = potential  # keeps the Potential alive (tests/test_potentials.py:36-48)
        self._h = _vp()
        _check(_lib.tm_bound_potential_create(potential._h, _ptr(p), _c_int(p.size), ctypes.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and _lib is not None:
            _lib.tm_bound_potential_destroy(h)

    def get_potential(self):
        return self._potential

    def set_params(self, params):
        p = _f64(params, "params")
        _check(_lib.tm_bound_potential_set_params(self._h, _ptr(p), _c_int(p.size)))

    def size(self):
        n = _c_int(0)
        _check(_lib.tm_bound_potential_size(self._h, ctypes.byref(n)))
        return n.value

    def execute(self, coords, box, compute_du_dx=True, compute_u=True):
        """-> (du_dx | None, u | None); wrap_kernels.cpp:1149-1186."""
        coords, box = _f64(coords, "coords"), _f64(box, "box")
        N = coords.shape[0]
        _verify_coords_and_box(coords, box)
        du_dx = np.full(N * 3, 9999, dtype=np.uint64) if compute_du_dx else None
        u = np.zeros(1, dtype=_I128) if compute_u else None
        _check(_lib.tm_bound_potential_execute(self._h, _c_int(N), _ptr(coords), _ptr(box), _ptr(du_dx), _ptr(u)))
        return (_fixed_to_float(du_dx).reshape(N, 3) if compute_du_dx else None, _energy_to_float(u[0]) if compute_u else None)

    def execute_batch(self, coords, boxes, compute_du_dx, compute_u):
        """-> (du_dx[C,N,3] | None, u[C] | None); wrap_kernels.cpp:1187-1274."""
        coords, boxes = _f64(coords, "coords"), _f64(boxes, "boxes")
        if coords.ndim != 3 and boxes.ndim != 3:
            raise RuntimeError("coords and boxes must have 3 dimensions")
        if coords.shape[0] != boxes.shape[0]:
            raise RuntimeError("number of batches of coords and boxes don't match")
        C, N = coords.shape[0], coords.shape[1]
        du_dx = np.full(C * N * 3, 9999, dtype=np.uint64) if compute_du_dx else None
        u = np.zeros(C, dtype=_I128) if compute_u else None
        _check(_lib.tm_bound_potential_execute_batch(self._h, _c_int(C), _c_int(N), _ptr(coords), _ptr(boxes), _ptr(du_dx), _ptr(u)))
        r_dx = _fixed_to_float(du_dx).reshape(C, N, 3) if compute_du_dx else None
        r_u = np.array([_energy_to_float(x) for x in u], dtype=np.float64) if compute_u else None
        return r_dx, r_u

    def execute_fixed(self, coords, box):
        """-> uint64[1]: raw fixed-point energy, LLONG_MAX when overflowed; wrap_kernels.cpp:1275-1308."""
        coords, box = _f64(coords, "coords"), _f64(box, "box")
        N = coords.shape[0]
        _verify_coords_and_box(coords, box)
        u = np.zeros(1, dtype=_I128)
        _check(_lib.tm_bound_potential_execute(self._h, _c_int(N), _ptr(coords), _ptr(box), None, _ptr(u)))
        v = _i128_to_int(u[0])
        if v >= _LLONG_MAX or v <= _LLONG_MIN:
            v = _LLONG_MAX
        return np.array([v & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)


# ---------------------------------------------------------------------------------------------------------------
class Integrator:
    """Base class (wrap_kernels.cpp:691-697)."""

    _h = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and _lib is not None:
            _lib.tm_integrator_destroy(h)


class LangevinIntegrator(Integrator):
    """LangevinIntegrator(masses f64[N], temperature, dt, friction, seed); wrap_kernels.cpp:699-715 (<float> arithmetic)."""

    def __init__(self, masses, temperature, dt, friction, seed):
        m = _f64(masses, "masses")
        self._h = _vp()
        _check(_lib.tm_langevin_integrator_create(
            _ptr(m), _c_int(m.size), _c_double(temperature), _c_double(dt), _c_double(friction), _c_int(int(seed)), ctypes.byref(self._h)))


class VelocityVerletIntegrator(Integrator):
    """VelocityVerletIntegrator(dt, cbs f64[N]) with cbs = -dt / mass; wrap_kernels.cpp:717-729 (double arithmetic)."""

    def __init__(self, dt, cbs):
        c = _f64(cbs, "cbs")
        self._h = _vp()
        _check(_lib.tm_velocity_verlet_integrator_create(_c_double(dt), _ptr(c), _c_int(c.size), ctypes.byref(self._h)))


class Mover:
    """Mover base (wrap_kernels.cpp:1591-1617): set_interval / get_interval / set_step / move(coords, box)."""

    _h = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and _lib is not None:
            _lib.tm_mover_destroy(h)

    def set_interval(self, interval):
        _check(_lib.tm_mover_set_interval(self._h, _c_int(int(interval))))

    def get_interval(self):
        n = _c_int(0)
        _check(_lib.tm_mover_get_interval(self._h, ctypes.byref(n)))
        return n.value

    def set_step(self, step):
        _check(_lib.tm_mover_set_step(self._h, _c_int(int(step))))

    def move(self, coords, box):
        """-> (coords[N,3], box[3,3]) after one call of the mover (it acts on every interval-th call)."""
        c, b = _f64(coords, "coords"), _f64(box, "box")
        _verify_coords_and_box(c, b)
        x_out, box_out = np.empty_like(c), np.empty_like(b)
        _check(_lib.tm_mover_move(self._h, _c_int(c.shape[0]), _ptr(c), _ptr(b), _ptr(x_out), _ptr(box_out)))
        return x_out, box_out


class MonteCarloBarostat(Mover):
    """MonteCarloBarostat(N, pressure [bar], temperature [K], group_idxs, interval, bps, seed, adaptive_scaling_enabled,
    initial_volume_scale_factor); wrap_kernels.cpp:1619-1659 (<float> arithmetic)."""

    def __init__(self, N, pressure, temperature, group_idxs, interval, bps, seed, adaptive_scaling_enabled, initial_volume_scale_factor):
        groups = [np.asarray(g, dtype=np.int32).reshape(-1) for g in group_idxs]
        flat = np.ascontiguousarray(np.concatenate(groups) if groups else np.zeros(0, np.int32), dtype=np.int32)
        offsets = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(g) for g in groups])]), dtype=np.int32)
        bps = list(bps)
        for bp in bps:
            if not isinstance(bp, BoundPotential):
                raise TypeError("bps must be custom_ops.BoundPotential instances")
        self._bps = bps  # the barostat evaluates them: keep them alive
        arr = (_vp * len(bps))(*[bp._h.value for bp in bps])
        self._h = _vp()
        _check(_lib.tm_monte_carlo_barostat_create(
            _c_int(int(N)), _c_double(pressure), _c_double(temperature), _ptr(flat), _ptr(offsets), _c_int(len(groups)),
            _c_int(int(interval)), arr, _c_int(len(bps)), _c_int(int(seed)), _c_int(1 if adaptive_scaling_enabled else 0),
            _c_double(initial_volume_scale_factor), ctypes.byref(self._h)))

    def set_volume_scale_factor(self, volume_scale_factor):
        _check(_lib.tm_barostat_set_volume_scale_factor(self._h, _c_double(volume_scale_factor)))

    def get_volume_scale_factor(self):
        f = _c_double(0)
        _check(_lib.tm_barostat_get_volume_scale_factor(self._h, ctypes.byref(f)))
        return f.value

    def set_adaptive_scaling(self, adaptive_scaling_enabled):
        _check(_lib.tm_barostat_set_adaptive_scaling(self._h, _c_int(1 if adaptive_scaling_enabled else 0)))

    def get_adaptive_scaling(self):
        n = _c_int(0)
        _check(_lib.tm_barostat_get_adaptive_scaling(self._h, ctypes.byref(n)))
        return bool(n.value)

    def set_pressure(self, pressure):
        _check(_lib.tm_barostat_set_pressure(self._h, _c_double(pressure)))

    def get_counters(self):
        """(accepted, attempted) since the last adaptive reset -- diagnostic, not in the reference surface"""
        a, b = _c_int(0), _c_int(0)
        _check(_lib.tm_barostat_get_counters(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def get_attempt_paths(self):
        """(attempts since construction, of which on the nonbonded potential's current list) -- diagnostic"""
        a, b = ctypes.c_longlong(0), ctypes.c_longlong(0)
        _check(_lib.tm_barostat_get_attempt_paths(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value


class Context:
    """Context(x0, v0, box, integrator, bps, movers=None); wrap_kernels.cpp:296-689."""

    def __init__(self, x0, v0, box, integrator, bps, movers=None):
        x0, v0, box = _f64(x0, "x0"), _f64(v0, "v0"), _f64(box, "box")
        _verify_coords_and_box(x0, box)
        if x0.shape[0] != v0.shape[0]:
            raise RuntimeError("v0 N != x0 N")
        if v0.ndim != 2 or x0.shape[1] != v0.shape[1]:
            raise RuntimeError("v0 D != x0 D")
        movers = list(movers) if movers else []
        for mv in movers:
            if not isinstance(mv, Mover):
                raise TypeError("movers must be custom_ops.Mover instances")
        if not isinstance(integrator, Integrator):
            raise TypeError("integrator must be a custom_ops.Integrator")
        bps = list(bps)
        for bp in bps:
            if not isinstance(bp, BoundPotential):
                raise TypeError("bps must be custom_ops.BoundPotential instances")
        self._integrator, self._bps, self._movers = integrator, bps, movers
        self._N = x0.shape[0]
        arr = (_vp * len(bps))(*[bp._h.value for bp in bps])
        marr = (_vp * max(len(movers), 1))(*[mv._h.value for mv in movers])
        self._h = _vp()
        _check(_lib.tm_context_create_with_movers(
            _ptr(x0), _ptr(v0), _ptr(box), _c_int(self._N), integrator._h, arr, _c_int(len(bps)), marr, _c_int(len(movers)),
            ctypes.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and _lib is not None:
            _lib.tm_context_destroy(h)

    def step(self):
        _check(_lib.tm_context_step(self._h))

    def initialize(self):
        _check(_lib.tm_context_initialize(self._h))

    def finalize(self):
        _check(_lib.tm_context_finalize(self._h))

    def last_multiple_steps_ms(self):
        """measurement aid (not in the reference surface): device time of the steps of the last multiple_steps call"""
        ms = _c_double(0)
        _check(_lib.tm_context_last_multiple_steps_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def multiple_steps(self, n_steps, store_x_interval=0):
        """-> (xs[F,N,3], boxes[F,3,3]), F = n_steps // (store_x_interval or n_steps); wrap_kernels.cpp:347-369."""
        if store_x_interval < 0:
            raise RuntimeError("store_x_interval must be greater than or equal to zero")
        n_steps = int(n_steps)
        x_interval = n_steps if store_x_interval == 0 else int(store_x_interval)
        n_samples = n_steps // x_interval if x_interval > 0 else 0
        xs = np.empty((n_samples, self._N, 3), dtype=np.float64)
        boxes = np.empty((n_samples, 3, 3), dtype=np.float64)
        _check(_lib.tm_context_multiple_steps(self._h, _c_int(n_steps), _c_int(n_samples), _ptr(xs), _ptr(boxes)))
        return xs, boxes

    # ---- local MD (wrap_kernels.cpp:399-631; context.cu:90-213) ----
    def setup_local_md(self, temperature, freeze_reference):
        """Builds the potentials local MD adds (restraints, free x frozen interaction group).  Done implicitly, with the
        Langevin integrator's temperature and a frozen reference, by the first local-MD call; idempotent for equal
        arguments, RuntimeError for different ones."""
        _check(_lib.tm_context_setup_local_md(self._h, _c_double(float(temperature)), _c_int(1 if freeze_reference else 0)))

    @staticmethod
    def _local_num_samples(n_steps, store_x_interval):
        # sizes only; the C-ABI call validates (and raises the binding's messages) before anything is written
        if n_steps <= 0 or store_x_interval < 0:
            return 0
        return n_steps // (n_steps if store_x_interval == 0 else store_x_interval)

    def multiple_steps_local(self, n_steps, local_idxs, store_x_interval=0, radius=1.2, k=10000.0, seed=2022):
        """Steps in which only atoms selected around a random member of ``local_idxs`` move (probability
        exp(-U_flat_bottom(r) / kT)); the chosen atom itself is frozen unless setup_local_md said otherwise.  Movers do not
        run.  -> (xs[F,N,3], boxes[F,3,3])."""
        n_steps, store_x_interval, seed = int(n_steps), int(store_x_interval), int(seed)
        idxs = _i32(local_idxs, "local_idxs").reshape(-1)
        n_samples = self._local_num_samples(n_steps, store_x_interval)
        xs = np.empty((n_samples, self._N, 3), dtype=np.float64)
        boxes = np.empty((n_samples, 3, 3), dtype=np.float64)
        _check(_lib.tm_context_multiple_steps_local(
            self._h, _c_int(n_steps), _ptr(idxs), _c_int(idxs.size), _c_int(store_x_interval), _c_double(float(radius)),
            _c_double(float(k)), _c_int(seed), _ptr(xs), _ptr(boxes)))
        return xs, boxes

    def multiple_steps_local_selection(self, n_steps, reference_idx, selection_idxs, store_x_interval=0, radius=1.2, k=10000.0):
        """Local MD with the free atoms chosen by the caller (restrained to ``reference_idx``, which stays frozen unless
        setup_local_md said otherwise).  -> (xs[F,N,3], boxes[F,3,3])."""
        n_steps, store_x_interval = int(n_steps), int(store_x_interval)
        idxs = _i32(selection_idxs, "selection_idxs").reshape(-1)
        n_samples = self._local_num_samples(n_steps, store_x_interval)
        xs = np.empty((n_samples, self._N, 3), dtype=np.float64)
        boxes = np.empty((n_samples, 3, 3), dtype=np.float64)
        _check(_lib.tm_context_multiple_steps_local_selection(
            self._h, _c_int(n_steps), _c_int(int(reference_idx)), _ptr(idxs), _c_int(idxs.size), _c_int(store_x_interval),
            _c_double(float(radius)), _c_double(float(k)), _ptr(xs), _ptr(boxes)))
        return xs, boxes

    def local_md_last_selection(self):
        """diagnostic (not in the reference surface): (reference atom, indices of the atoms that moved) of the last
        local-MD call; (-1, empty) before the first."""
        ref = _c_int(-1)
        free = np.full(self._N, self._N, dtype=np.uint32)
        _check(_lib.tm_context_local_md_last_selection(self._h, ctypes.byref(ref), _ptr(free)))
        return ref.value, np.flatnonzero(free < self._N).astype(np.int32) if ref.value >= 0 else np.zeros(0, np.int32)

    def set_x_t(self, coords):
        c = _f64(coords, "coords")
        if c.shape[0] != self._N:
            raise RuntimeError("number of new coords disagree with current coords")
        _check(_lib.tm_context_set_x_t(self._h, _ptr(c)))

    def set_v_t(self, velocities):
        v = _f64(velocities, "velocities")
        if v.shape[0] != self._N:
            raise RuntimeError("number of new velocities disagree with current coords")
        _check(_lib.tm_context_set_v_t(self._h, _ptr(v)))

    def set_box(self, box):
        b = _f64(box, "box")
        if b.size != 9 or b.shape[0] != 3:
            raise RuntimeError("box must be 3x3")
        _check(_lib.tm_context_set_box(self._h, _ptr(b)))

    def get_x_t(self):
        out = np.empty((self._N, 3), dtype=np.float64)
        _check(_lib.tm_context_get_x_t(self._h, _ptr(out)))
        return out

    def get_v_t(self):
        out = np.empty((self._N, 3), dtype=np.float64)
        _check(_lib.tm_context_get_v_t(self._h, _ptr(out)))
        return out

    def get_box(self):
        out = np.empty((3, 3), dtype=np.float64)
        _check(_lib.tm_context_get_box(self._h, _ptr(out)))
        return out

    def get_integrator(self):
        return self._integrator

    def get_potentials(self):
        return list(self._bps)

    def get_movers(self):
        return list(self._movers)

    def get_barostat(self):
        """The first MonteCarloBarostat among the movers, else None (wrap_kernels.cpp:671-680; context.cu:311-319)."""
        for mv in self._movers:
            if isinstance(mv, MonteCarloBarostat):
                return mv
        return None


# ---------------------------------------------------------------------------------------------------------------
class _Neighborlist:
    """Neighborlist_f32/_f64(N); wrap_kernels.cpp:113-172."""

    _prec = None

    def __init__(self, N):
        self._h = _vp()
        _check(_lib.tm_neighborlist_create(_c_int(self._prec), _c_int(int(N)), ctypes.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and _lib is not None:
            _lib.tm_neighborlist_destroy(h)

    def compute_block_bounds(self, coords, box, block_size):
        if block_size != 32:
            raise RuntimeError("Block size must be 32.")
        coords, box = _f64(coords, "coords"), _f64(box, "box")
        _verify_coords_and_box(coords, box)
        N = coords.shape[0]
        B = (N + block_size - 1) // block_size
        ctrs, exts = np.empty((B, 3)), np.empty((B, 3))
        _check(_lib.tm_neighborlist_compute_block_bounds(self._h, _c_int(N), _ptr(coords), _ptr(box), _c_int(block_size), _ptr(ctrs), _ptr(exts)))
        return ctrs, exts

    def get_nblist(self, coords, box, cutoff):
        coords, box = _f64(coords, "coords"), _f64(box, "box")
        _verify_coords_and_box(coords, box)
        nrb, total = _c_int(0), _c_int(0)
        _check(_lib.tm_neighborlist_get_nblist(self._h, _c_int(coords.shape[0]), _ptr(coords), _ptr(box), _c_double(cutoff), ctypes.byref(nrb), ctypes.byref(total)))
        offsets = np.zeros(nrb.value + 1, dtype=np.int32)
        atoms = np.zeros(max(total.value, 1), dtype=np.int32)
        _check(_lib.tm_neighborlist_copy_nblist(self._h, _ptr(offsets), _ptr(atoms)))
        return [atoms[offsets[r] : offsets[r + 1]].tolist() for r in range(nrb.value)]

    def set_row_idxs(self, idxs):
        i = _u32(idxs, "idxs")
        _check(_lib.tm_neighborlist_set_row_idxs(self._h, _ptr(i), _c_int(i.size)))

    def reset_row_idxs(self):
        _check(_lib.tm_neighborlist_reset_row_idxs(self._h))

    def resize(self, size):
        _check(_lib.tm_neighborlist_resize(self._h, _c_int(int(size))))

    def get_tile_ixn_count(self):
        n = ctypes.c_uint(0)
        _check(_lib.tm_neighborlist_get_tile_ixn_count(self._h, ctypes.byref(n)))
        return n.value

    def get_max_ixn_count(self):
        n = _c_int(0)
        _check(_lib.tm_neighborlist_get_max_ixn_count(self._h, ctypes.byref(n)))
        return n.value

    def get_num_row_idxs(self):
        n = _c_int(0)
        _check(_lib.tm_neighborlist_get_num_row_idxs(self._h, ctypes.byref(n)))
        return n.value


class Neighborlist_f32(_Neighborlist):
    _prec = _F32


class Neighborlist_f64(_Neighborlist):
    _prec = _F64


class HilbertSort:
    """HilbertSort(size).sort(coords, box) -> uint32[N]; wrap_kernels.cpp:174-194."""

    def __init__(self, size):
        self._h = _vp()
        _check(_lib.tm_hilbert_sort_create(_c_int(int(size)), ctypes.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and _lib is not None:
            _lib.tm_hilbert_sort_destroy(h)

    def sort(self, coords, box):
        coords, box = _f64(coords, "coords"), _f64(box, "box")
        _verify_coords_and_box(coords, box)
        perm = np.zeros(coords.shape[0], dtype=np.uint32)
        _check(_lib.tm_hilbert_sort_sort(self._h, _c_int(coords.shape[0]), _ptr(coords), _ptr(box), _ptr(perm)))
        return perm


def hilbert_lut():
    """Host-only: the 128^3 bin -> Hilbert index table (cpp/src/hilbert_sort.cu:18-31)."""
    out = np.zeros(128 * 128 * 128, dtype=np.uint32)
    _check(_lib.tm_hilbert_lut(_ptr(out)))
    return out


def es_force_table(beta):
    """Host-only: [256, 6] polynomial coefficients of the f64 kernels' electrostatic force factor F(d^2) (nb_es_table.hip.hpp)."""
    out = np.zeros((256, 6), dtype=np.float64)
    _check(_lib.tm_es_force_table(_c_double(float(beta)), _ptr(out)))
    return out


def es_energy_table(beta):
    """Host-only: [256, 6] polynomial coefficients of the f64 kernels' electrostatic energy factor G(d^2) (nb_es_table.hip.hpp)."""
    out = np.zeros((256, 6), dtype=np.float64)
    _check(_lib.tm_es_energy_table(_c_double(float(beta)), _ptr(out)))
    return out


def hrex_run_neighbor_swaps(replica_idx_by_state, neighbor_pairs, log_q_kl, pair_idxs, uniform_samples):
    """The swap chain of one HREX exchange step, in native code (timemachine/md/hrex.py:50-130 is a jitted lax.scan; a
    Python loop over n_states**3 attempts costs 25 ms at 24 states).  -> (replica_idx_by_state, proposed, accepted)"""
    perm = np.ascontiguousarray(replica_idx_by_state, dtype=np.int64)
    pairs = np.ascontiguousarray(neighbor_pairs, dtype=np.int64).reshape(-1, 2)
    log_q = np.ascontiguousarray(log_q_kl, dtype=np.float64)
    idx = np.ascontiguousarray(pair_idxs, dtype=np.int64).reshape(-1)
    uni = np.ascontiguousarray(uniform_samples, dtype=np.float64).reshape(-1)
    if log_q.ndim != 2 or log_q.shape[1] != perm.size or idx.size != uni.size:
        raise RuntimeError("run_neighbor_swaps: inconsistent shapes")
    out = np.zeros(perm.size, dtype=np.int64)
    proposed, accepted = np.zeros(len(pairs), dtype=np.uint32), np.zeros(len(pairs), dtype=np.uint32)
    _check(_lib.tm_hrex_run_neighbor_swaps(
        _c_int(log_q.shape[0]), _c_int(perm.size), _ptr(perm), _c_int(len(pairs)), _ptr(pairs), _ptr(log_q), _c_int(idx.size), _ptr(idx),
        _ptr(uni), _ptr(out), _ptr(proposed), _ptr(accepted)))
    return out, proposed, accepted


def debug_float_to_fixed(values, precision, kind=0):
    """The device's fixed-point conversion of `values` (see include/timemachine_amd.h: tm_debug_float_to_fixed).
    kind 3 takes an [n, 2] array of (prefactor, delta) pairs.  -> uint64[n]"""
    v = np.ascontiguousarray(values, dtype=np.float64)
    n = v.shape[0] if kind == 3 else v.size
    out = np.zeros(n, dtype=np.uint64)
    _check(_lib.tm_debug_float_to_fixed(_c_int(_F64 if np.dtype(precision) == np.float64 else _F32), _c_int(int(kind)), _ptr(v), _c_int(n), _ptr(out)))
    return out


def debug_float_to_fixed_energy(values, precision):
    """FLOAT_TO_FIXED_ENERGY on the device -> list of python ints (signed 128-bit values)."""
    v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    out = np.zeros(v.size, dtype=[("lo", np.uint64), ("hi", np.int64)])
    _check(_lib.tm_debug_float_to_fixed_energy(_c_int(_F64 if np.dtype(precision) == np.float64 else _F32), _ptr(v), _c_int(v.size), _ptr(out)))
    return [(int(r["hi"]) << 64) | int(r["lo"]) for r in out]


def debug_set_box_scaling_reuse(enabled):
    """A/B aid: 0 = every box change rebuilds the neighbor lists (as the reference does); results are bit-identical."""
    _check(_lib.tm_debug_set_box_scaling_reuse(_c_int(1 if enabled else 0)))


def debug_check_guards():
    """-DTM_GUARD builds: violated guard zones so far; -1 in product builds"""
    n = _c_int(0)
    _check(_lib.tm_debug_check_guards(ctypes.byref(n)))
    return n.value


def version():
    return _lib.tm_version().decode()


def debug_set_static_list_max_k(max_atoms):
    """A/B aid: potentials over at most `max_atoms` atoms keep a static, complete interaction list (0: off); -> the old value.
    Results are bit-identical either way."""
    prev = _c_int(0)
    _check(_lib.tm_debug_set_static_list_max_k(_c_int(int(max_atoms)), ctypes.byref(prev)))
    return prev.value


def multiple_steps_group(contexts, n_steps):
    """n_steps of several distinct Contexts, interleaved step by step on their own streams (not in the reference surface: windows or
    HREX replicas that share a GPU; one context's list / update kernels run underneath another's force kernel)."""
    contexts = list(contexts)
    arr = (_vp * max(len(contexts), 1))(*[c._h.value for c in contexts])
    _check(_lib.tm_context_multiple_steps_group(arr, _c_int(len(contexts)), _c_int(int(n_steps))))


def debug_set_rowblock_min_k(min_atoms):
    """A/B aid (variant library libtimemachine_amd_rowblock.so only; the product library refuses anything but INT_MAX): forces-only
    nonbonded launches over at least `min_atoms` atoms run the row-block kernel (0: always); -> the old value.  Results are
    bit-identical either way."""
    prev = _c_int(0)
    _check(_lib.tm_debug_set_rowblock_min_k(_c_int(int(min_atoms)), ctypes.byref(prev)))
    return prev.value


def debug_last_host_call_device_ms():
    """device time of the evaluations of the last tm_potential_execute*_f64 call of this process (diagnostic; the ctypes twin's own
    execute methods call the u64 forms, which do not set it)"""
    ms = ctypes.c_double(0.0)
    _check(_lib.tm_debug_last_host_call_device_ms(ctypes.byref(ms)))
    return ms.value


def debug_set_same_frame_hint(enabled):
    """A/B aid: the batch entry points' same-frame hint honoured (True) or ignored (False); -> the old value.  Bit-identical either way."""
    prev = _c_int(0)
    _check(_lib.tm_debug_set_same_frame_hint(_c_int(1 if enabled else 0), ctypes.byref(prev)))
    return bool(prev.value)


def debug_set_energy_memo(enabled):
    """A/B aid: energy-only evaluations remembered on the device (True) or always recomputed (False); -> the old value.
    Bit-identical either way."""
    prev = _c_int(0)
    _check(_lib.tm_debug_set_energy_memo(_c_int(1 if enabled else 0), ctypes.byref(prev)))
    return bool(prev.value)


def debug_set_merge_producers(enabled):
    """A/B aid: an all-pairs potential and an interaction group on exactly its atoms run as one pipeline (True) or each for itself
    (False); -> the old value.  Bit-identical either way."""
    prev = _c_int(0)
    _check(_lib.tm_debug_set_merge_producers(_c_int(1 if enabled else 0), ctypes.byref(prev)))
    return bool(prev.value)


def debug_set_barostat_fast_path(enabled):
    """A/B aid: MonteCarloBarostat attempts on the nonbonded potential's current list (True) or reference-shaped (False); -> the old
    value.  Bit-identical either way."""
    prev = _c_int(0)
    _check(_lib.tm_debug_set_barostat_fast_path(_c_int(1 if enabled else 0), ctypes.byref(prev)))
    return bool(prev.value)


def debug_rowblock_available():
    """does the loaded library carry the row-block kernel (libtimemachine_amd_rowblock.so, the parity tests' variant library)?"""
    yes = _c_int(0)
    _check(_lib.tm_debug_rowblock_available(ctypes.byref(yes)))
    return bool(yes.value)


def profile_set_enabled(enabled):
    _check(_lib.tm_profile_set_enabled(_c_int(1 if enabled else 0)))


def profile_read(name="nonbonded_tiles"):
    ms, n = _c_double(0), ctypes.c_longlong(0)
    _check(_lib.tm_profile_read(name.encode(), ctypes.byref(ms), ctypes.byref(n)))
    return ms.value, n.value


def profile_reset():
    _check(_lib.tm_profile_reset())


# Entries of the reference module that are outside the MI355X hot path: fail loudly, by name.
def _not_on_hot_path(name):
    class _Missing:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"custom_ops.{name} is outside the MI355X hot path of timemachine_amd (see DESIGN.md, 'out of scope')")

    _Missing.__name__ = name
    return _Missing


for _name in (
    # exchange movers and their device helpers (cpp/src/bd_exchange_move.cu, tibd_exchange_move.cu, segmented_*.cu)
    "BDExchangeMove_f32", "BDExchangeMove_f64", "TIBDExchangeMove_f32", "TIBDExchangeMove_f64",
    "NonbondedMolEnergyPotential_f32", "NonbondedMolEnergyPotential_f64", "SegmentedSumExp_f32", "SegmentedSumExp_f64",
    "SegmentedWeightedRandomSampler_f32", "SegmentedWeightedRandomSampler_f64",
    # module-level helpers of the exchange movers / local MD (wrap_kernels.cpp:2228-2309)
    "atom_by_atom_energies_f32", "atom_by_atom_energies_f64", "inner_and_outer_mols_f32", "inner_and_outer_mols_f64", "rmsd_align",
    "rotate_and_translate_mol_f32", "rotate_and_translate_mol_f64", "rotate_coords_f32", "rotate_coords_f64",
    "translations_inside_and_outside_sphere_host_f32", "translations_inside_and_outside_sphere_host_f64",
):
    globals()[_name] = _not_on_hot_path(_name)
