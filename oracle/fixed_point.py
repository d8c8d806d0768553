"""Integer model of the reference's fixed-point accumulation.  TEST INFRASTRUCTURE ONLY.  Bit-exact.

cpp/src/fixed_point.hpp:5-34 (exponents, FIXED_TO_FLOAT*, fixed_point_overflow),
cpp/src/kernels/k_fixed_point.cuh:10-98 (real_to_int64, FLOAT_TO_FIXED*, FLOAT_TO_FIXED_ENERGY),
cpp/src/wrap_kernels.cpp:83-89 (convert_energy_to_fp), timemachine/lib/fixed_point.py:8-15.
"""
import numpy as np

FIXED_EXPONENT = 0x1000000000  # 2**36, fixed_point.hpp:5
FIXED_EXPONENT_DU_DCHARGE = 0x1000000000  # fixed_point.hpp:8
FIXED_EXPONENT_DU_DSIG = 0x2000000000  # fixed_point.hpp:9
FIXED_EXPONENT_DU_DEPS = 0x4000000000  # fixed_point.hpp:10
FIXED_EXPONENT_DU_DW = 0x1000000000  # fixed_point.hpp:11
NB_DU_DP_EXPONENTS = (FIXED_EXPONENT_DU_DCHARGE, FIXED_EXPONENT_DU_DSIG, FIXED_EXPONENT_DU_DEPS, FIXED_EXPONENT_DU_DW)

LLONG_MAX = np.iinfo(np.int64).max
LLONG_MIN = np.iinfo(np.int64).min


def float_to_fixed(v, exponent=FIXED_EXPONENT, real=np.float64):
    """(u64)(i64) llrint(v * exponent) with the product formed in ``real`` precision
    (k_fixed_point.cuh:56-71; the f32 bit trick at :10-24 equals round-to-nearest-even)."""
    v = np.asarray(v, dtype=real)
    prod = (v * real(exponent)).astype(np.float64)  # exact widening
    return np.rint(prod).astype(np.int64).view(np.uint64)


def fixed_to_float(v, exponent=FIXED_EXPONENT):
    """(double)(i64)v / exponent, fixed_point.hpp:13-21."""
    return np.asarray(v, dtype=np.uint64).view(np.int64).astype(np.float64) / float(exponent)


def nb_du_dp_fixed_to_float(du_dp_fixed):
    """Per-column exponents of the (N,4) nonbonded du_dp, cpp/src/nonbonded_all_pairs.cu:292-308."""
    a = np.asarray(du_dp_fixed, dtype=np.uint64).reshape(-1, 4)
    out = np.empty(a.shape, dtype=np.float64)
    for c, e in enumerate(NB_DU_DP_EXPONENTS):
        out[:, c] = fixed_to_float(a[:, c], e)
    return out


def float_to_fixed_energy(u, real=np.float64):
    """k_fixed_point.cuh:88-98 -> python int (128-bit capable).  Non-finite or |u*2^36| >= 2^63 -> LLONG_MAX."""
    x = float(real(u) * real(FIXED_EXPONENT))
    if not np.isfinite(x) or int(x) >= int(LLONG_MAX) or int(x) <= int(LLONG_MIN):
        return int(LLONG_MAX)
    return int(np.rint(x))


def fixed_point_overflow(v: int) -> bool:
    """fixed_point.hpp:30-34."""
    return v >= int(LLONG_MAX) or v <= int(LLONG_MIN)


def energy_to_float(v: int) -> float:
    """wrap_kernels.cpp:83-89: NaN on overflow."""
    if fixed_point_overflow(v):
        return float("nan")
    return float(v) / float(FIXED_EXPONENT)


def wrapping_sum(values_u64, axis=None):
    """Two's-complement wrapping sum of u64 contributions (the accumulator semantics of every atomicAdd)."""
    with np.errstate(over="ignore"):
        return np.asarray(values_u64, dtype=np.uint64).sum(axis=axis, dtype=np.uint64)
