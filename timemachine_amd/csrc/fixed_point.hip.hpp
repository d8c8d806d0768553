// Device-side fixed-point contract (integer work: must be bit-exact with the reference).
//   reference: cpp/src/fixed_point.hpp:5-34, cpp/src/kernels/k_fixed_point.cuh:10-98
// Forces / du_dp accumulate as wrapping u64 two's-complement with scale 2^36 (du/dsig 2^37, du/deps 2^38);
// energies as signed 128-bit with "beyond int64 => invalid" semantics.
#pragma once
#include "common.hpp"

namespace tmamd {

#define TM_FIXED_EXPONENT 0x1000000000ULL
#define TM_FIXED_EXPONENT_DU_DCHARGE 0x1000000000ULL
#define TM_FIXED_EXPONENT_DU_DSIG 0x2000000000ULL
#define TM_FIXED_EXPONENT_DU_DEPS 0x4000000000ULL
#define TM_FIXED_EXPONENT_DU_DW 0x1000000000ULL

// round-to-nearest-even float -> int64.  The reference's f32 bit trick (k_fixed_point.cuh:10-24) is an
// exact llrintf for every finite input in range, so llrintf/llrint are the contract.
//
// f64 fast path: for |x| < 2^51 adding 1.5*2^52 leaves round-half-even(x) in the low mantissa bits, so
// the integer is one add + one 32-bit integer subtract instead of the ~8-instruction cvt sequence.  The slow
// conversion (|x| >= 2^51, i.e. forces beyond 2^15 kJ/mol/nm, or NaN) sits behind a WAVE-UNIFORM branch: a per-lane
// branch would be if-converted and its instructions issued (masked off) for every pair.
#define TM_FIXED_MAGIC 6755399441055744.0     // 1.5 * 2^52
#define TM_FIXED_FAST_LIMIT 2251799813685248.0 // 2^51
__device__ __forceinline__ long long real_to_int64_fast(double x) {
    return __double_as_longlong(x + TM_FIXED_MAGIC) - __double_as_longlong(TM_FIXED_MAGIC);
}
// The slow conversion of a value whose NEGATION is applied to the pair's other atom as the two's-complement negation of the result
// (pair_force_fixed*: FIX(-v) == -FIX(v), three conversions per pair instead of six).  llrint is odd for every value an int64
// holds; BEYOND that range (a force component above 2^27 kJ/mol/nm: the accumulator's own range, here as in the reference) the
// device's conversion saturates its high word to 2^31 - 1 on one side and -2^31 on the other, so FIX(-v) and -FIX(v) differed by
// 2^32 units -- 2^-4 kJ/mol/nm -- and a pair's contribution depended on which of its atoms the tile had as its row: the list's
// state showed in the result (found by scripts/fuzz_campaign.py on a synthetic ligand whose excluded 1-3 pairs clash that hard).
// Odd by construction: the same integers for every in-range value, and whatever the saturation gives, mirrored.
__device__ __forceinline__ long long tm_llrint_odd(const double x) {
    const long long r = llrint(__builtin_fabs(x));
    return x < 0 ? -r : r;
}
__device__ __forceinline__ long long real_to_int64(double x) {
    long long r = real_to_int64_fast(x);
    const bool big = !(__builtin_fabs(x) < TM_FIXED_FAST_LIMIT);
    if (__ballot(big) != 0ull) {
        if (big) {
            r = llrint(x);
        }
    }
    return r;
}
__device__ __forceinline__ long long real_to_int64(float x) {
    // widening is exact; reuse the f64 path (|x| < 2^51 almost always)
    return real_to_int64(static_cast<double>(x));
}

template <typename Real, u64 EXPONENT> __device__ __forceinline__ u64 float_to_fixed_exp(Real v) {
    return static_cast<u64>(real_to_int64(v * static_cast<Real>(EXPONENT)));
}
template <typename Real> __device__ __forceinline__ u64 float_to_fixed(Real v) {
    return float_to_fixed_exp<Real, TM_FIXED_EXPONENT>(v);
}

template <typename Real> __host__ __device__ __forceinline__ Real fixed_to_float(u64 v) {
    return static_cast<Real>(static_cast<long long>(v)) / static_cast<Real>(TM_FIXED_EXPONENT);
}

// k_fixed_point.cuh:88-98: anything non-finite or outside (LLONG_MIN, LLONG_MAX) is pinned to LLONG_MAX so
// that clashes cannot cancel with a different-sign clash; only exclusion subtraction cancels them.
template <typename Real> __device__ __forceinline__ i128 float_to_fixed_energy(Real u_orig) {
    Real u = u_orig * static_cast<Real>(TM_FIXED_EXPONENT);
    const Real lim = static_cast<Real>(9223372036854775808.0); // 2^63
    if (!isfinite(u) || u >= lim || u <= -lim) {
        return static_cast<i128>(LLONG_MAX);
    }
    long long r = real_to_int64(u);
    // (i128)u >= LLONG_MAX in the reference truncates toward zero first: values that round up to exactly
    // 2^63-ish are already excluded by the |u| >= 2^63 test above; LLONG_MAX itself is not representable in Real.
    return static_cast<i128>(r);
}

// The same value for a caller inside a hot loop (the tile kernel's energy launches): the common case -- |u 2^36| < 2^51, i.e. a pair
// energy below 32 768 kJ/mol -- is one multiply, the magic add and an integer subtract; everything else (large, infinite, NaN)
// takes float_to_fixed_energy() behind ONE wave-uniform branch (left to itself the compiler if-converts the general function:
// both conversions and the range logic, ~20 instructions, issue for every pair).  Same integers.
template <typename Real> __device__ __forceinline__ i128 float_to_fixed_energy_hot(Real u_orig) {
    const double u = static_cast<double>(u_orig * static_cast<Real>(TM_FIXED_EXPONENT)); // (the scaling in Real, as the reference forms it; widening is exact)
    long long r = real_to_int64_fast(u);
    const bool rare = !(__builtin_fabs(u) < TM_FIXED_FAST_LIMIT); // (true for NaN)
    if (__ballot(rare) != 0ull) {
        if (rare) {
            return float_to_fixed_energy<Real>(u_orig);
        }
    }
    return static_cast<i128>(r);
}

__host__ __device__ __forceinline__ bool fixed_point_overflow(i128 v) {
    return v >= static_cast<i128>(LLONG_MAX) || v <= static_cast<i128>(LLONG_MIN);
}

// wave64 sum of a signed 128-bit value (shuffles move 32-bit pieces; integer add is associative => deterministic)
__device__ __forceinline__ i128 wave_sum_i128(i128 v) {
    u64 lo = static_cast<u64>(v);
    long long hi = static_cast<long long>(v >> 64);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        u64 olo = __shfl_down(lo, off, 64);
        long long ohi = __shfl_down(hi, off, 64);
        i128 a = (static_cast<i128>(hi) << 64) | static_cast<i128>(lo);
        i128 b = (static_cast<i128>(ohi) << 64) | static_cast<i128>(olo);
        a += b;
        lo = static_cast<u64>(a);
        hi = static_cast<long long>(a >> 64);
    }
    return (static_cast<i128>(hi) << 64) | static_cast<i128>(lo);
}

} // namespace tmamd
