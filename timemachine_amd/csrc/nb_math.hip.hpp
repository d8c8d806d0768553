// Branch-free f64 special functions for the per-pair nonbonded math on gfx950.
//
// Why not libm (OCML) erfc / exp / sincos: they are general-purpose (full argument range, several branches per
// call) and cost ~270 f64 VALU instructions per pair with divergent paths; the arguments here live in narrow,
// known ranges (x = beta*d in [0, ~3], switch angle in [0, pi/2]), so fixed-degree polynomials evaluated with explicit
// FMAs do the same job in ~90 instructions with no divergence and a fixed, reproducible instruction sequence
// (the exclusion kernel must reproduce the tile kernel's bits).  Coefficients: tools/gen_math_coeffs.py (Chebyshev
// interpolation at 60 digits); degrees chosen so that every piece stays <= 3e-11 (erfcx 1.7e-11 relative, exp 1.1e-12,
// sin 2.7e-11 / cos 7.5e-13 absolute; the script reports them) -- more than two orders of magnitude inside the 1e-8
// contract on forces, and below the 4e-11 the rest of the f64 arithmetic (summation order, fixed-point rounding) leaves.
#pragma once
#include "nb_math_coeffs.h"

namespace tmamd {

// r = a * b + c with the polynomial coefficient c held in a scalar register pair.  Left to itself the compiler emits
// "v_mov_b64 tmp, c ; v_fmac_f64 tmp, a, b" for every Horner step (two-address form) and parks ~55 coefficients in
// ~110 VGPRs; the VOP3 form reads the coefficient straight from SGPRs: one instruction per step, no VGPR cost.
// Inputs and output are plain VALU values (never the direct result of a transcendental op), so no wait states apply.
__device__ __forceinline__ double tm_fma_sc(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

__device__ __forceinline__ double tm_sgpr_to_v(double c) { return c; }

template <int N> __device__ __forceinline__ double tm_horner(const double (&c)[N], double x) {
    double r = tm_fma_sc(x, tm_sgpr_to_v(c[N - 1]), c[N - 2]);
#pragma unroll
    for (int k = N - 3; k >= 0; k--) {
        r = tm_fma_sc(r, x, c[k]);
    }
    return r;
}

// even/odd split: two independent Horner chains in x^2 (instruction-level parallelism for the in-order SIMD)
// same, coefficient held in a VGPR pair (used for one of the polynomials so that the scalar register file is not
// over-subscribed: ~55 coefficients = 110 SGPRs would spill through v_readlane/v_writelane)
__device__ __forceinline__ double tm_fma_vc(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int N> __device__ __forceinline__ double tm_poly_eo_v(const double (&c)[N], double x) {
    const double x2 = x * x;
    constexpr int NE = (N + 1) / 2;
    constexpr int NO = N / 2;
    double e = tm_fma_vc(x2, tm_sgpr_to_v(c[2 * (NE - 1)]), c[2 * (NE - 2)]);
#pragma unroll
    for (int k = NE - 3; k >= 0; k--) {
        e = tm_fma_vc(e, x2, c[2 * k]);
    }
    double o = tm_fma_vc(x2, tm_sgpr_to_v(c[2 * (NO - 1) + 1]), c[2 * (NO - 2) + 1]);
#pragma unroll
    for (int k = NO - 3; k >= 0; k--) {
        o = tm_fma_vc(o, x2, c[2 * k + 1]);
    }
    return __builtin_fma(o, x, e);
}

template <int N> __device__ __forceinline__ double tm_poly_eo(const double (&c)[N], double x) {
    const double x2 = x * x;
    constexpr int NE = (N + 1) / 2; // even-index coefficients
    constexpr int NO = N / 2;       // odd-index coefficients
    double e = tm_fma_sc(x2, tm_sgpr_to_v(c[2 * (NE - 1)]), c[2 * (NE - 2)]);
#pragma unroll
    for (int k = NE - 3; k >= 0; k--) {
        e = tm_fma_sc(e, x2, c[2 * k]);
    }
    double o = tm_fma_sc(x2, tm_sgpr_to_v(c[2 * (NO - 1) + 1]), c[2 * (NO - 2) + 1]);
#pragma unroll
    for (int k = NO - 3; k >= 0; k--) {
        o = tm_fma_sc(o, x2, c[2 * k + 1]);
    }
    return __builtin_fma(o, x, e);
}

// Newton steps on top of the hardware rsq / rcp estimates.  v_rsq_f64 / v_rcp_f64 are good to ~2^-23 relative, so ONE
// step lands at ~2e-14 (error squared, times 3/2 for rsqrt) -- three orders of magnitude below the 3e-11 the polynomials
// leave; the second step bought nothing measurable (force error 3.94e-11 with one or two) and cost 6 f64 VALU
// instructions per pair (-2 us on the tile kernel).
#ifndef TM_NR_STEPS
#define TM_NR_STEPS 1
#endif

// 1/sqrt(x): hardware estimate + Newton  y <- y + y (1/2 - x y^2 / 2)
__device__ __forceinline__ double tm_rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int it = 0; it < TM_NR_STEPS; it++) {
        const double t = x * y;
        const double h = 0.5 * y;
        const double e = __builtin_fma(-t, h, 0.5);
        y = __builtin_fma(y, e, y);
    }
    return y;
}

// 1/x: hardware estimate + Newton  y <- y + y (1 - x y)
__device__ __forceinline__ double tm_rcp_f64(double x) {
    double y = __builtin_amdgcn_rcp(x);
#pragma unroll
    for (int it = 0; it < TM_NR_STEPS; it++) {
        const double e = __builtin_fma(-x, y, 1.0);
        y = __builtin_fma(y, e, y);
    }
    return y;
}

// exp(t) for t <= 0 (underflows cleanly to 0 for very negative t)
__device__ __forceinline__ double tm_exp_neg_f64(double t) {
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01; // 0x3fe62e42fee00000
    const double LN2_LO = 1.90821492927058770002e-10; // 0x3dea39ef35793c76
    const double n = __builtin_rint(t * LOG2E);
    double r = __builtin_fma(n, -LN2_HI, t);
    r = __builtin_fma(n, -LN2_LO, r);
    const double p = tm_poly_eo(TM_EXP_C, r);
    return __builtin_amdgcn_ldexp(p, static_cast<int>(n));
}

// exp(x^2) erfc(x) for x >= 0; accurate to 1.7e-11 on [0, 6]; for x > 6 the argument is clamped (erfc(6) = 2e-17: the
// product with exp(-x^2) is below 1e-17 in absolute terms whatever this returns there)
__device__ __forceinline__ double tm_erfcx_f64(double x) {
    const double xc = x < 6.0 ? x : 6.0;
    const double u = tm_rcp_f64(__builtin_fma(0.5, xc, 1.0));
    const double y = __builtin_fma(u, 8.0 / 3.0, -5.0 / 3.0);
    return tm_poly_eo_v(TM_ERFCX_C, y);
}

// sin and cos of a = (pi/2) q for q in [0, 1], from z = 2 q^2 - 1 (no pi in the argument reduction)
__device__ __forceinline__ void tm_sincos_halfpi_f64(double q, double &s, double &c) {
    const double z = __builtin_fma(q + q, q, -1.0);
    const double HALF_PI = 1.57079632679489661923;
    s = (HALF_PI * q) * tm_horner(TM_SIN_C, z);
    c = tm_horner(TM_COS_C, z);
}

} // namespace tmamd
