// Tabulated electrostatic force factor of the f64 nonbonded kernels.
//
//   F(s) = inv * (D'(d) * inv - D(d) * inv^2),   s = d^2,  inv = 1/d,  D(d) = erfc(beta d) * S(d),
//   S(d) = cos^3(pi/2 (d/1.2)^8)   (reference: k_nonbonded_common.cuh:16-94; the 1.2 is hard-coded there)
//
// so that the electrostatic part of a pair's force prefactor is  charge_scale * q_i q_j * F(d^2): a function of d^2 ALONE
// (beta is a property of the potential).  Evaluated analytically it costs erfcx + exp + sincos + rsqrt ~ 70 f64 VALU
// instructions per pair -- almost half of the tile kernel's pair math.  Here it is piecewise polynomial in s:
//   * the intervals follow the floating-point format: ES_TAB_PER_OCTAVE equal intervals per binade of s, so the index
//     and the position t in [0, 1) inside the interval are bit fields of s (no rsqrt, no division, no search), and the
//     relative resolution is the same from d = 0.09 nm to the cutoff;
//   * degree-5 Chebyshev interpolants (6 coefficients = 48 B per interval, 256 intervals = 12 KB: lives in LDS in the
//     tile kernel, L1/L2-resident global memory in the pair-list kernels); fitted on the host in long double from the
//     analytic form above, continued smoothly through d = 1.2 (the kernels then select 0 for s >= 1.44, as the analytic
//     code does), to <= 5e-12 relative -- tighter than the 3e-11 polynomials of the analytic f64 path (nb_math.hip.hpp);
//   * s below the table (d < 0.088 nm: clashing atoms only) takes the analytic form behind a wave-uniform branch.
// Every kernel that evaluates a pair (tiles, pair lists, exclusions, fused plan) goes through es_force_factor() with a
// table made by the same host routine from the same beta, so excluded pairs still cancel bit for bit.
//
// A second table of the same layout holds the ENERGY factor
//   G(s) = D(d) / d        (u_es = charge_scale * q_i q_j * G(d^2);  du/dq_i = charge_scale * q_j * G(d^2))
// for the calls that ask for energies or du/dp (barostat attempts, HREX energy rows, free-energy derivatives): it sits
// behind the force table in the same device array (es_force_table_device() returns 2 * ES_TAB_DOUBLES doubles).
#pragma once
#include "common.hpp"

namespace tmamd {

static const int ES_TAB_BITS = 5;                        // log2(intervals per binade)
static const int ES_TAB_PER_OCTAVE = 1 << ES_TAB_BITS;   // 32
static const int ES_TAB_EXP_LO = -7;                     // first binade: s in [2^-7, 2^-6)  (d >= 0.0884 nm)
static const int ES_TAB_OCTAVES = 8;                     // ... last binade [1, 2): covers s < 1.44 = 1.2^2
static const int ES_TAB_INTERVALS = ES_TAB_PER_OCTAVE * ES_TAB_OCTAVES; // 256
static const int ES_TAB_COEFFS = 6;                      // degree 5
static const int ES_TAB_DOUBLES = ES_TAB_INTERVALS * ES_TAB_COEFFS;     // 1536 doubles = 12 KB
static const unsigned int ES_TAB_IDX0 = static_cast<unsigned int>(1023 + ES_TAB_EXP_LO) << ES_TAB_BITS;
#define TM_ES_TAB_S_MIN 0.0078125          // 2^-7
#define TM_ES_SWITCH_D 1.2                 // k_nonbonded_common.cuh:16-30: the switch ends here whatever `cutoff` is

// host: the tables for `beta` as ONE device array of 2 * ES_TAB_DOUBLES doubles -- force factor F, then energy factor G
// (built once per beta and device, never freed)
const double *es_force_table_device(double beta);
// host: the same coefficients on the host (tests / documentation): ES_TAB_DOUBLES doubles each
void es_force_table_host(double beta, double *out);
void es_energy_table_host(double beta, double *out);

#ifdef __HIPCC__
// index and in-interval position from the bits of s; idx >= ES_TAB_INTERVALS (as unsigned) <=> s outside [2^-7, 2)
__device__ __forceinline__ unsigned int es_tab_index(const double s, double &t) {
    const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(s));
    const unsigned int idx = static_cast<unsigned int>(bits >> (52 - ES_TAB_BITS)) - ES_TAB_IDX0;
    const unsigned long long frac = ((bits & ((1ull << (52 - ES_TAB_BITS)) - 1ull)) << ES_TAB_BITS) | 0x3ff0000000000000ull;
    t = __longlong_as_double(static_cast<long long>(frac)) - 1.0;
    return idx;
}
#endif

} // namespace tmamd
