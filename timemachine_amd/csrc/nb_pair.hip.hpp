// The single per-pair nonbonded device function.  Both the tile kernel (all pairs) and the pair-list kernel
// (exclusions, subtracted in fixed point) inline THIS function, so a fully excluded pair cancels bit-for-bit
// (reference: the shared compute_electrostatics / compute_lj of cpp/src/kernels/k_nonbonded_common.cuh:180-246 and
// the note at k_nonbonded_pair_list.cuh:3-6).  The translation units are compiled with -ffp-contract=off; every
// fused multiply-add below is explicit, so the instruction sequence cannot differ between call sites.
//
// Math (k_nonbonded_common.cuh:16-94,184-246):
//   inv = rsqrt(d2); d = d2*inv; inv2 = inv*inv
//   S(d)  = cos^3(pi/2 (d/1.2)^8)           (0 for d >= 1.2; the 1.2 is hard-coded, independent of `cutoff`)
//   damp  = erfc(beta d) S(d)
//   es_prefactor = s_q q_i q_j inv [ (erfc S' + (-2 beta/sqrt(pi) e^{-(beta d)^2}) S) inv - damp inv2 ]
//   u_es  = s_q q_i q_j inv damp
//   LJ: eps_ij = e_i e_j, sig_ij = s_i + s_j, s6 = (sig_ij inv)^6
//   lj_prefactor = s_lj eps_ij s6 inv2 (48 s6 - 24);  u_lj = s_lj 4 eps_ij (s6 - 1) s6
//   prefactor = es_prefactor - lj_prefactor
#pragma once
#include "fixed_point.hip.hpp"
#include "nb_es_table.hip.hpp"
#include "nb_math.hip.hpp"

namespace tmamd {

#define TM_PI 3.141592653589793115997963468544185161
#define TM_TWO_OVER_SQRT_PI 1.128379167095512595889238330988549829708

template <typename Real> struct PairOut {
    Real prefactor; // dU/dr / r : force on i is prefactor * delta
    Real u;         // pair energy
    Real inv_dij;
    Real ebd;       // erfc(beta d) * S(d)
    Real sig_grad;  // dU/dsig_i (== dU/dsig_j)
    Real eps_grad;  // dU/deps_i = eps_grad * eps_j ; dU/deps_j = eps_grad * eps_i
    bool has_lj;
};

// ---------------- f64 ----------------
// S(d) = cos^3((pi/2) (d/1.2)^8) and S'(d) = -12 pi/1.2^8 d^7 sin cos^2  (0 for d >= 1.2), branch-free
__device__ __forceinline__ double switch_fn_and_deriv(double dij, double &dsdr) {
    const double cutoff = 1.2;
    const double inv_cutoff = 1.0 / cutoff;
    const double pi = static_cast<double>(TM_PI);
    const double k2 = inv_cutoff * inv_cutoff;
    const double k4 = k2 * k2;
    const double k8 = k4 * k4;
    const bool inside = dij < cutoff;
    const double d = inside ? dij : cutoff; // keep the polynomial argument in range; results are discarded outside
    const double d2 = d * d;
    const double d4 = d2 * d2;
    const double d7 = d4 * d2 * d;
    const double q = (d4 * d4) * k8; // (d/1.2)^8 in [0, 1]
    double s, c;
    tm_sincos_halfpi_f64(q, s, c);
    const double c2 = c * c;
    const double minus_12_pi_k8 = -12 * pi * k8;
    const double ds = d7 * s * c2 * minus_12_pi_k8;
    dsdr = inside ? ds : 0.0;
    return inside ? c2 * c : 0.0;
}

__device__ __forceinline__ double real_es_factor(double beta, double dij, double inv_dij, double inv_d2ij, double &damping) {
    const double bd = beta * dij;
    const double e2 = tm_exp_neg_f64(-(bd * bd)); // exp(-(beta d)^2)
    const double e = e2 * tm_erfcx_f64(bd);       // erfc(beta d)
    double dsdr;
    const double sr = switch_fn_and_deriv(dij, dsdr);
    damping = e * sr;
    const double debd = (-static_cast<double>(TM_TWO_OVER_SQRT_PI) * beta) * e2;
    const double damping_prime = __builtin_fma(e, dsdr, debd * sr);
    return __builtin_fma(damping_prime, inv_dij, -(damping * inv_d2ij));
}

// ---------------- f32 (what production MD runs; k_nonbonded_common.cuh:98-178) ----------------
__device__ __forceinline__ float switch_fn_and_deriv(float dij, float &dsdr) {
    const float cutoff = 1.2f;
    if (dij >= cutoff) {
        dsdr = 0.0f;
        return 0.0f;
    }
    const float pi = static_cast<float>(TM_PI);
    const float inv_cutoff = 1.0f / cutoff;
    const float k2 = inv_cutoff * inv_cutoff;
    const float k4 = k2 * k2;
    const float k8 = k4 * k4;
    float d2 = dij * dij;
    float d4 = d2 * d2;
    float d7 = d4 * d2 * dij;
    float d8 = d4 * d4;
    float arg = (0.5f * pi) * (d8 * k8);
    float s, c;
    __sincosf(arg, &s, &c);
    float c2 = c * c;
    const float minus_12_pi_k8 = -12 * pi * k8;
    dsdr = minus_12_pi_k8 * d7 * s * c2;
    return c2 * c;
}

__device__ __forceinline__ float real_es_factor(float beta, float dij, float inv_dij, float inv_d2ij, float &damping) {
    float x = beta * dij;
    float exp_x2 = __expf(-x * x);
    // Abramowitz & Stegun 7.1.26 (|err| < 1.5e-7), same approximation the reference's f32 path uses
    float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x); // v_rcp_f32: 1 ulp, no division sequence
    float ebd = (0.254829592f + (-0.284496736f + (1.421413741f + (-1.453152027f + 1.061405429f * t) * t) * t) * t) * t * exp_x2;
    float debd = beta * (-static_cast<float>(TM_TWO_OVER_SQRT_PI) * exp_x2);
    float dsdr;
    float sr = switch_fn_and_deriv(dij, dsdr);
    damping = ebd * sr;
    float damping_prime = (ebd * dsdr) + (debd * sr);
    return damping_prime * inv_dij - damping * inv_d2ij;
}

__device__ __forceinline__ double tm_rsqrt(double x) { return tm_rsqrt_f64(x); }
__device__ __forceinline__ float tm_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); } // v_rsq_f32: 1 ulp; d2 is never denormal

// d2ij must already satisfy d2ij < cutoff^2.
template <typename Real>
__device__ __forceinline__ void nb_pair(
    Real charge_scale, Real lj_scale, Real qi, Real qj, Real sig_i, Real sig_j, Real eps_i, Real eps_j, Real d2ij, Real beta,
    PairOut<Real> &o) {
    Real inv_dij = tm_rsqrt(d2ij);
    Real dij = d2ij * inv_dij;
    Real inv_d2ij = inv_dij * inv_dij;
    Real qij = qi * qj;
    Real damping;
    Real es_factor = real_es_factor(beta, dij, inv_dij, inv_d2ij, damping);
    Real es_prefactor = charge_scale * qij * inv_dij * es_factor;
    Real u = charge_scale * qij * inv_dij * damping;
    Real prefactor = es_prefactor;
    o.has_lj = (eps_i != 0 && eps_j != 0);
    o.sig_grad = 0;
    o.eps_grad = 0;
    if (o.has_lj) {
        Real eps_ij = eps_i * eps_j;
        Real sig_ij = sig_i + sig_j;
        Real sig_inv = sig_ij * inv_dij;
        Real sig2 = sig_inv * sig_inv;
        Real sig4 = sig2 * sig2;
        Real sig6 = sig4 * sig2;
        Real sig6_inv_d8 = sig6 * inv_d2ij;
        Real sig5_inv_d6 = sig_ij * sig4 * inv_d2ij;
        Real lj_prefactor = lj_scale * eps_ij * sig6_inv_d8 * (sig6 * 48 - 24);
        u += lj_scale * 4 * eps_ij * (sig6 - 1) * sig6;
        prefactor -= lj_prefactor;
        o.sig_grad = lj_scale * 24 * eps_ij * sig5_inv_d6 * (2 * sig6 - 1);
        o.eps_grad = lj_scale * 4 * (sig6 - 1) * sig6;
    }
    o.prefactor = prefactor;
    o.u = u;
    o.inv_dij = inv_dij;
    o.ebd = damping;
}

// ---------------- f64: electrostatic force factor from the table (nb_es_table.hip.hpp) ----------------
// Where a kernel keeps the table: LDS (tile kernel: a copy made at kernel start) or global memory (pair lists).  Both
// hold the same doubles and feed the same arithmetic, so the two give identical bits.
struct EsTableGlobal {
    const double *__restrict__ tab; // force factor F; the energy factor G follows it (nb_es_table.hip.hpp)
    __device__ __forceinline__ void load(const unsigned int idx, double (&c)[ES_TAB_COEFFS]) const {
        const double2 *p = reinterpret_cast<const double2 *>(tab + static_cast<size_t>(idx) * ES_TAB_COEFFS);
        const double2 a = p[0], b = p[1], e = p[2];
        c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = e.x; c[5] = e.y;
    }
    __device__ __forceinline__ void load_g(const unsigned int idx, double (&c)[ES_TAB_COEFFS]) const {
        const double2 *p = reinterpret_cast<const double2 *>(tab + ES_TAB_DOUBLES + static_cast<size_t>(idx) * ES_TAB_COEFFS);
        const double2 a = p[0], b = p[1], e = p[2];
        c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = e.x; c[5] = e.y;
    }
};
struct EsTableNone {}; // f32 kernels: analytic A&S erfc + hardware exp / sincos, no table

// the analytic form for d2 below the table (clashing atoms).  Deliberately NOT inlined: its ~60 polynomial coefficients
// would otherwise be hoisted out of the callers' hot loops and sit in registers there for good.
__device__ __attribute__((noinline, cold)) double es_force_factor_below_table(const double beta, const double d2) {
    const double inv = tm_rsqrt_f64(d2);
    double damping;
    return inv * real_es_factor(beta, d2 * inv, inv, inv * inv, damping);
}

// the degree-5 polynomial of one table interval at in-interval position t.  ONE definition for every kernel and both tables
// (excluded pairs cancel bit for bit only if every call site rounds alike).
__device__ __forceinline__ double es_tab_poly(const double (&c)[ES_TAB_COEFFS], const double t) {
#ifdef TM_ESTRIN
    // Estrin: three dependent levels instead of Horner's five (one multiplication more)
    const double t2 = t * t;
    const double a = __builtin_fma(c[1], t, c[0]);
    const double b = __builtin_fma(c[3], t, c[2]);
    const double e = __builtin_fma(c[5], t, c[4]);
    return __builtin_fma(__builtin_fma(e, t2, b), t2, a);
#else
    double p = __builtin_fma(c[5], t, c[4]);
    p = __builtin_fma(p, t, c[3]);
    p = __builtin_fma(p, t, c[2]);
    p = __builtin_fma(p, t, c[1]);
    return __builtin_fma(p, t, c[0]);
#endif
}

// F(d2) such that the electrostatic force prefactor of a pair is charge_scale * q_i q_j * F(d2): the table part, without a
// branch.  `below` says that d2 lies under the table (the caller owes the analytic form; the value returned is then 0).
// INSIDE_SWITCH: the caller knows d2 < TM_ES_SWITCH_D^2 (its cutoff is not beyond the end of the switch): d2 can then only
// leave the table downwards, and the value returned for such a lane is garbage instead of 0 -- three selects less.
template <bool INSIDE_SWITCH = false, typename Tab> __device__ __forceinline__ double es_force_factor_table(const double d2, const Tab &tab, bool &below) {
    double t;
    unsigned int idx = es_tab_index(d2, t);
    const bool outside = idx >= static_cast<unsigned int>(ES_TAB_INTERVALS); // d2 < 2^-7 or d2 >= 2 (or NaN)
    if constexpr (INSIDE_SWITCH) {
        idx = idx < static_cast<unsigned int>(ES_TAB_INTERVALS) ? idx : static_cast<unsigned int>(ES_TAB_INTERVALS - 1);
    } else {
        idx = outside ? 0u : idx;
    }
    double c[ES_TAB_COEFFS];
#if defined(TM_ABLATE) && (TM_ABLATE == 11 || TM_ABLATE == 13) // ablation (timing only): table reads at conflict-free addresses (16 consecutive intervals per lane group)
    tab.load(static_cast<unsigned int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))) & 15u, c);
#else
    tab.load(idx, c);
#endif
    const double p = es_tab_poly(c, t);
    if constexpr (INSIDE_SWITCH) {
        below = outside;
        return p;
    } else {
        const double switch_end2 = static_cast<double>(TM_ES_SWITCH_D) * static_cast<double>(TM_ES_SWITCH_D);
        below = outside && d2 < static_cast<double>(TM_ES_TAB_S_MIN);
        return (d2 < switch_end2 && !outside) ? p : 0.0; // beyond the switch the damping function is exactly zero
    }
}

template <typename Tab> __device__ __forceinline__ double es_force_factor(const double beta, const double d2, const Tab &tab) {
    bool below;
    double f = es_force_factor_table(d2, tab, below);
    // below the table: clashing atoms (d < 0.088 nm) only -- the analytic form, behind a wave-uniform branch
    if (__ballot(below) != 0ull) {
        if (below) {
            f = es_force_factor_below_table(beta, d2);
        }
    }
    return f;
}

// G(d2) = erfc(beta d) S(d) / d from the second table: the electrostatic energy of a pair is charge_scale * q_i q_j * G(d2)
// and du/dq_i = charge_scale * q_j * G(d2).  0 beyond the end of the switch; the analytic form under the table.
__device__ __attribute__((noinline, cold)) double es_energy_factor_below_table(const double beta, const double d2) {
    const double inv = tm_rsqrt_f64(d2);
    double damping;
    (void)real_es_factor(beta, d2 * inv, inv, inv * inv, damping);
    return inv * damping;
}
template <typename Tab> __device__ __forceinline__ double es_energy_factor(const double beta, const double d2, const Tab &tab) {
    double t;
    unsigned int idx = es_tab_index(d2, t);
    const bool outside = idx >= static_cast<unsigned int>(ES_TAB_INTERVALS); // d2 < 2^-7 or d2 >= 2 (or NaN)
    idx = outside ? 0u : idx;
    double c[ES_TAB_COEFFS];
    tab.load_g(idx, c);
    const double p = es_tab_poly(c, t);
    const double switch_end2 = static_cast<double>(TM_ES_SWITCH_D) * static_cast<double>(TM_ES_SWITCH_D);
    double g = (d2 < switch_end2 && !outside) ? p : 0.0; // beyond the switch the damping function is exactly zero
    const bool below = outside && d2 < static_cast<double>(TM_ES_TAB_S_MIN);
    if (__ballot(below) != 0ull) { // clashing atoms (d < 0.088 nm) only
        if (below) {
            g = es_energy_factor_below_table(beta, d2);
        }
    }
    return g;
}

// The forces-only f64 pair function with its two rare cases left to the caller (the tile kernel folds them into ONE
// wave-uniform escape together with the fixed-point overflow case): the operations of nb_pair<false> below in the same
// order, so the same bits.  `below`: d2 lies under the table, the value returned is not to be used --
// nb_pair_prefactor_below_table() gives the prefactor then.
template <bool INSIDE_SWITCH, typename Tab>
__device__ __forceinline__ double nb_pair_prefactor_deferred(
    const double charge_scale, const double lj_scale, const double qi, const double qj, const double sig_i, const double sig_j,
    const double eps_i, const double eps_j, const double d2ij, const Tab &tab, bool &below) {
    const double qij = qi * qj;
#ifdef TM_LJ_EARLY
    // experiment: the table's LDS reads are ISSUED first, the Lennard-Jones chain (rcp + nine dependent multiplications, written
    // without a branch) runs underneath their round trip, the polynomial follows; the same operations on the same operands for
    // the lanes that have an LJ term, a select for the others: same bits
    if constexpr (INSIDE_SWITCH) {
        double t;
        unsigned int idx = es_tab_index(d2ij, t);
        below = idx >= static_cast<unsigned int>(ES_TAB_INTERVALS);
        idx = idx < static_cast<unsigned int>(ES_TAB_INTERVALS) ? idx : static_cast<unsigned int>(ES_TAB_INTERVALS - 1);
        double c[ES_TAB_COEFFS];
        tab.load(idx, c);
        const double inv_d2 = tm_rcp_f64(d2ij);
        const double eps_ij = eps_i * eps_j;
        const double sig_ij = sig_i + sig_j;
        const double sig2 = (sig_ij * sig_ij) * inv_d2;
        const double sig4 = sig2 * sig2;
        const double sig6 = sig4 * sig2;
        const double lj_prefactor = lj_scale * eps_ij * (sig6 * inv_d2) * (sig6 * 48 - 24);
        const bool has_lj = eps_i != 0 && eps_j != 0;
        const double es_prefactor = charge_scale * qij * es_tab_poly(c, t);
        return has_lj ? es_prefactor - lj_prefactor : es_prefactor;
    }
#endif
    const double es_prefactor = charge_scale * qij * es_force_factor_table<INSIDE_SWITCH>(d2ij, tab, below);
    const double inv_d2ij = tm_rcp_f64(d2ij);
    double prefactor = es_prefactor;
#if defined(TM_ABLATE) && TM_ABLATE == 9 // ablation (timing only): what the MD kernel would gain if no batch ever ran the LJ block
    if (false) {
#else
    if (eps_i != 0 && eps_j != 0) {
#endif
        const double eps_ij = eps_i * eps_j;
        const double sig_ij = sig_i + sig_j;
        const double sig2 = (sig_ij * sig_ij) * inv_d2ij;
        const double sig4 = sig2 * sig2;
        const double sig6 = sig4 * sig2;
        const double lj_prefactor = lj_scale * eps_ij * (sig6 * inv_d2ij) * (sig6 * 48 - 24);
        prefactor -= lj_prefactor;
    }
    return prefactor;
}
__device__ __attribute__((noinline, cold)) double nb_pair_prefactor_below_table(
    const double charge_scale, const double lj_scale, const double qi, const double qj, const double sig_i, const double sig_j,
    const double eps_i, const double eps_j, const double d2ij, const double beta) {
    const double qij = qi * qj;
    const double es_prefactor = charge_scale * qij * es_force_factor_below_table(beta, d2ij);
    const double inv_d2ij = tm_rcp_f64(d2ij);
    double prefactor = es_prefactor;
    if (eps_i != 0 && eps_j != 0) {
        const double eps_ij = eps_i * eps_j;
        const double sig_ij = sig_i + sig_j;
        const double sig2 = (sig_ij * sig_ij) * inv_d2ij;
        const double sig4 = sig2 * sig2;
        const double sig6 = sig4 * sig2;
        const double lj_prefactor = lj_scale * eps_ij * (sig6 * inv_d2ij) * (sig6 * 48 - 24);
        prefactor -= lj_prefactor;
    }
    return prefactor;
}

// The f64 pair function.  WANT_U_DP = false (MD: forces only) never forms 1/d, erfc, exp or the switch function:
//   prefactor = s_q q_i q_j F(d2) - s_lj eps_ij sig6 (48 sig6 - 24) / d2,   sig6 = (sig_ij^2 / d2)^3
// WANT_U_DP = true adds, on top of the SAME prefactor arithmetic (so du/dx has the same bits whichever outputs are asked
// for), the tabulated energy factor G(d2) for the energy and du/dq, and the LJ parameter derivatives.
// WANT_PREFACTOR = false (energy-only launches: barostat attempts, energy matrices): the force factor is not formed at all -- its
// table fetch sits behind a wave-wide ballot and a call (the below-the-table escape), which the compiler must keep even when
// nobody reads the result; o.prefactor is then 0.  The energy's own arithmetic is untouched: the same bits.
template <bool WANT_U_DP, bool WANT_PREFACTOR = true, typename Tab>
__device__ __forceinline__ void nb_pair(
    double charge_scale, double lj_scale, double qi, double qj, double sig_i, double sig_j, double eps_i, double eps_j, double d2ij,
    double beta, PairOut<double> &o, const Tab &tab) {
    const double qij = qi * qj;
    double es_prefactor = 0;
    if constexpr (WANT_PREFACTOR) {
        es_prefactor = charge_scale * qij * es_force_factor(beta, d2ij, tab);
    }
    const double inv_d2ij = tm_rcp_f64(d2ij);
    double prefactor = es_prefactor;
    double u = 0;
    o.has_lj = (eps_i != 0 && eps_j != 0);
    o.sig_grad = 0;
    o.eps_grad = 0;
    o.inv_dij = 0;
    o.ebd = 0;
    if (o.has_lj) {
        const double eps_ij = eps_i * eps_j;
        const double sig_ij = sig_i + sig_j;
        const double sig2 = (sig_ij * sig_ij) * inv_d2ij;
        const double sig4 = sig2 * sig2;
        const double sig6 = sig4 * sig2;
        if constexpr (WANT_PREFACTOR) {
            const double lj_prefactor = lj_scale * eps_ij * (sig6 * inv_d2ij) * (sig6 * 48 - 24);
            prefactor -= lj_prefactor;
        }
        if constexpr (WANT_U_DP) {
            u = lj_scale * 4 * eps_ij * (sig6 - 1) * sig6;
            o.sig_grad = lj_scale * 24 * eps_ij * (sig_ij * sig4 * inv_d2ij) * (2 * sig6 - 1);
            o.eps_grad = lj_scale * 4 * (sig6 - 1) * sig6;
        }
    }
    if constexpr (WANT_U_DP) {
        // (PairOut carries the factor as inv_dij * ebd: the callers' q_j * inv_dij * ebd is q_j * G exactly)
        const double g = es_energy_factor(beta, d2ij, tab);
        u += charge_scale * qij * g;
        o.inv_dij = 1.0;
        o.ebd = g;
    }
    o.prefactor = prefactor;
    o.u = u;
}
// f32: the analytic pair function above, whatever is asked for (the table argument is an empty tag)
template <bool WANT_U_DP, bool WANT_PREFACTOR = true, typename AnyTab>
__device__ __forceinline__ void nb_pair(
    float charge_scale, float lj_scale, float qi, float qj, float sig_i, float sig_j, float eps_i, float eps_j, float d2ij, float beta,
    PairOut<float> &o, const AnyTab &) {
    nb_pair<float>(charge_scale, lj_scale, qi, qj, sig_i, sig_j, eps_i, eps_j, d2ij, beta, o);
}

// minimum-image displacement component: delta -= L * nearbyint(delta / L)   (round-half-even, Appendix B.2).
// Odd in delta bit for bit (rint and fma are sign-symmetric), so swapping i and j only flips signs.
__device__ __forceinline__ double min_image(double delta, double box, double inv_box) {
    return __builtin_fma(-box, __builtin_rint(delta * inv_box), delta);
}
__device__ __forceinline__ float min_image(float delta, float box, float inv_box) {
    return __builtin_fmaf(-box, __builtin_rintf(delta * inv_box), delta);
}

// Fixed-point force components of one pair.
//   reference: g_i += FIX(p * d), g_j += FIX(-p * d) with FIX(v) = llrint(v * 2^36)   (k_nonbonded.cuh:244-252)
// Scaling by 2^36 is exact, so llrint((p * 2^36) * d) == llrint((p * d) * 2^36) bit for bit: one multiply by 2^36 per
// pair instead of one per component; and llrint is odd, so g_j is the two's-complement negation of g_i (callers use
// an integer subtract / LDS ds_sub for it) -- three conversions per pair instead of six.
__device__ __forceinline__ void pair_force_fixed_fast(double prefactor, double dx, double dy, double dz, u64 &fx, u64 &fy, u64 &fz, bool &big) {
    const double ps = prefactor * static_cast<double>(TM_FIXED_EXPONENT);
    const double a = ps * dx, b = ps * dy, c = ps * dz;
    fx = static_cast<u64>(real_to_int64_fast(a));
    fy = static_cast<u64>(real_to_int64_fast(b));
    fz = static_cast<u64>(real_to_int64_fast(c));
    big = !(__builtin_fabs(a) < TM_FIXED_FAST_LIMIT && __builtin_fabs(b) < TM_FIXED_FAST_LIMIT && __builtin_fabs(c) < TM_FIXED_FAST_LIMIT);
}
// The same for a caller that knows a bound on the displacement: with |dx|, |dy|, |dz| < cutoff (they are components of a
// vector that passed d2 < cutoff^2) and ps_limit = 2^51 / cutoff less a hair, |ps| < ps_limit implies that all three products
// are in the fast conversion's range -- one comparison instead of three.  `big` is then a superset of the exact condition;
// the slow path is exact for every value, so the bits do not depend on which path converts.
__device__ __forceinline__ void pair_force_fixed_fast_bounded(double prefactor, double dx, double dy, double dz, double ps_limit, u64 &fx, u64 &fy, u64 &fz, bool &big) {
    const double ps = prefactor * static_cast<double>(TM_FIXED_EXPONENT);
    fx = static_cast<u64>(real_to_int64_fast(ps * dx));
    fy = static_cast<u64>(real_to_int64_fast(ps * dy));
    fz = static_cast<u64>(real_to_int64_fast(ps * dz));
    big = !(__builtin_fabs(ps) < ps_limit);
}
__device__ __forceinline__ void pair_force_fixed_slow(double prefactor, double dx, double dy, double dz, u64 &fx, u64 &fy, u64 &fz) {
    const double ps = prefactor * static_cast<double>(TM_FIXED_EXPONENT);
    fx = static_cast<u64>(tm_llrint_odd(ps * dx));
    fy = static_cast<u64>(tm_llrint_odd(ps * dy));
    fz = static_cast<u64>(tm_llrint_odd(ps * dz));
}
__device__ __forceinline__ void pair_force_fixed(double prefactor, double dx, double dy, double dz, u64 &fx, u64 &fy, u64 &fz) {
    bool big;
    pair_force_fixed_fast(prefactor, dx, dy, dz, fx, fy, fz, big);
    // one wave-uniform escape for all three components (see real_to_int64)
    if (__ballot(big) != 0ull) {
        if (big) {
            pair_force_fixed_slow(prefactor, dx, dy, dz, fx, fy, fz);
        }
    }
}
// f32 kernels: each product is rounded to f32 first, exactly as the reference forms it (RealType prefactor * RealType delta,
// k_nonbonded.cuh:248-254, then FLOAT_TO_FIXED_NONBONDED<float>, k_fixed_point.cuh:65-67); the scaling by 2^36 that follows
// is exact in either precision, so it is done after the (exact) widening.  FIX(-v) == -FIX(v) still holds.
__device__ __forceinline__ void pair_force_fixed(float prefactor, float dx, float dy, float dz, u64 &fx, u64 &fy, u64 &fz) {
    const double a = static_cast<double>(prefactor * dx) * static_cast<double>(TM_FIXED_EXPONENT);
    const double b = static_cast<double>(prefactor * dy) * static_cast<double>(TM_FIXED_EXPONENT);
    const double c = static_cast<double>(prefactor * dz) * static_cast<double>(TM_FIXED_EXPONENT);
    fx = static_cast<u64>(real_to_int64_fast(a));
    fy = static_cast<u64>(real_to_int64_fast(b));
    fz = static_cast<u64>(real_to_int64_fast(c));
    const bool big = !(__builtin_fabs(a) < TM_FIXED_FAST_LIMIT && __builtin_fabs(b) < TM_FIXED_FAST_LIMIT &&
                       __builtin_fabs(c) < TM_FIXED_FAST_LIMIT);
    if (__ballot(big) != 0ull) {
        if (big) {
            fx = static_cast<u64>(tm_llrint_odd(a));
            fy = static_cast<u64>(tm_llrint_odd(b));
            fz = static_cast<u64>(tm_llrint_odd(c));
        }
    }
}
// the same with ONE range test: |prefactor * d| <= |prefactor| * cutoff for every component of a pair inside the cutoff, so
// |prefactor| below `limit` = 2^51 / 2^36 / cutoff (less a hair) keeps all three products in the fast conversion's range
__device__ __forceinline__ void pair_force_fixed_bounded(float prefactor, float dx, float dy, float dz, const float limit, u64 &fx, u64 &fy, u64 &fz) {
    const double a = static_cast<double>(prefactor * dx) * static_cast<double>(TM_FIXED_EXPONENT);
    const double b = static_cast<double>(prefactor * dy) * static_cast<double>(TM_FIXED_EXPONENT);
    const double c = static_cast<double>(prefactor * dz) * static_cast<double>(TM_FIXED_EXPONENT);
    fx = static_cast<u64>(real_to_int64_fast(a));
    fy = static_cast<u64>(real_to_int64_fast(b));
    fz = static_cast<u64>(real_to_int64_fast(c));
    const bool big = !(__builtin_fabsf(prefactor) < limit); // (true for NaN)
    if (__ballot(big) != 0ull) {
        if (big) {
            fx = static_cast<u64>(tm_llrint_odd(a));
            fy = static_cast<u64>(tm_llrint_odd(b));
            fz = static_cast<u64>(tm_llrint_odd(c));
        }
    }
}
__device__ __forceinline__ void pair_force_fixed_bounded(double prefactor, double dx, double dy, double dz, const double, u64 &fx, u64 &fy, u64 &fz) {
    pair_force_fixed(prefactor, dx, dy, dz, fx, fy, fz);
}

} // namespace tmamd
