// HarmonicBond / HarmonicAngle / PeriodicTorsion device code for gfx950.  Included by bonded.hip and fused.hip.
// reference kernels: cpp/src/kernels/k_harmonic_bond.cuh:6-60, k_harmonic_angle.cuh:9-146, k_periodic_torsion.cuh:20-131
// reference JAX energies: timemachine/potentials/bonded.py:34-216
//
// One thread per term; fixed-point u64 atomics into du_dx / du_dp; energies are summed per wave (64 lanes) into
// signed-128 partials and then reduced -- one 16-byte store per wave instead of one per term.
// Compiled with -ffp-contract=off: a*b - c*d style expressions are never fused, which is what keeps the
// index-reversal bitwise symmetry the reference gets from __dmul_rn/__dadd_rn (cpp/src/gpu_utils.cuh:111-121).
#pragma once
#include "engine.hpp"
#include "fixed_point.hip.hpp"

namespace tmamd {

template <typename Real> __device__ __forceinline__ Real tm_sqrt(Real x);
template <> __device__ __forceinline__ float tm_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double tm_sqrt<double>(double x) { return sqrt(x); }

template <typename Real> __device__ __forceinline__ void store_wave_energy(i128 e, i128 *__restrict__ u_partials) {
    const i128 total = wave_sum_i128(e);
    if ((threadIdx.x & 63) == 0) {
        u_partials[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = total;
    }
}

// Where a term's force goes: component d of atom a at du_dx[a * atom + d * comp].  {3, 1} is the [N, 3] array of the
// Potential interface; the integrators' own accumulator is component-major {1, stride}: lanes working on neighbouring
// atoms then share cache lines, and memory-side atomics are served one 64-byte line request at a time
// (scripts/microbench/atomic_scope.hip: 64 lanes on 64 different lines cost 6.6x what 64 lanes on 8 lines cost).
// `remap` (optional): the accumulator is indexed by remap[atom] instead of atom -- a nonbonded potential that covers every atom
// takes the step's bonded terms into its own Hilbert-ordered accumulator (remap = its slot_of_atom), and the integrator then
// finds the whole force of an atom in one place.
// `win` (optional; fused evaluation, see fused_dispatch in kernels_nonbonded.hip.hpp): a per-wave LDS window of FORCE_WINDOW
// atoms starting at atom index win_base, component d of atom a at win[d * FORCE_WINDOW + (a - win_base)].  Forces on atoms
// inside the window are added there (ds_add_u64) and reach the global accumulator once per (atom, component) per 64-term
// slice.  Memory-side atomics on one 64-byte line are served one after the other (~70 ns each): the ~100 bonded terms and
// exclusions that touch every atom of a protein-like solute otherwise queue up on a few hundred lines -- measured on the
// DHFR-shaped bench box, 248 000 atomics on 934 lines: 18.6 us of an 84.6 us f64 tile launch (scripts/fused_cost.py,
// ablation build TM_ABLATE=5).  Integer sums: the window changes no bit.
static const int FORCE_WINDOW = 96;
typedef __attribute__((address_space(3))) u64 *lds_u64_ptr;
struct ForceLayout {
    int atom, comp;
    const int *remap = nullptr;
    lds_u64_ptr win = nullptr;
    int win_base = 0;
    __device__ __forceinline__ size_t row(const int a) const { return static_cast<size_t>(remap ? remap[a] : a) * atom; }
};
__device__ __forceinline__ void force_add(u64 *__restrict__ du_dx, const ForceLayout fl, const int atom, const int d, const u64 v) {
#if defined(TM_ABLATE) && TM_ABLATE == 5
    if (v != 0x123456789abcull) { // ablation (timing only): the term is computed, its atomics are not issued
        return;
    }
#endif
    if (fl.win != nullptr) {
        const unsigned int off = static_cast<unsigned int>(atom - fl.win_base);
        if (off < static_cast<unsigned int>(FORCE_WINDOW)) {
            __hip_atomic_fetch_add(fl.win + d * FORCE_WINDOW + off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
    }
    atomicAdd(du_dx + fl.row(atom) + static_cast<size_t>(d) * fl.comp, v);
}

// term `b` of the list; returns its energy in fixed point (0 unless want_u)
template <typename Real>
__device__ __forceinline__ i128 harmonic_bond_term(
    const int b, const double *__restrict__ coords, const double *__restrict__ params, const int *__restrict__ bond_idxs,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, const bool want_u,
    const ForceLayout fl = ForceLayout{3, 1}) {
    i128 energy = 0;
    const int src = bond_idxs[b * 2 + 0], dst = bond_idxs[b * 2 + 1];
    Real dx[3];
    Real d2 = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        // the displacement is formed in double and rounded once (k_harmonic_bond.cuh:27): the f32 kernels then keep
        // ~1e-8 nm of resolution however far the unwrapped coordinates have drifted
        const Real delta = static_cast<Real>(coords[src * 3 + d] - coords[dst * 3 + d]);
        dx[d] = delta;
        d2 += delta * delta;
    }
    const Real kb = static_cast<Real>(params[b * 2 + 0]);
    const Real b0 = static_cast<Real>(params[b * 2 + 1]);
    const Real dij = tm_sqrt<Real>(d2);
    const Real db = dij - b0;
    if (du_dx) {
        const Real inv_dij = 1 / dij;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const Real g = b0 != 0 ? kb * db * dx[d] * inv_dij : kb * dx[d];
            force_add(du_dx, fl, src, d, float_to_fixed<Real>(g));
            force_add(du_dx, fl, dst, d, float_to_fixed<Real>(-g));
        }
    }
    if (du_dp) {
        atomicAdd(du_dp + b * 2 + 0, float_to_fixed<Real>(static_cast<Real>(0.5) * db * db));
        atomicAdd(du_dp + b * 2 + 1, float_to_fixed<Real>(-kb * db));
    }
    if (want_u) {
        energy = float_to_fixed_energy<Real>(kb / 2 * db * db);
    }

    return energy;
}

template <typename Real>
__global__ __launch_bounds__(256) void k_harmonic_bond(
    const int B, const double *__restrict__ coords, const double *__restrict__ params, const int *__restrict__ bond_idxs,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, i128 *__restrict__ u_partials) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    i128 energy = 0;
    if (b < B) {
        energy = harmonic_bond_term<Real>(b, coords, params, bond_idxs, du_dx, du_dp, u_partials != nullptr);
    }
    if (u_partials) {
        store_wave_energy<Real>(energy, u_partials);
    }
}

// term `a_idx` of the list; returns its energy in fixed point (0 unless want_u)
template <typename Real>
__device__ __forceinline__ i128 harmonic_angle_term(
    const int a_idx, const double *__restrict__ coords, const double *__restrict__ params, const int *__restrict__ angle_idxs,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, const bool want_u,
    const ForceLayout fl = ForceLayout{3, 1}) {
    i128 energy = 0;
    const int i = angle_idxs[a_idx * 3 + 0], j = angle_idxs[a_idx * 3 + 1], k = angle_idxs[a_idx * 3 + 2];
    const Real ka = static_cast<Real>(params[a_idx * 3 + 0]);
    const Real a0 = static_cast<Real>(params[a_idx * 3 + 1]);
    const Real eps = static_cast<Real>(params[a_idx * 3 + 2]);

    // 4-D vectors j->i and j->k with eps as the stabilising 4th component (bonded.py:82-97)
    Real rji[4], rjk[4];
    Real nji = 0, njk = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const double cj = coords[j * 3 + d]; // differences in double, rounded once (k_harmonic_angle.cuh:44-45)
        rji[d] = static_cast<Real>(coords[i * 3 + d] - cj);
        rjk[d] = static_cast<Real>(coords[k * 3 + d] - cj);
        nji += rji[d] * rji[d];
        njk += rjk[d] * rjk[d];
    }
    rji[3] = eps;
    rjk[3] = eps;
    nji = tm_sqrt<Real>(nji + eps * eps);
    njk = tm_sqrt<Real>(njk + eps * eps);

    // Kahan's angle: 2 atan2(|njk rji - nji rjk|, |njk rji + nji rjk|); products are rounded separately
    // (no FMA) so swapping i and k reproduces the same bits
    Real hi = 0, lo = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const Real p = njk * rji[d];
        const Real q = nji * rjk[d];
        const Real s = p - q;
        const Real t = p + q;
        hi += s * s;
        lo += t * t;
    }
    const Real angle = 2 * atan2(tm_sqrt<Real>(hi), tm_sqrt<Real>(lo));
    const Real delta = angle - a0;

    // gradient direction via a x (a x b) = a (a.b) - b (a.a)
    Real a_dot_b = 0, a_dot_a = 0, b_dot_b = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        a_dot_b += rji[d] * rjk[d];
        a_dot_a += rji[d] * rji[d];
        b_dot_b += rjk[d] * rjk[d];
    }
    Real aab[4], bba[4];
    Real aab_n = 0, bba_n = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        aab[d] = rji[d] * a_dot_b - rjk[d] * a_dot_a;
        bba[d] = rjk[d] * a_dot_b - rji[d] * b_dot_b;
        aab_n += aab[d] * aab[d];
        bba_n += bba[d] * bba[d];
    }
    aab_n = tm_sqrt<Real>(aab_n);
    bba_n = tm_sqrt<Real>(bba_n);
    const Real prefactor = ka * delta;
    const Real coeff_i = prefactor * (1 / nji);
    const Real coeff_k = prefactor * (1 / njk);

    if (du_dx) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const Real gi = coeff_i * ((aab_n == 0) ? static_cast<Real>(0) : aab[d] / aab_n);
            const Real gk = coeff_k * ((bba_n == 0) ? static_cast<Real>(0) : bba[d] / bba_n);
            force_add(du_dx, fl, i, d, float_to_fixed<Real>(gi));
            force_add(du_dx, fl, k, d, float_to_fixed<Real>(gk));
            force_add(du_dx, fl, j, d, float_to_fixed<Real>(-gi - gk));
        }
    }
    if (du_dp) {
        atomicAdd(du_dp + a_idx * 3 + 0, float_to_fixed<Real>(delta * delta / 2));
        atomicAdd(du_dp + a_idx * 3 + 1, float_to_fixed<Real>(-delta * ka));
        const Real e0 = (aab_n == 0) ? static_cast<Real>(0) : coeff_i * aab[3] / aab_n;
        const Real e1 = (bba_n == 0) ? static_cast<Real>(0) : coeff_k * bba[3] / bba_n;
        atomicAdd(du_dp + a_idx * 3 + 2, float_to_fixed<Real>(e0 + e1));
    }
    if (want_u) {
        energy = float_to_fixed_energy<Real>((ka / 2) * delta * delta);
    }

    return energy;
}

template <typename Real>
__global__ __launch_bounds__(256) void k_harmonic_angle(
    const int A, const double *__restrict__ coords, const double *__restrict__ params, const int *__restrict__ angle_idxs,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, i128 *__restrict__ u_partials) {
    const int a_idx = blockIdx.x * blockDim.x + threadIdx.x;
    i128 energy = 0;
    if (a_idx < A) {
        energy = harmonic_angle_term<Real>(a_idx, coords, params, angle_idxs, du_dx, du_dp, u_partials != nullptr);
    }
    if (u_partials) {
        store_wave_energy<Real>(energy, u_partials);
    }
}

template <typename Real> __device__ __forceinline__ Real dot3(const Real a[3], const Real b[3]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
// every product individually rounded => cross(a, b) == -cross(b, a) bit for bit
template <typename Real> __device__ __forceinline__ void cross3(const Real a[3], const Real b[3], Real c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// term `t_idx` of the list; returns its energy in fixed point (0 unless want_u)
template <typename Real>
__device__ __forceinline__ i128 periodic_torsion_term(
    const int t_idx, const double *__restrict__ coords, const double *__restrict__ params, const int *__restrict__ torsion_idxs,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, const bool want_u,
    const ForceLayout fl = ForceLayout{3, 1}) {
    i128 energy = 0;
    const int i = torsion_idxs[t_idx * 4 + 0], j = torsion_idxs[t_idx * 4 + 1];
    const int k = torsion_idxs[t_idx * 4 + 2], l = torsion_idxs[t_idx * 4 + 3];
    Real rij[3], rkj[3], rkl[3];
    Real rkj_n2 = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const double ci = coords[i * 3 + d], cj = coords[j * 3 + d]; // differences in double, rounded once
        const double ck = coords[k * 3 + d], cl = coords[l * 3 + d]; // (k_periodic_torsion.cuh:49-51)
        rij[d] = static_cast<Real>(cj - ci);
        rkj[d] = static_cast<Real>(cj - ck);
        rkl[d] = static_cast<Real>(cl - ck);
        rkj_n2 += rkj[d] * rkj[d];
    }
    const Real rkj_norm = tm_sqrt<Real>(rkj_n2);
    Real n1[3], n2[3], n3[3];
    cross3(rij, rkj, n1);
    cross3(rkj, rkl, n2);
    cross3(n1, n2, n3);
    const Real n1_n2sq = dot3(n1, n1), n2_n2sq = dot3(n2, n2);
    const Real rij_dot_rkj = dot3(rij, rkj), rkl_dot_rkj = dot3(rkl, rkj);

    Real dR0[3], dR1[3], dR2[3], dR3[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        dR0[d] = rkj_norm / n1_n2sq * n1[d];
        dR3[d] = -rkj_norm / n2_n2sq * n2[d];
        dR1[d] = (rij_dot_rkj / rkj_n2 - 1) * dR0[d] - dR3[d] * rkl_dot_rkj / rkj_n2;
        dR2[d] = (rkl_dot_rkj / rkj_n2 - 1) * dR3[d] - dR0[d] * rij_dot_rkj / rkj_n2;
    }
    const Real rkj_n = tm_sqrt<Real>(dot3(rkj, rkj));
    Real rkj_hat[3] = {rkj[0] / rkj_n, rkj[1] / rkj_n, rkj[2] / rkj_n};
    const Real angle = atan2(dot3(n3, rkj_hat), dot3(n1, n2));

    const Real kt = static_cast<Real>(params[t_idx * 3 + 0]);
    const Real phase = static_cast<Real>(params[t_idx * 3 + 1]);
    const Real period = static_cast<Real>(params[t_idx * 3 + 2]);
    const Real arg = period * angle - phase;
    const Real s = sin(arg), c = cos(arg);
    const Real prefactor = kt * s * period;

    if (du_dx) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            force_add(du_dx, fl, i, d, float_to_fixed<Real>(dR0[d] * prefactor));
            force_add(du_dx, fl, j, d, float_to_fixed<Real>(dR1[d] * prefactor));
            force_add(du_dx, fl, k, d, float_to_fixed<Real>(dR2[d] * prefactor));
            force_add(du_dx, fl, l, d, float_to_fixed<Real>(dR3[d] * prefactor));
        }
    }
    if (du_dp) {
        atomicAdd(du_dp + t_idx * 3 + 0, float_to_fixed<Real>(1 + c));
        atomicAdd(du_dp + t_idx * 3 + 1, float_to_fixed<Real>(kt * s));
        atomicAdd(du_dp + t_idx * 3 + 2, float_to_fixed<Real>(-kt * s * angle));
    }
    if (want_u) {
        energy = float_to_fixed_energy<Real>(kt * (1 + c));
    }

    return energy;
}

template <typename Real>
__global__ __launch_bounds__(256) void k_periodic_torsion(
    const int T, const double *__restrict__ coords, const double *__restrict__ params, const int *__restrict__ torsion_idxs,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, i128 *__restrict__ u_partials) {
    const int t_idx = blockIdx.x * blockDim.x + threadIdx.x;
    i128 energy = 0;
    if (t_idx < T) {
        energy = periodic_torsion_term<Real>(t_idx, coords, params, torsion_idxs, du_dx, du_dp, u_partials != nullptr);
    }
    if (u_partials) {
        store_wave_energy<Real>(energy, u_partials);
    }
}

// ---- chiral restraints --------------------------------------------------------------------------------------
// reference: cpp/src/kernels/k_chiral_restraint.cuh:9-183, chiral_utils.cuh:92-177; JAX: potentials/chiral_restraints.py:9-125
// U = k vol^2 where the normalised chiral volume has the penalised sign, else 0.
//
// Gradients are written in closed form instead of chaining 3x3 Jacobians: for a unit vector u^ = u/|u| and a scalar
// f(u^, ...), df/du = (g - u^ (u^.g)) / |u| with g = df/du^ (projection onto the tangent plane of the unit sphere).
template <typename Real> struct Vec3 {
    Real x, y, z;
};
template <typename Real> __device__ __forceinline__ Vec3<Real> v_sub(const Vec3<Real> a, const Vec3<Real> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename Real> __device__ __forceinline__ Vec3<Real> v_add(const Vec3<Real> a, const Vec3<Real> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename Real> __device__ __forceinline__ Vec3<Real> v_neg(const Vec3<Real> a) { return {-a.x, -a.y, -a.z}; }
template <typename Real> __device__ __forceinline__ Real v_dot(const Vec3<Real> a, const Vec3<Real> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename Real> __device__ __forceinline__ Vec3<Real> v_cross(const Vec3<Real> a, const Vec3<Real> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename Real> __device__ __forceinline__ Vec3<Real> v_load(const double *__restrict__ coords, const int a) {
    return {static_cast<Real>(coords[a * 3 + 0]), static_cast<Real>(coords[a * 3 + 1]), static_cast<Real>(coords[a * 3 + 2])};
}
// u -> (u^, 1/|u|)
template <typename Real> __device__ __forceinline__ Vec3<Real> v_unit(const Vec3<Real> u, Real &inv_norm) {
    inv_norm = 1 / tm_sqrt<Real>(v_dot(u, u));
    return {u.x * inv_norm, u.y * inv_norm, u.z * inv_norm};
}
// gradient w.r.t. the un-normalised vector, given g = df/d(unit vector)
template <typename Real> __device__ __forceinline__ Vec3<Real> v_unit_pullback(const Vec3<Real> uhat, const Real inv_norm, const Vec3<Real> g) {
    const Real along = v_dot(uhat, g);
    return {(g.x - uhat.x * along) * inv_norm, (g.y - uhat.y * along) * inv_norm, (g.z - uhat.z * along) * inv_norm};
}
template <typename Real> __device__ __forceinline__ void v_atomic_add_scaled(u64 *__restrict__ du_dx, const ForceLayout fl, const int a, const Vec3<Real> g, const Real scale) {
    force_add(du_dx, fl, a, 0, float_to_fixed<Real>(g.x * scale));
    force_add(du_dx, fl, a, 1, float_to_fixed<Real>(g.y * scale));
    force_add(du_dx, fl, a, 2, float_to_fixed<Real>(g.z * scale));
}

// centre c with neighbours 1, 2, 3: vol = (a^ x b^) . c^ with a = x1 - xc, b = x2 - xc, c = x3 - xc; penalised when vol > 0
template <typename Real>
__device__ __forceinline__ i128 chiral_atom_term(
    const int r, const double *__restrict__ coords, const double *__restrict__ params, const int *__restrict__ idxs,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, const bool want_u,
    const ForceLayout fl = ForceLayout{3, 1}) {
    const int ic = idxs[r * 4 + 0], i1 = idxs[r * 4 + 1], i2 = idxs[r * 4 + 2], i3 = idxs[r * 4 + 3];
    const Vec3<Real> xc = v_load<Real>(coords, ic);
    Real na, nb, nc;
    const Vec3<Real> a = v_unit(v_sub(v_load<Real>(coords, i1), xc), na);
    const Vec3<Real> b = v_unit(v_sub(v_load<Real>(coords, i2), xc), nb);
    const Vec3<Real> c = v_unit(v_sub(v_load<Real>(coords, i3), xc), nc);
    const Vec3<Real> ab = v_cross(a, b);
    const Real vol = v_dot(ab, c);
    const Real k = static_cast<Real>(params[r]);
    i128 energy = 0;
    if (want_u && vol > 0) {
        energy = float_to_fixed_energy<Real>(k * vol * vol);
    }
    if (!(vol > 0)) {
        return energy;
    }
    // (the reference kernel also returns early when k == 0, which zeroes du/dk there; d(k vol^2)/dk = vol^2 regardless,
    // as its JAX definition has it)
    if (du_dx && k != 0) {
        // vol is the scalar triple product: d vol / d a^ = b^ x c^, and cyclically
        const Vec3<Real> ga = v_unit_pullback(a, na, v_cross(b, c));
        const Vec3<Real> gb = v_unit_pullback(b, nb, v_cross(c, a));
        const Vec3<Real> gc = v_unit_pullback(c, nc, ab);
        const Real pref = 2 * k * vol;
        v_atomic_add_scaled(du_dx, fl, ic, v_neg(v_add(v_add(ga, gb), gc)), pref);
        v_atomic_add_scaled(du_dx, fl, i1, ga, pref);
        v_atomic_add_scaled(du_dx, fl, i2, gb, pref);
        v_atomic_add_scaled(du_dx, fl, i3, gc, pref);
    }
    if (du_dp) {
        atomicAdd(du_dp + r, float_to_fixed<Real>(vol * vol));
    }
    return energy;
}

// bond j-k with substituents i, l: vol = (a^ x b^) . (b^ x c^), a = xj - xi, b = xj - xk, c = xl - xk; penalised when
// sign * vol > 0
template <typename Real>
__device__ __forceinline__ i128 chiral_bond_term(
    const int r, const double *__restrict__ coords, const double *__restrict__ params, const int *__restrict__ idxs,
    const int *__restrict__ signs, u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, const bool want_u,
    const ForceLayout fl = ForceLayout{3, 1}) {
    const int ii = idxs[r * 4 + 0], ij = idxs[r * 4 + 1], ik = idxs[r * 4 + 2], il = idxs[r * 4 + 3];
    const Vec3<Real> xi = v_load<Real>(coords, ii), xj = v_load<Real>(coords, ij);
    const Vec3<Real> xk = v_load<Real>(coords, ik), xl = v_load<Real>(coords, il);
    Real na, nb, nc;
    const Vec3<Real> a = v_unit(v_sub(xj, xi), na);
    const Vec3<Real> b = v_unit(v_sub(xj, xk), nb);
    const Vec3<Real> c = v_unit(v_sub(xl, xk), nc);
    const Vec3<Real> n1 = v_cross(a, b);
    const Vec3<Real> n2 = v_cross(b, c);
    const Real vol = v_dot(n1, n2);
    const Real k = static_cast<Real>(params[r]);
    const Real signed_vol = static_cast<Real>(signs[r]) * vol;
    i128 energy = 0;
    if (want_u && signed_vol > 0) {
        energy = float_to_fixed_energy<Real>(k * vol * vol);
    }
    if (!(signed_vol > 0)) {
        return energy;
    }
    if (du_dx && k != 0) {
        // (a x b).n2 = a.(b x n2);  n1.(b x c) = c.(n1 x b);  and b enters both factors: n2 x a + c x n1
        const Vec3<Real> ga = v_unit_pullback(a, na, v_cross(b, n2));
        const Vec3<Real> gb = v_unit_pullback(b, nb, v_add(v_cross(n2, a), v_cross(c, n1)));
        const Vec3<Real> gc = v_unit_pullback(c, nc, v_cross(n1, b));
        const Real pref = 2 * k * vol;
        v_atomic_add_scaled(du_dx, fl, ii, v_neg(ga), pref);
        v_atomic_add_scaled(du_dx, fl, ij, v_add(ga, gb), pref);
        v_atomic_add_scaled(du_dx, fl, ik, v_neg(v_add(gb, gc)), pref);
        v_atomic_add_scaled(du_dx, fl, il, gc, pref);
    }
    if (du_dp) {
        atomicAdd(du_dp + r, float_to_fixed<Real>(vol * vol));
    }
    return energy;
}

template <typename Real>
__global__ __launch_bounds__(256) void k_chiral_atom_restraint(
    const int R, const double *__restrict__ coords, const double *__restrict__ params, const int *__restrict__ idxs,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, i128 *__restrict__ u_partials) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    i128 energy = 0;
    if (r < R) {
        energy = chiral_atom_term<Real>(r, coords, params, idxs, du_dx, du_dp, u_partials != nullptr);
    }
    if (u_partials) {
        store_wave_energy<Real>(energy, u_partials);
    }
}

template <typename Real>
__global__ __launch_bounds__(256) void k_chiral_bond_restraint(
    const int R, const double *__restrict__ coords, const double *__restrict__ params, const int *__restrict__ idxs,
    const int *__restrict__ signs, u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, i128 *__restrict__ u_partials) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    i128 energy = 0;
    if (r < R) {
        energy = chiral_bond_term<Real>(r, coords, params, idxs, signs, du_dx, du_dp, u_partials != nullptr);
    }
    if (u_partials) {
        store_wave_energy<Real>(energy, u_partials);
    }
}

// ---- flat-bottom restraints -------------------------------------------------------------------------------------
// reference: cpp/src/kernels/k_flat_bottom_bond.cuh:8-171, k_log_flat_bottom_bond.cuh:9-120; JAX: bonded.py:219-253
//   U_fb(r) = k/4 (r - rmax)^4 for r > rmax, k/4 (r - rmin)^4 for r < rmin, else 0   (minimum-image distance)
//   U_log   = -log(1 - exp(-beta U_fb)) / beta
template <typename Real, bool LOG>
__device__ __forceinline__ i128 flat_bottom_bond_term(
    const int b, const double *__restrict__ coords, const double *__restrict__ box, const double *__restrict__ params,
    const int *__restrict__ bond_idxs, const double beta_d, u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, const bool want_u,
    const ForceLayout fl = ForceLayout{3, 1}) {
    const int src = bond_idxs[b * 2 + 0], dst = bond_idxs[b * 2 + 1];
    const Real k = static_cast<Real>(params[b * 3 + 0]);
    const Real rmin = static_cast<Real>(params[b * 3 + 1]);
    const Real rmax = static_cast<Real>(params[b * 3 + 2]);
    Real dx[3];
    Real r2 = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        // displacement and re-imaging in double, like the reference kernel (k_flat_bottom_bond.cuh:112-117)
        double delta = coords[src * 3 + d] - coords[dst * 3 + d];
        delta -= box[d * 4] * nearbyint(delta / box[d * 4]);
        dx[d] = static_cast<Real>(delta);
        r2 += dx[d] * dx[d];
    }
    const Real r = tm_sqrt<Real>(r2);
    const Real above = r > rmax ? static_cast<Real>(1) : static_cast<Real>(0);
    const Real below = r < rmin ? static_cast<Real>(1) : static_cast<Real>(0);
    const Real dlo = r - rmin, dhi = r - rmax;
    const Real dlo3 = dlo * dlo * dlo, dhi3 = dhi * dhi * dhi;
    const Real quartic = above * (dhi3 * dhi) + below * (dlo3 * dlo);
    const Real nrg = (k / 4) * quartic;
    Real chain = 1; // dU/dU_fb
    i128 energy = 0;
    if constexpr (LOG) {
        const Real beta = static_cast<Real>(beta_d);
        const Real e = exp(-beta * nrg);
        // inside the flat region (U_fb = 0, or so small that exp rounds to 1) the log restraint's derivative is -inf * 0:
        // the reference's f32 kernel turns that NaN into a zero force in its float -> fixed conversion (PTX cvt of NaN
        // is 0; k_fixed_point.cuh:10-24) -- local MD relies on it for frozen atoms inside the radius.  Said explicitly here.
        chain = (1 - e) > 0 ? -e / (1 - e) : static_cast<Real>(0);
        if (want_u) {
            const Real x = beta * nrg; // -log(1 - exp(-x)), evaluated stably on both sides of log 2
            const Real l = x < static_cast<Real>(0.693147180559945309417232121) ? log(-expm1(-x)) : log1p(-exp(-x));
            energy = float_to_fixed_energy<Real>(-l / beta);
        }
    } else {
        if (want_u) {
            energy = float_to_fixed_energy<Real>(nrg);
        }
    }
    if (du_dp) {
        atomicAdd(du_dp + b * 3 + 0, float_to_fixed<Real>(chain * (quartic / 4)));
        atomicAdd(du_dp + b * 3 + 1, float_to_fixed<Real>(chain * (below * (-k * dlo3))));
        atomicAdd(du_dp + b * 3 + 2, float_to_fixed<Real>(chain * (above * (-k * dhi3))));
    }
    if (du_dx) {
        const Real du_dr = k * (above * dhi3 + below * dlo3);
        const Real inv_r = 1 / r;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const Real g = chain * (du_dr * dx[d] * inv_r);
            force_add(du_dx, fl, src, d, float_to_fixed<Real>(g));
            force_add(du_dx, fl, dst, d, float_to_fixed<Real>(-g));
        }
    }
    return energy;
}

template <typename Real, bool LOG>
__global__ __launch_bounds__(256) void k_flat_bottom_bond(
    const int B, const double *__restrict__ coords, const double *__restrict__ box, const double *__restrict__ params,
    const int *__restrict__ bond_idxs, const double beta, u64 *__restrict__ du_dx, u64 *__restrict__ du_dp,
    i128 *__restrict__ u_partials) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    i128 energy = 0;
    if (b < B) {
        energy = flat_bottom_bond_term<Real, LOG>(b, coords, box, params, bond_idxs, beta, du_dx, du_dp, u_partials != nullptr);
    }
    if (u_partials) {
        store_wave_energy<Real>(energy, u_partials);
    }
}

// ---- centroid restraint --------------------------------------------------------------------------------------
// reference: cpp/src/kernels/k_centroid_restraint.cuh:8-86; JAX: bonded.py:8-31.  U = kb (|<x_A> - <x_B>| - b0)^2
// (geometric centroids, no periodic imaging, no parameters).  Pass 1 sums the coordinates of both groups in fixed
// point (deterministic), pass 2 hands every group atom its share of the gradient.
template <typename Real>
__global__ __launch_bounds__(256) void k_centroid_sums(
    const int NA, const int NB, const double *__restrict__ coords, const int *__restrict__ a_idxs, const int *__restrict__ b_idxs,
    u64 *__restrict__ sums) { // [2][3], zeroed by the caller
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= NA + NB) {
        return;
    }
    const bool in_a = t < NA;
    const int atom = in_a ? a_idxs[t] : b_idxs[t - NA];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        atomicAdd(sums + (in_a ? 0 : 3) + d, float_to_fixed<Real>(static_cast<Real>(coords[atom * 3 + d])));
    }
}

template <typename Real>
__global__ __launch_bounds__(256) void k_centroid_restraint(
    const int NA, const int NB, const int *__restrict__ a_idxs, const int *__restrict__ b_idxs, const u64 *__restrict__ sums,
    const double kb_d, const double b0_d, u64 *__restrict__ du_dx, i128 *__restrict__ u) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= NA + NB) {
        return;
    }
    const Real kb = static_cast<Real>(kb_d), b0 = static_cast<Real>(b0_d);
    Real delta[3];
    Real d2 = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        delta[d] = fixed_to_float<Real>(sums[d]) / NA - fixed_to_float<Real>(sums[3 + d]) / NB;
        d2 += delta[d] * delta[d];
    }
    const Real dij = tm_sqrt<Real>(d2);
    if (t == 0 && u) {
        u[0] = float_to_fixed_energy<Real>(kb * (dij - b0) * (dij - b0));
    }
    if (du_dx) {
        const bool in_a = t < NA;
        const int atom = in_a ? a_idxs[t] : b_idxs[t - NA];
        const Real share = (in_a ? static_cast<Real>(1) : static_cast<Real>(-1)) / static_cast<Real>(in_a ? NA : NB);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            // b0 == 0: the gradient 2 kb delta is well defined at dij == 0 as well (bonded.py:26-31)
            const Real g = b0 != 0 ? 2 * kb * (dij - b0) * (delta[d] / dij) : 2 * kb * delta[d];
            force_add(du_dx, ForceLayout{3, 1}, atom, d, float_to_fixed<Real>(share * g));
        }
    }
}

} // namespace tmamd
