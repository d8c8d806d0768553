// LangevinIntegrator (BAOAB) and the MD Context loop for gfx950.
// reference: cpp/src/langevin_integrator.cu:14-88, cpp/src/kernels/k_integrator.cuh:5-62, cpp/src/context.cu:28-303,
//            timemachine/integrator.py:124-150 (python reference of the same step)
//
// Differences on purpose:
//  * noise is generated INSIDE the update kernel with a counter-based Philox4x32-10 keyed on (seed; atom, step):
//    no noise buffer, no separate RNG launch, no 12 B/atom/step of HBM traffic.  cuRAND's XORWOW stream cannot be
//    reproduced by any independent implementation, so trajectories are comparable exactly at friction == 0
//    (ccs == 0) and statistically otherwise -- same contract as the reference's own tests (tests/test_md.py:142-247).
//  * BOLTZ is the C++ value (cpp/src/constants.hpp:5), not the python one (timemachine/constants.py:5-8).
#include "engine.hpp"
#include "fixed_point.hip.hpp"
#include "nb_snapshot_test.hip.hpp"
#include "philox.hip.hpp"
#include "profiler.hpp"

#include <algorithm>
#include <exception>
#include <thread>
#include <chrono>
#include <cstdio>
#include <iterator>
#include <map>
#include <mutex>
#include <cmath>
#include <iostream>

namespace tmamd {

static const double BOLTZ = 0.008314462618; // kJ/mol/K, cpp/src/constants.hpp:5

// three N(0,1) variates for (atom, step) by Box-Muller on Philox output, in the integrator's own precision (the
// reference draws RealType normals: curandGenerateNormal / curandGenerateNormalDouble, langevin_integrator.cu:74-79).
// float: the radial uniforms use all 32 random bits, (r + 1/2) 2^-32 in (0, 1]: exact for the small values that make the
// tails, which reach sqrt(-2 ln 2^-33) = 6.8 sigma.  double: 53-bit uniforms from a second Philox block; tails to 8.6 sigma.
__device__ __forceinline__ void normal3(unsigned long long seed, unsigned long long step, unsigned int atom, float n[3]) {
    unsigned int r[4];
    philox4x32_10(atom, static_cast<unsigned int>(step), static_cast<unsigned int>(step >> 32), 0x54494d45u,
                  static_cast<unsigned int>(seed), static_cast<unsigned int>(seed >> 32), r);
    const float two_pow_m32 = 2.3283064365386963e-10f;
    const float u0 = fminf((static_cast<float>(r[0]) + 0.5f) * two_pow_m32, 1.0f); // (0, 1]: never 0
    const float u1 = static_cast<float>(r[1]) * two_pow_m32;
    const float u2 = fminf((static_cast<float>(r[2]) + 0.5f) * two_pow_m32, 1.0f);
    const float u3 = static_cast<float>(r[3]) * two_pow_m32;
    const float two_pi = 6.2831853071795864769f;
    const float ra = sqrtf(-2.0f * logf(u0));
    const float rb = sqrtf(-2.0f * logf(u2));
    float s, c;
    sincosf(two_pi * u1, &s, &c);
    n[0] = ra * c;
    n[1] = ra * s;
    n[2] = rb * cosf(two_pi * u3);
}
__device__ __forceinline__ void normal3(unsigned long long seed, unsigned long long step, unsigned int atom, double n[3]) {
    unsigned int r[4], q[4];
    philox4x32_10(atom, static_cast<unsigned int>(step), static_cast<unsigned int>(step >> 32), 0x54494d45u,
                  static_cast<unsigned int>(seed), static_cast<unsigned int>(seed >> 32), r);
    philox4x32_10(atom, static_cast<unsigned int>(step), static_cast<unsigned int>(step >> 32), 0x54494d46u,
                  static_cast<unsigned int>(seed), static_cast<unsigned int>(seed >> 32), q);
    const double two_pow_m53 = 1.1102230246251565e-16;
    auto u53 = [&](unsigned int hi, unsigned int lo) { // (0, 1)
        const unsigned long long bits = (static_cast<unsigned long long>(hi) << 21) | (lo >> 11);
        return (static_cast<double>(bits) + 0.5) * two_pow_m53;
    };
    const double u0 = u53(r[0], q[0]), u1 = u53(r[1], q[1]), u2 = u53(r[2], q[2]), u3 = u53(r[3], q[3]);
    const double two_pi = 6.2831853071795864769;
    const double ra = sqrt(-2.0 * log(u0));
    const double rb = sqrt(-2.0 * log(u2));
    double s, c;
    sincos(two_pi * u1, &s, &c);
    n[0] = ra * c;
    n[1] = ra * s;
    n[2] = rb * cos(two_pi * u3);
}

// Leaves a force producer's next gather done (PregatherTarget): new position into its sorted record, rebuild test against
// its snapshot with the arithmetic of k_check_gather (producer precision), consumed accumulator slot zeroed.
template <typename GReal>
__device__ __forceinline__ void pregather_atom_as(const PregatherTarget &t, const int slot, const int atom, const double xn, const double yn, const double zn) {
    GReal *g = static_cast<GReal *>(t.gathered) + static_cast<size_t>(slot) * 8;
    const GReal gx = static_cast<GReal>(xn), gy = static_cast<GReal>(yn), gz = static_cast<GReal>(zn);
    g[0] = gx;
    g[1] = gy;
    g[2] = gz;
    if (t.second_records != 0) { // (wave-uniform) a merged producer: the atom's second record follows (holes' and guest rows' are never read)
        GReal *g2 = static_cast<GReal *>(t.gathered) + (static_cast<size_t>(t.second_records) + slot) * 8;
        g2[0] = gx;
        g2[1] = gy;
        g2[2] = gz;
    }
    bool rebuild;
    if (t.snap_box != nullptr) { // wave-uniform: a barostat works on the producer
        rebuild = snapshot_calls_for_rebuild(xn, yn, zn, t.snap_x + atom * 3, t.cur_box, t.snap_box, t.pad2_quarter);
    } else {
        const GReal dx = static_cast<GReal>(t.snap_x[atom * 3 + 0]) - gx;
        const GReal dy = static_cast<GReal>(t.snap_x[atom * 3 + 1]) - gy;
        const GReal dz = static_cast<GReal>(t.snap_x[atom * 3 + 2]) - gz;
        const GReal d2 = dx * dx + dy * dy + dz * dz;
        rebuild = static_cast<double>(d2) > t.pad2_quarter;
    }
    if (rebuild) {
        if (t.nbl_counters == nullptr) {
            *t.flag_set = 1; // benign race: every writer stores the same value
        } else if (atomicExch(t.flag_set, 1) == 0) {
            // sorted hand-over: the FIRST atom to raise the flag also resets the list counters the coming build accumulates into
            // (what the producer's bounds kernel does on the other paths; kernels_nblist.hip.hpp) -- counters[3], the build
            // count, stays.  On a rebuild step hundreds of atoms get here; one of them writes.
            t.nbl_counters[0] = 0;
            t.nbl_counters[1] = 0;
            t.nbl_counters[2] = 0;
            for (int k = NB_COUNTER_CLASS0; k < NB_NUM_COUNTERS; k++) {
                t.nbl_counters[k] = 0;
            }
        }
    }
    t.g_du_dx[0 * static_cast<size_t>(t.stride) + slot] = 0;
    t.g_du_dx[1 * static_cast<size_t>(t.stride) + slot] = 0;
    t.g_du_dx[2 * static_cast<size_t>(t.stride) + slot] = 0;
}

__device__ __forceinline__ void pregather_atom(const PregatherTarget &t, const int slot, const int atom, const double xn, const double yn, const double zn) {
    if (t.gathered == nullptr || slot < 0) {
        return;
    }
    if (t.real_bytes == 8) {
        pregather_atom_as<double>(t, slot, atom, xn, yn, zn);
    } else {
        pregather_atom_as<float>(t, slot, atom, xn, yn, zn);
    }
}

// reference: k_update_forward_baoab (k_integrator.cuh:5-62).  x, v stored f64; arithmetic in Real with the same
// promotion points: v_mid = Real(v + cb*F); v' = ca*v_mid + cc*noise (Real); x += Real(0.5*dt) * (v_mid + v') in f64.
template <typename Real>
__global__ __launch_bounds__(256) void k_update_forward_baoab(
    const int N, const Real ca, const unsigned int *__restrict__ idxs, const Real *__restrict__ cbs, const Real *__restrict__ ccs,
    const unsigned long long seed, const unsigned long long step, double *__restrict__ x_t, double *__restrict__ v_t,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dx_cm, const int cm_stride, // [N, 3] and component-major force accumulators
    const Real dt,
    // up to two force contributions picked up from their producers' sorted accumulators (DeferredForces); nullptr = none
    // (component-major: component d of slot s at g[d * stride + s])
    const u64 *__restrict__ g0, const int *__restrict__ slot0, const int stride0, const u64 *__restrict__ g1,
    const int *__restrict__ slot1, const int stride1,
    // where to leave the producers' next gather (gathered == nullptr: not wanted)
    const PregatherTarget pg0, const PregatherTarget pg1,
    // the integrator's progress word (pinned host memory; Integrator::progress_word) and what to leave in it
    unsigned int *__restrict__ progress, const unsigned int progress_value) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (pg0.gathered) {
            *pg0.flag_clear = 0; // the flag of the call just consumed becomes the one after next's
        }
        if (pg1.gathered) {
            *pg1.flag_clear = 0;
        }
        __hip_atomic_store(progress, progress_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int kidx = blockIdx.x * blockDim.x + threadIdx.x; kidx < N; kidx += gridDim.x * blockDim.x) {
        const int atom = idxs == nullptr ? kidx : static_cast<int>(idxs[kidx]);
        if (atom < N) {
            const Real cb = cbs[atom];
            const Real cc = ccs[atom];
            Real nz[3] = {0, 0, 0};
            if (cc != 0) {
                normal3(seed, step, static_cast<unsigned int>(atom), nz);
            }
            const Real half_dt = static_cast<Real>(0.5) * dt;
            const int s0 = g0 ? slot0[atom] : -1;
            const int s1 = g1 ? slot1[atom] : -1;
            double xn[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                // wrapping integer sum: same bits as a scatter-add of every contribution into one array would give
                u64 f = 0;
                if (du_dx_cm != nullptr) { // nullptr: nothing was added to it this step (ForcePlan::cm_written)
                    f = du_dx_cm[static_cast<size_t>(d) * cm_stride + atom];
                    du_dx_cm[static_cast<size_t>(d) * cm_stride + atom] = 0;
                }
                if (du_dx != nullptr) { // nullptr: nothing was added to the [N, 3] array this step (it stays zero)
                    f += du_dx[atom * 3 + d];
                    du_dx[atom * 3 + d] = 0; // consumed: the next force evaluation accumulates from zero
                }
                if (s0 >= 0) {
                    f += g0[static_cast<size_t>(d) * stride0 + s0];
                }
                if (s1 >= 0) {
                    f += g1[static_cast<size_t>(d) * stride1 + s1];
                }
                const Real force = -fixed_to_float<Real>(f);
                const Real v_mid = static_cast<Real>(v_t[atom * 3 + d] + static_cast<double>(cb * force));
                const Real v_new = ca * v_mid + cc * nz[d];
                v_t[atom * 3 + d] = static_cast<double>(v_new);
                xn[d] = x_t[atom * 3 + d] + static_cast<double>(half_dt) * (static_cast<double>(v_mid) + static_cast<double>(v_new));
                x_t[atom * 3 + d] = xn[d];
            }
            pregather_atom(pg0, s0, atom, xn[0], xn[1], xn[2]);
            pregather_atom(pg1, s1, atom, xn[0], xn[1], xn[2]);
        } else if (idxs != nullptr) {
#pragma unroll
            for (int d = 0; d < 3; d++) {
                if (du_dx != nullptr) {
                    du_dx[kidx * 3 + d] = 0;
                }
                if (du_dx_cm != nullptr) {
                    du_dx_cm[static_cast<size_t>(d) * cm_stride + kidx] = 0;
                }
            }
        }
    }
}

// The same update, walking the producer's SORTED slots instead of the atoms (PregatherTarget's sorted hand-over: one
// deferred producer whose order covers every atom).  Thread t owns slot t: atom = perm[t]; the producer's accumulator is
// read and its record written coalesced, no slot_of_atom indirection; and since a 64-lane wave now holds two whole
// 32-atom blocks of the producer's order, their bounding boxes (the input of a neighbor-list build) fall out of five
// shuffle steps per dimension, every step -- the producer's bounds kernel is not launched on MD steps at all.
// Same arithmetic, same Philox counters (keyed by atom) as the atom-order kernel: trajectories are bit-identical.
// FROM_CACHE: x, v, cb, cc come from the integrator's slot-ordered copies (xs, vs, cbs_s, ccs_s; see LangevinIntegrator) instead
// of the atom-order arrays behind perm: every load of the update is then one coalesced hop; perm is still read, in parallel,
// for the noise key and the scattered stores.  The copies are (re)written on every launch.
template <typename Real, typename GReal, bool FROM_CACHE>
__global__ __launch_bounds__(64) void k_update_forward_baoab_sorted(
    const int N, const Real ca, const Real *__restrict__ cbs, const Real *__restrict__ ccs, const unsigned long long seed,
    const unsigned long long step, double *__restrict__ x_t, double *__restrict__ v_t, u64 *__restrict__ du_dx,
    u64 *__restrict__ du_dx_cm, const int cm_stride, const Real dt, const u64 *__restrict__ g0, const int stride0,
    const double *__restrict__ box, const PregatherTarget pg, double *__restrict__ xs, double *__restrict__ vs,
    Real *__restrict__ cbs_s, Real *__restrict__ ccs_s, unsigned int *__restrict__ progress, const unsigned int progress_value) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *pg.flag_clear = 0; // the flag of the call just consumed becomes the one after next's
        __hip_atomic_store(progress, progress_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (as in k_update_forward_baoab)
    }
    const int slot = blockIdx.x * 64 + threadIdx.x;
    const int lane = threadIdx.x;
    // (N counts the SLOTS of the producer's order; a merged order pads its guest rows to a block boundary with holes: perm == 0xffffffff)
    const int atom = slot < N ? static_cast<int>(pg.perm[slot]) : -1;
    const bool valid = atom >= 0;
    GReal p[3] = {0, 0, 0}; // the new position as the producer's record stores it
    if (valid) {
        Real cb, cc;
        double xo[3], vo[3];
        if constexpr (FROM_CACHE) {
            cb = cbs_s[slot];
            cc = ccs_s[slot];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                xo[d] = xs[slot * 3 + d];
                vo[d] = vs[slot * 3 + d];
            }
        } else {
            cb = cbs[atom];
            cc = ccs[atom];
            cbs_s[slot] = cb;
            ccs_s[slot] = cc;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                xo[d] = x_t[atom * 3 + d];
                vo[d] = v_t[atom * 3 + d];
            }
        }
        Real nz[3] = {0, 0, 0};
        if (cc != 0) {
            normal3(seed, step, static_cast<unsigned int>(atom), nz);
        }
        const Real half_dt = static_cast<Real>(0.5) * dt;
        double xn[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            u64 f = g0[static_cast<size_t>(d) * stride0 + slot];
            if (du_dx_cm != nullptr) { // nullptr: nothing was added to it this step (ForcePlan::cm_written)
                f += du_dx_cm[static_cast<size_t>(d) * cm_stride + atom];
                du_dx_cm[static_cast<size_t>(d) * cm_stride + atom] = 0;
            }
            if (du_dx != nullptr) {
                f += du_dx[atom * 3 + d];
                du_dx[atom * 3 + d] = 0;
            }
            const Real force = -fixed_to_float<Real>(f);
            const Real v_mid = static_cast<Real>(vo[d] + static_cast<double>(cb * force));
            const Real v_new = ca * v_mid + cc * nz[d];
            v_t[atom * 3 + d] = static_cast<double>(v_new);
            vs[slot * 3 + d] = static_cast<double>(v_new);
            xn[d] = xo[d] + static_cast<double>(half_dt) * (static_cast<double>(v_mid) + static_cast<double>(v_new));
            x_t[atom * 3 + d] = xn[d];
            xs[slot * 3 + d] = xn[d];
            p[d] = static_cast<GReal>(xn[d]);
        }
        pregather_atom_as<GReal>(pg, slot, atom, xn[0], xn[1], xn[2]);
    }
    // bounding boxes of the two 32-slot blocks of this wave: every atom imaged next to the block's first one, butterfly
    // min / max (the arithmetic of k_block_bounds<Real, false>, kernels_nblist.hip.hpp)
    const GReal half = static_cast<GReal>(0.5);
    const int first = lane & 32;
    GReal lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const GReal b = static_cast<GReal>(box[d * 4]);
        const GReal ib = 1 / b;
        const GReal p0 = __shfl(p[d], first, 64);
        const GReal img = valid ? p[d] - b * nearbyint((p[d] - p0) * ib) : p0;
        GReal l = img, h = img;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            l = min(l, __shfl_xor(l, o, 64));
            h = max(h, __shfl_xor(h, o, 64));
        }
        lo[d] = l;
        hi[d] = h;
    }
    const int sub = lane & 31;
    if (sub < 3 && (slot - sub) < N) { // lane `first + d` writes component d of its block
        const int blk = slot >> 5;
        const GReal l = sub == 0 ? lo[0] : (sub == 1 ? lo[1] : lo[2]);
        const GReal h = sub == 0 ? hi[0] : (sub == 1 ? hi[1] : hi[2]);
        static_cast<GReal *>(pg.blk_ctr)[blk * 3 + sub] = half * (h + l);
        static_cast<GReal *>(pg.blk_ext)[blk * 3 + sub] = half * (h - l);
    }
}

template <typename Real>
LangevinIntegrator<Real>::LangevinIntegrator(
    const int N, const double *masses, const double temperature, const double dt, const double friction, const int seed)
    : N_(N), temperature_(temperature), dt_(static_cast<Real>(dt)), friction_(friction), seed_(static_cast<unsigned long long>(static_cast<long long>(seed))),
      step_(0), d_cbs_(N), d_ccs_(N), d_xs_(static_cast<size_t>(N + TILE) * 3), d_vs_(static_cast<size_t>(N + TILE) * 3), d_cbs_s_(N + TILE), d_ccs_s_(N + TILE), d_du_dx_(static_cast<size_t>(N) * 3), cm_stride_((N + 7) & ~7), d_du_dx_cm_(static_cast<size_t>((N + 7) & ~7) * 3) {
    ca_ = static_cast<Real>(std::exp(-friction * dt));
    const double kT = BOLTZ * temperature;
    const double ccs_adjustment = std::sqrt(1 - std::exp(-2 * friction * dt));
    std::vector<Real> h_cbs(N_), h_ccs(N_);
    for (int i = 0; i < N_; i++) {
        h_cbs[i] = static_cast<Real>(dt_ / masses[i]);
        h_ccs[i] = static_cast<Real>(ccs_adjustment * std::sqrt(kT / masses[i]));
    }
    d_cbs_.copy_from(h_cbs.data());
    d_ccs_.copy_from(h_ccs.data());
    HIP_CHECK(hipMemset(d_du_dx_.data, 0, d_du_dx_.size()));
    HIP_CHECK(hipMemset(d_du_dx_cm_.data, 0, d_du_dx_cm_.size()));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h_progress_), 64, hipHostMallocDefault)); // a cache line of its own
    *h_progress_ = 0;
}

template <typename Real> LangevinIntegrator<Real>::~LangevinIntegrator() {
    if (h_progress_ != nullptr) {
        (void)hipHostFree(h_progress_);
    }
}

template <typename Real>
void LangevinIntegrator<Real>::step_fwd(
    std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs,
    hipStream_t stream) {
    // forces only: every bound potential describes itself to one plan, so short per-term kernels share a launch
    plan_.clear();
    for (auto &bp : bps) {
        bp->potential->plan_forces(N_, bp->size, bp->size > 0 ? bp->d_p.data : nullptr, plan_);
    }
    deferred_.clear();
    const bool wrote_du_dx = plan_.run(N_, d_x_t, d_box_t, d_du_dx_.data, stream, &deferred_, 2, d_du_dx_cm_.data, cm_stride_);
    u64 *cm = plan_.cm_written() ? d_du_dx_cm_.data : nullptr; // untouched this step: the update kernels skip it
    const DeferredForces none;
    const DeferredForces &df0 = deferred_.size() > 0 ? deferred_[0] : none;
    const DeferredForces &df1 = deferred_.size() > 1 ? deferred_[1] : none;
#ifndef TM_BAOAB_TPB
#define TM_BAOAB_TPB 64 // 368 small workgroups reach every CU: the kernel is a latency chain per atom, not a throughput problem
#endif
    const int tpb = TM_BAOAB_TPB;
    // every atom moves (no local-MD index list): the update kernel can leave the producers' next gather done
    const PregatherTarget no_target;
    const bool pregather = d_idxs == nullptr;
    // one producer whose sorted order covers every atom: walk its slots (and leave its block bounds done as well)
    const bool sorted = pregather && deferred_.size() == 1 && df0.next.gathered != nullptr && df0.next.sorted_n > 0 && df0.next.covers_atoms == N_ && df0.next.perm != nullptr;
    const int prof = Profiler::get().begin("integrator_update", stream);
    enqueued_++; // what the update kernel of this step leaves in the progress word
    if (sorted) {
        u64 *dx = wrote_du_dx ? d_du_dx_.data : nullptr;
        // the slot-ordered copies of x, v, cb, cc are current iff the last launch here wrote them for this producer and these
        // arrays, the producer has just consumed the hand-over that launch left (same inputs, same order), and nobody has
        // touched x / v since (invalidate_state_cache)
        const bool from_cache = state_cache_valid_ && df0.consumed_sorted_pregather && cache_owner_ == df0.owner && cache_x_ == d_x_t && cache_v_ == d_v_t;
#define TM_LAUNCH_SORTED(GREAL, CACHE)                                                                                 \
    k_update_forward_baoab_sorted<Real, GREAL, CACHE><<<ceil_divide(df0.next.sorted_n, 64), 64, 0, stream>>>(            \
        df0.next.sorted_n, ca_, d_cbs_.data, d_ccs_.data, seed_, step_, d_x_t, d_v_t, dx, cm, cm_stride_, dt_, df0.g_du_dx, df0.stride, d_box_t, df0.next, \
        d_xs_.data, d_vs_.data, d_cbs_s_.data, d_ccs_s_.data, h_progress_, enqueued_)
        if (df0.next.real_bytes == 8) {
            if (from_cache) {
                TM_LAUNCH_SORTED(double, true);
            } else {
                TM_LAUNCH_SORTED(double, false);
            }
        } else {
            if (from_cache) {
                TM_LAUNCH_SORTED(float, true);
            } else {
                TM_LAUNCH_SORTED(float, false);
            }
        }
#undef TM_LAUNCH_SORTED
        state_cache_valid_ = true;
        cache_owner_ = df0.owner;
        cache_x_ = d_x_t;
        cache_v_ = d_v_t;
    } else {
        PregatherTarget t0 = pregather ? df0.next : no_target, t1 = pregather ? df1.next : no_target;
        t0.nbl_counters = nullptr; // the atom-order kernel leaves no block bounds: the producer's own bounds kernel resets them
        t1.nbl_counters = nullptr;
        state_cache_valid_ = false; // this kernel does not maintain the slot-ordered copies
        k_update_forward_baoab<Real><<<ceil_divide(N_, tpb), tpb, 0, stream>>>(
            N_, ca_, d_idxs, d_cbs_.data, d_ccs_.data, seed_, step_, d_x_t, d_v_t, wrote_du_dx ? d_du_dx_.data : nullptr, cm, cm_stride_, dt_,
            df0.g_du_dx, df0.slot_of_atom, df0.stride, df1.g_du_dx, df1.slot_of_atom, df1.stride, t0, t1, h_progress_, enqueued_);
    }
    Profiler::get().end("integrator_update", prof, stream);
    HIP_CHECK(hipGetLastError());
    if (pregather) {
        for (const DeferredForces &df : deferred_) {
            if (df.owner != nullptr && df.next.gathered != nullptr) {
                df.owner->pregather_committed(d_x_t, d_box_t, sorted);
            }
        }
    }
    step_++;
}

// ------------------------------------------------------------------------------------------------------------
// reference: update_forward_velocity_verlet / half_step_velocity_verlet (k_integrator.cuh:64-130)
template <int MODE>
__global__ __launch_bounds__(256) void k_velocity_verlet(
    const int N, const unsigned int *__restrict__ idxs, const double *__restrict__ cbs, double *__restrict__ x_t, double *__restrict__ v_t,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dx_cm, const int cm_stride, const double dt, const u64 *__restrict__ g0,
    const int *__restrict__ slot0, const int stride0,
    const u64 *__restrict__ g1, const int *__restrict__ slot1, const int stride1) {
    const int kidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (kidx >= N) {
        return;
    }
    const int atom = idxs == nullptr ? kidx : static_cast<int>(idxs[kidx]);
    if (atom >= N) {
        if (idxs != nullptr) { // frozen slot: its accumulators still have to start the next evaluation from zero
#pragma unroll
            for (int d = 0; d < 3; d++) {
                du_dx[kidx * 3 + d] = 0;
                du_dx_cm[static_cast<size_t>(d) * cm_stride + kidx] = 0;
            }
        }
        return;
    }
    const double cb = MODE == 0 ? cbs[atom] : 0.5 * cbs[atom];
    const int s0 = g0 ? slot0[atom] : -1;
    const int s1 = g1 ? slot1[atom] : -1;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        u64 f = du_dx[atom * 3 + d] + du_dx_cm[static_cast<size_t>(d) * cm_stride + atom];
        if (s0 >= 0) {
            f += g0[static_cast<size_t>(d) * stride0 + s0];
        }
        if (s1 >= 0) {
            f += g1[static_cast<size_t>(d) * stride1 + s1];
        }
        const double force = fixed_to_float<double>(f);
        const double v = v_t[atom * 3 + d] + cb * force;
        v_t[atom * 3 + d] = v;
        if (MODE != 2) {
            x_t[atom * 3 + d] += dt * v;
        }
        du_dx[atom * 3 + d] = 0;
        du_dx_cm[static_cast<size_t>(d) * cm_stride + atom] = 0;
    }
}

VelocityVerletIntegrator::VelocityVerletIntegrator(const int N, const double dt, const double *h_cbs)
    : N_(N), dt_(dt), initialized_(false), d_cbs_(N), d_du_dx_(static_cast<size_t>(N) * 3), cm_stride_((N + 7) & ~7),
      d_du_dx_cm_(static_cast<size_t>((N + 7) & ~7) * 3) {
    d_cbs_.copy_from(h_cbs);
    HIP_CHECK(hipMemset(d_du_dx_.data, 0, d_du_dx_.size()));
    HIP_CHECK(hipMemset(d_du_dx_cm_.data, 0, d_du_dx_cm_.size()));
}

void VelocityVerletIntegrator::forces_then_update(
    const int mode, std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t,
    unsigned int *d_idxs, hipStream_t stream) {
    plan_.clear();
    for (auto &bp : bps) {
        bp->potential->plan_forces(N_, bp->size, bp->size > 0 ? bp->d_p.data : nullptr, plan_);
    }
    deferred_.clear();
    plan_.run(N_, d_x_t, d_box_t, d_du_dx_.data, stream, &deferred_, 2, d_du_dx_cm_.data, cm_stride_);
    const DeferredForces none;
    const DeferredForces &a = deferred_.size() > 0 ? deferred_[0] : none;
    const DeferredForces &b = deferred_.size() > 1 ? deferred_[1] : none;
    const int tpb = 256, blocks = ceil_divide(N_, tpb);
#define TM_VV(MODE)                                                                                                    \
    k_velocity_verlet<MODE><<<blocks, tpb, 0, stream>>>(                                                               \
        N_, d_idxs, d_cbs_.data, d_x_t, d_v_t, d_du_dx_.data, d_du_dx_cm_.data, cm_stride_, dt_, a.g_du_dx, a.slot_of_atom, a.stride, b.g_du_dx, b.slot_of_atom, b.stride)
    if (mode == 0) {
        TM_VV(0);
    } else if (mode == 1) {
        TM_VV(1);
    } else {
        TM_VV(2);
    }
#undef TM_VV
    HIP_CHECK(hipGetLastError());
}

void VelocityVerletIntegrator::step_fwd(
    std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs,
    hipStream_t stream) {
    this->forces_then_update(0, bps, d_x_t, d_v_t, d_box_t, d_idxs, stream);
}

void VelocityVerletIntegrator::initialize(
    std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs,
    hipStream_t stream) {
    if (initialized_) {
        throw std::runtime_error("initialized twice");
    }
    this->forces_then_update(1, bps, d_x_t, d_v_t, d_box_t, d_idxs, stream);
    initialized_ = true;
}

void VelocityVerletIntegrator::finalize(
    std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs,
    hipStream_t stream) {
    if (!initialized_) {
        throw std::runtime_error("not initialized");
    }
    this->forces_then_update(2, bps, d_x_t, d_v_t, d_box_t, d_idxs, stream);
    initialized_ = false;
}

template class LangevinIntegrator<float>;
template class LangevinIntegrator<double>;

// Waiting for a long MD call without burning a CPU: hipStreamSynchronize spins (the runtime's default wait policy), so a rank whose
// host thread has enqueued its 2 000 steps in 16 ms then keeps one CPU at 100 % for the 140 ms the device needs -- eight ranks: eight
// of the sixteen CPUs a GPU box grants.  Instead: an event behind the enqueued work, polled with short sleeps (the first polls come
// quickly so that short calls -- single steps, tests -- do not wait a sleep's length for nothing).  TM_AMD_SPIN_WAIT=1 restores the
// runtime's own wait.
static void wait_for_stream(hipStream_t stream) {
    static const bool spin = std::getenv("TM_AMD_SPIN_WAIT") != nullptr;
    if (spin) {
        HIP_CHECK(hipStreamSynchronize(stream));
        return;
    }
    // (one event per host thread AND device: an event belongs to the device that was current when it was made)
    static thread_local hipEvent_t ev = nullptr;
    static thread_local int ev_device = -1;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (ev == nullptr || ev_device != dev) {
        if (ev != nullptr) {
            (void)hipEventDestroy(ev);
        }
        HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ev_device = dev;
    }
    HIP_CHECK(hipEventRecord(ev, stream));
    for (int polls = 0;; polls++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) {
            return;
        }
        if (e != hipErrorNotReady) {
            HIP_CHECK(e);
        }
        if (polls >= 200) { // ~ the first 100-200 us are polled back to back; after that the call is a long one
            std::this_thread::sleep_for(std::chrono::microseconds(30));
        }
    }
}

// Bounded run-ahead.  The enqueueing host is several times faster than the device (10-12 us against 70 us per DHFR-sized step), and
// what the HIP runtime does with the lead costs CPU time: past a few thousand launches in flight every further launch SPINS inside
// the runtime until a slot frees up (a 2 000-step call kept the calling thread 0.15 busy, an 8 000-step call 0.6, every call after it
// 1.0), and well before that the runtime's own helper thread grows busier with the number of commands in flight (0.05 of a CPU at
// ~50 of them, 0.2-0.5 at several hundred: scripts/host_cpu_probe3.py).  So the stepping loops keep the host a SMALL number of
// steps ahead of the device -- a few milliseconds of work, far more than the device needs to stay busy:
//   * an integrator that keeps a progress word (Integrator::progress_word: the update kernel of every step leaves the count of
//     steps enqueued so far in pinned host memory) is throttled on that word: no runtime call, no extra packet in the stream; once
//     the host is more than RUN_AHEAD_HI steps ahead it sleeps until the lead is down to half of that.  The bound trades the
//     helper thread's load (0.20-0.26 CPUs busy per process at 32, 0.20-0.33 at 64, 0.22-0.45 at 128, 0.25-0.68 at 256) against the
//     hiccup of the host that the device rides out without going idle (32 steps: 2.2 ms of DHFR-sized work, 0.5 ms at 256 atoms);
//     TM_AMD_RUN_AHEAD_STEPS in the environment overrides it;
//   * any other integrator: an event every RUN_AHEAD_EVENT_STEPS steps, and a sleeping wait for the event before the last (each
//     event costs the device ~4 us, hence the larger spacing).
// The word says "this step's update kernel has started", which is all a throttle needs; completion is wait_for_stream's business.
struct RunAhead {
    static const int RUN_AHEAD_HI = 64, RUN_AHEAD_EVENT_STEPS = 32;
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool pending[2] = {false, false};
    int chunk = 0;
    ~RunAhead() {
        for (hipEvent_t e : ev) {
            if (e) {
                (void)hipEventDestroy(e);
            }
        }
    }
    static void nap() { std::this_thread::sleep_for(std::chrono::microseconds(50)); }
    void after_step(const int i, hipStream_t stream, const Integrator &intg) { // i: steps enqueued so far by this call on `stream`
        static const bool spin = std::getenv("TM_AMD_SPIN_WAIT") != nullptr;
        if (spin) {
            return;
        }
        if (const volatile unsigned int *word = intg.progress_word()) {
            static const int hi = std::getenv("TM_AMD_RUN_AHEAD_STEPS") ? std::max(2, std::atoi(std::getenv("TM_AMD_RUN_AHEAD_STEPS"))) : RUN_AHEAD_HI;
            const unsigned int enqueued = intg.progress_enqueued();
            if (static_cast<int>(enqueued - *word) <= hi) { // (wrapping difference: the counter is 32 bits wide)
                return;
            }
            unsigned int seen = *word;
            for (int naps = 0; static_cast<int>(enqueued - seen) > hi / 2; naps++) {
                nap();
                const unsigned int now = *word;
                if (now != seen) {
                    seen = now;
                    naps = 0;
                } else if (naps >= 20000) { // ~2 s without a step: a device error surfaces here; an idle stream means the word is not coming
                    const hipError_t e = hipStreamQuery(stream);
                    if (e == hipSuccess) {
                        return;
                    }
                    if (e != hipErrorNotReady) {
                        HIP_CHECK(e);
                    }
                    naps = 0;
                }
            }
            return;
        }
        if (i % RUN_AHEAD_EVENT_STEPS != 0) {
            return;
        }
        const int cur = chunk & 1, prev = cur ^ 1;
        if (ev[cur] == nullptr) {
            HIP_CHECK(hipEventCreateWithFlags(&ev[cur], hipEventDisableTiming));
        }
        HIP_CHECK(hipEventRecord(ev[cur], stream));
        pending[cur] = true;
        if (pending[prev]) {
            for (;;) {
                const hipError_t e = hipEventQuery(ev[prev]);
                if (e == hipSuccess) {
                    break;
                }
                if (e != hipErrorNotReady) {
                    HIP_CHECK(e);
                }
                nap();
            }
            pending[prev] = false;
        }
        chunk++;
    }
};

// ------------------------------------------------------------------------------------------------------------
Context::Context(
    int N, const double *x_0, const double *v_0, const double *box_0, std::shared_ptr<Integrator> intg,
    std::vector<std::shared_ptr<BoundPotential>> &bps, std::vector<std::shared_ptr<Mover>> &movers)
    : N_(N), movers_(movers), step_(0), d_x_t_(static_cast<size_t>(N) * 3), d_v_t_(static_cast<size_t>(N) * 3), d_box_t_(9), intg_(intg),
      bps_(bps), stream_(0) {
    d_x_t_.copy_from(x_0);
    d_v_t_.copy_from(v_0);
    d_box_t_.copy_from(box_0);
    for (auto &bp : bps_) {
        collect_nonbonded_cutoffs(bp->potential, nb_cutoffs_with_padding_);
    }
}

Context::~Context() {
    if (ev_start_) {
        (void)hipEventDestroy(ev_start_);
        (void)hipEventDestroy(ev_stop_);
    }
}

double Context::last_multiple_steps_ms() {
    if (!ev_valid_) {
        throw std::runtime_error("no multiple_steps call has been timed yet");
    }
    float ms = 0;
    HIP_CHECK(hipEventSynchronize(ev_stop_));
    HIP_CHECK(hipEventElapsedTime(&ms, ev_start_, ev_stop_));
    return static_cast<double>(ms);
}

void Context::_verify_coords_and_box(const double *coords, const double *box, hipStream_t stream) {
    // reference: context.cu:52-78 (messages are matched by tests/test_md.py:929,1007)
    if (nb_cutoffs_with_padding_.empty()) {
        return;
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    for (double cutoff : nb_cutoffs_with_padding_) {
        const double db_cutoff = 2 * cutoff;
        for (int i = 0; i < 3; i++) {
            if (box[i * 3 + i] < db_cutoff) {
                throw std::runtime_error("cutoff with padding is more than half of the box width, neighborlist is no longer reliable");
            }
        }
    }
    const double max_box_dim = std::max(box[0], std::max(box[4], box[8]));
    const auto mm = std::minmax_element(coords, coords + static_cast<size_t>(N_) * 3);
    if (max_box_dim * 100.0 < *mm.second - *mm.first) {
        throw std::runtime_error("simulation unstable: dimensions of coordinates two orders of magnitude larger than max box dimension");
    }
}

void Context::_step(hipStream_t stream) {
    intg_->step_fwd(bps_, d_x_t_.data, d_v_t_.data, d_box_t_.data, nullptr, stream);
    for (auto &mover : movers_) {
        mover->move(N_, d_x_t_.data, d_box_t_.data, stream);
        if (mover->acted_last_call()) {
            if (mover->kept_potential_inputs()) {
                // (the barostat's fast path: an accepted proposal was committed INTO the potentials' pre-gathered state; only the
                // integrator's own slot-ordered copies of x are behind)
                intg_->invalidate_state_cache();
            } else {
                this->invalidate_potential_inputs(); // coordinates / box may have changed behind the same pointers
            }
        }
    }
    step_ += 1;
}

void Context::invalidate_potential_inputs() {
    for (auto &bp : bps_) {
        bp->potential->invalidate_cached_inputs();
    }
    intg_->invalidate_state_cache(); // coordinates (setters, movers) changed behind the integrator's back as well
}

void Context::multiple_steps(const int n_steps, const int n_samples, double *h_x, double *h_box) {
    if (n_samples < 0) {
        throw std::runtime_error("n_samples < 0");
    }
    const int store_x_interval = n_samples > 0 ? n_steps / n_samples : n_steps + 1;
    if (n_steps % store_x_interval != 0) {
        std::cout << "warning:: n_steps modulo store_x_interval does not equal zero" << std::endl;
    }
    hipStream_t stream = stream_;
    intg_->initialize(bps_, d_x_t_.data, d_v_t_.data, d_box_t_.data, nullptr, stream);
    if (ev_start_ == nullptr) {
        HIP_CHECK(hipEventCreate(&ev_start_));
        HIP_CHECK(hipEventCreate(&ev_stop_));
    }
    ev_valid_ = false;
    if (n_steps > 0) {
        HIP_CHECK(hipEventRecord(ev_start_, stream));
    }
    RunAhead run_ahead;
    for (int i = 1; i <= n_steps; i++) {
        this->_step(stream);
        run_ahead.after_step(i, stream, *intg_);
        if (i == n_steps) {
            HIP_CHECK(hipEventRecord(ev_stop_, stream));
            ev_valid_ = true;
        }
        if (i % store_x_interval == 0) {
            double *box_ptr = h_box + static_cast<size_t>(i / store_x_interval - 1) * 9;
            double *coord_ptr = h_x + static_cast<size_t>(i / store_x_interval - 1) * N_ * 3;
            // (a device-to-host copy into the caller's pageable array blocks the host -- spinning -- until everything enqueued before it
            // has run: wait for that politely first, then the copy is a matter of microseconds)
            wait_for_stream(stream);
            HIP_CHECK(hipMemcpyAsync(coord_ptr, d_x_t_.data, static_cast<size_t>(N_) * 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(box_ptr, d_box_t_.data, 9 * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            this->_verify_coords_and_box(coord_ptr, box_ptr, stream);
        }
    }
    intg_->finalize(bps_, d_x_t_.data, d_v_t_.data, d_box_t_.data, nullptr, stream);
    wait_for_stream(stream);
    for (auto &mover : movers_) {
        mover->after_wait(); // (what the movers' kernels reported in host-visible memory: throws on a reported failure)
    }
}

// The streams a group of contexts is stepped on: created once, one after the other, so that the runtime hands them distinct
// hardware queues (streams that share a hardware queue serialise: two contexts on streams created at unrelated times stepped no
// faster together than alone).  Every Context entry point ends with a synchronisation of the stream it used, so a context can be
// stepped on the null stream by one call and on a group stream by the next.
static hipStream_t group_stream(const size_t k) {
    static std::map<int, std::vector<hipStream_t>> pools; // per device (a process that switches devices gets a pool on each)
    static std::mutex pools_mutex;                        // (threads driving different devices hold different API locks)
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(pools_mutex);
    std::vector<hipStream_t> &pool = pools[dev];
    if (pool.empty()) {
        const char *e = std::getenv("TM_AMD_GROUP_STREAMS");
        const int n = e ? std::max(1, std::atoi(e)) : 8;
        pool.resize(static_cast<size_t>(n));
        for (hipStream_t &st : pool) {
            HIP_CHECK(hipStreamCreate(&st));
        }
    }
    return pool[k % pool.size()];
}

// every Potential object (containers and their children alike) reachable from a context, and its integrator and movers
static void collect_objects(const std::shared_ptr<Potential> &pot, std::vector<const void *> &out) {
    out.push_back(pot.get());
    if (auto f = std::dynamic_pointer_cast<FanoutSummedPotential>(pot)) {
        for (auto &c : f->get_potentials()) {
            collect_objects(c, out);
        }
    } else if (auto sp = std::dynamic_pointer_cast<SummedPotential>(pot)) {
        for (auto &c : sp->get_potentials()) {
            collect_objects(c, out);
        }
    }
}

void Context::multiple_steps_group(const std::vector<Context *> &ctxts, const int n_steps) {
    if (n_steps < 0) {
        throw std::runtime_error("n_steps < 0");
    }
    // Contexts stepped together must share NOTHING that holds device state: a Potential bound twice (one unbound potential, two
    // BoundPotentials -- fine one call after the other) keeps ONE neighbor list and ONE set of accumulators, and an integrator or
    // a mover handed to two contexts keeps one state.  Two streams in the same buffers would not fail, they would be wrong.
    {
        std::vector<std::vector<const void *>> owned(ctxts.size());
        for (size_t k = 0; k < ctxts.size(); k++) {
            for (auto &bp : ctxts[k]->bps_) {
                owned[k].push_back(bp.get());
                collect_objects(bp->potential, owned[k]);
            }
            owned[k].push_back(ctxts[k]->intg_.get());
            for (auto &m : ctxts[k]->movers_) {
                owned[k].push_back(m.get());
                // ... and whatever the mover evaluates on its own: a barostat built on ANOTHER context's bound potentials would
                // run its energy launches in that context's neighbor list and accumulators, on this context's stream
                for (auto &bp : m->held_potentials()) {
                    owned[k].push_back(bp.get());
                    collect_objects(bp->potential, owned[k]);
                }
            }
            std::sort(owned[k].begin(), owned[k].end());
            owned[k].erase(std::unique(owned[k].begin(), owned[k].end()), owned[k].end());
        }
        for (size_t a = 0; a < ctxts.size(); a++) {
            for (size_t b = a + 1; b < ctxts.size(); b++) {
                std::vector<const void *> both;
                std::set_intersection(owned[a].begin(), owned[a].end(), owned[b].begin(), owned[b].end(), std::back_inserter(both));
                if (!both.empty() && ctxts[a] != ctxts[b]) {
                    throw std::runtime_error(
                        "multiple_steps_group: contexts " + std::to_string(a) + " and " + std::to_string(b) +
                        " share a potential, integrator or mover object; contexts stepped together need their own (bind fresh potentials per context)");
                }
            }
        }
    }
    for (size_t a = 0; a < ctxts.size(); a++) {
        for (size_t b = a + 1; b < ctxts.size(); b++) {
            if (ctxts[a] == ctxts[b]) {
                throw std::runtime_error("multiple_steps_group: the contexts must be distinct");
            }
        }
    }
    std::vector<hipStream_t> st(ctxts.size());
    for (size_t k = 0; k < ctxts.size(); k++) {
        st[k] = group_stream(k);
    }
    try {
        for (size_t k = 0; k < ctxts.size(); k++) {
            Context *c = ctxts[k];
            c->intg_->initialize(c->bps_, c->d_x_t_.data, c->d_v_t_.data, c->d_box_t_.data, nullptr, st[k]);
            if (c->ev_start_ == nullptr) {
                HIP_CHECK(hipEventCreate(&c->ev_start_));
                HIP_CHECK(hipEventCreate(&c->ev_stop_));
            }
            c->ev_valid_ = false;
            if (n_steps > 0) {
                HIP_CHECK(hipEventRecord(c->ev_start_, st[k]));
            }
        }
        // Enqueueing a step costs the host 6-10 us (two or three launches and the plan's bookkeeping): plenty of slack against a
        // 55 us DHFR-sized replica-step, but for small systems (8 us per replica-step on the device at 2.2k atoms) one launching
        // thread is the limit.  So the contexts are dealt to up to TM_AMD_GROUP_THREADS host threads, each feeding its own
        // contexts' streams; the contexts share nothing, so the threads need no coordination beyond the join.  Default: two
        // threads while every context is small (<= 5000 atoms), one above -- there the second thread measured nothing (55.0 us per
        // DHFR-sized replica-step either way) and a multi-GPU node has 2 CPUs per rank to give (16-CPU quota, 8 ranks).
        const auto t_enqueue = std::chrono::steady_clock::now();
        const char *e_threads = std::getenv("TM_AMD_GROUP_THREADS");
        int largest = 0;
        for (const Context *c : ctxts) {
            largest = std::max(largest, c->N_);
        }
        const int default_threads = largest <= 5000 ? 2 : 1;
        size_t n_threads = std::min<size_t>(ctxts.size(), static_cast<size_t>(std::max(1, e_threads ? std::atoi(e_threads) : default_threads)));
        if (Profiler::get().enabled()) {
            n_threads = 1; // the per-launch profiler keeps its events in one unsynchronised table
        }
#ifdef TM_GUARD
        n_threads = 1; // (so does the guard-zone allocator of debug builds)
#endif
        if (n_threads <= 1) {
            std::vector<RunAhead> run_ahead(ctxts.size());
            for (int i = 1; i <= n_steps; i++) {
                for (size_t k = 0; k < ctxts.size(); k++) { // one step of every context per round: their launches alternate in the device's queues
                    ctxts[k]->_step(st[k]);
                    run_ahead[k].after_step(i, st[k], *ctxts[k]->intg_);
                }
            }
        } else {
            int dev = 0;
            HIP_CHECK(hipGetDevice(&dev));
            std::vector<std::exception_ptr> failed(n_threads);
            std::vector<std::thread> workers;
            for (size_t w = 0; w < n_threads; w++) {
                workers.emplace_back([&, w]() {
                    try {
                        HIP_CHECK(hipSetDevice(dev)); // (the current device is per-thread state)
                        std::vector<RunAhead> run_ahead(ctxts.size());
                        for (int i = 1; i <= n_steps; i++) {
                            for (size_t k = w; k < ctxts.size(); k += n_threads) {
                                ctxts[k]->_step(st[k]);
                                run_ahead[k].after_step(i, st[k], *ctxts[k]->intg_);
                            }
                        }
                    } catch (...) {
                        failed[w] = std::current_exception();
                    }
                });
            }
            for (std::thread &t : workers) {
                t.join();
            }
            for (const std::exception_ptr &f : failed) {
                if (f) {
                    std::rethrow_exception(f);
                }
            }
        }
        if (std::getenv("TM_AMD_DEBUG_ENQUEUE")) { // how long the host needs to enqueue a step (meaningful while the device's queues do not fill up)
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enqueue).count();
            fprintf(stderr, "[enqueue] %zu contexts x %d steps on %zu host thread(s): %.2f us of host wall time per context-step\n", ctxts.size(), n_steps, n_threads,
                    us / (static_cast<double>(n_steps) * ctxts.size()));
        }
        for (size_t k = 0; k < ctxts.size(); k++) {
            Context *c = ctxts[k];
            if (n_steps > 0) {
                HIP_CHECK(hipEventRecord(c->ev_stop_, st[k]));
                c->ev_valid_ = true;
            }
            c->intg_->finalize(c->bps_, c->d_x_t_.data, c->d_v_t_.data, c->d_box_t_.data, nullptr, st[k]);
        }
    } catch (...) {
        for (hipStream_t q : st) {
            (void)hipStreamSynchronize(q);
        }
        throw;
    }
    for (hipStream_t q : st) {
        wait_for_stream(q);
    }
    for (Context *c : ctxts) {
        for (auto &mover : c->movers_) {
            mover->after_wait();
        }
    }
}

double Context::_get_temperature() const {
    // reference: context.cu:80-88 (only a Langevin thermostat knows a temperature)
    if (auto li = std::dynamic_pointer_cast<LangevinIntegrator<float>>(intg_)) {
        return li->get_temperature();
    }
    if (auto li = std::dynamic_pointer_cast<LangevinIntegrator<double>>(intg_)) {
        return li->get_temperature();
    }
    throw std::runtime_error("integrator must be LangevinIntegrator.");
}

void Context::setup_local_md(const double temperature, const bool freeze_reference) {
    if (local_md_pots_ != nullptr) {
        if (local_md_pots_->temperature != temperature || local_md_pots_->freeze_reference != freeze_reference) {
            throw std::runtime_error(
                "local md configured with different parameters, current parameters: Temperature " +
                std::to_string(local_md_pots_->temperature) + " Freeze Reference " + std::to_string(local_md_pots_->freeze_reference));
        }
        return;
    }
    local_md_pots_.reset(new LocalMDPotentials(N_, bps_, freeze_reference, temperature));
}

void Context::_ensure_local_md_initialized() {
    if (local_md_pots_ == nullptr) {
        this->setup_local_md(this->_get_temperature(), true);
    }
}

int Context::local_md_last_reference() const { return local_md_pots_ ? local_md_pots_->last_reference_idx() : -1; }

std::vector<unsigned int> Context::local_md_last_free_idxs() const {
    return local_md_pots_ ? local_md_pots_->last_free_idxs() : std::vector<unsigned int>();
}

// the steps of a local-MD call, after its setup: only the free atoms are integrated, movers stay out (context.cu:268)
void Context::_run_local_steps(const int n_steps, const int n_samples, double *h_x, double *h_box) {
    const int store_x_interval = n_samples > 0 ? n_steps / n_samples : n_steps + 1;
    hipStream_t stream = stream_;
    unsigned int *d_free_idxs = local_md_pots_->get_free_idxs();
    std::vector<std::shared_ptr<BoundPotential>> &local_pots = local_md_pots_->get_potentials();
    // whatever the global-MD steps before this call left with the potentials / the integrator (pre-gathered positions,
    // slot-ordered state) describes another atom set
    this->invalidate_potential_inputs();
    try {
        intg_->initialize(local_pots, d_x_t_.data, d_v_t_.data, d_box_t_.data, d_free_idxs, stream);
        for (int i = 1; i <= n_steps; i++) {
            intg_->step_fwd(local_pots, d_x_t_.data, d_v_t_.data, d_box_t_.data, d_free_idxs, stream);
            step_ += 1;
            if (i % store_x_interval == 0) {
                double *box_ptr = h_box + static_cast<size_t>(i / store_x_interval - 1) * 9;
                double *coord_ptr = h_x + static_cast<size_t>(i / store_x_interval - 1) * N_ * 3;
                wait_for_stream(stream); // (see Context::multiple_steps)
                HIP_CHECK(hipMemcpyAsync(coord_ptr, d_x_t_.data, static_cast<size_t>(N_) * 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipMemcpyAsync(box_ptr, d_box_t_.data, 9 * sizeof(double), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                this->_verify_coords_and_box(coord_ptr, box_ptr, stream);
            }
        }
        intg_->finalize(local_pots, d_x_t_.data, d_v_t_.data, d_box_t_.data, d_free_idxs, stream);
    } catch (...) {
        (void)hipStreamSynchronize(stream);
        local_md_pots_->reset_potentials();
        this->invalidate_potential_inputs();
        throw;
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    local_md_pots_->reset_potentials();
    this->invalidate_potential_inputs();
}

void Context::multiple_steps_local(
    const int n_steps, const std::vector<int> &local_idxs, const int n_samples, const double radius, const double k, const int seed,
    double *h_x, double *h_box) {
    if (n_samples < 0) {
        throw std::runtime_error("n_samples < 0");
    }
    const int store_x_interval = n_samples > 0 ? n_steps / n_samples : n_steps + 1;
    if (n_steps % store_x_interval != 0) {
        std::cout << "warning:: n_steps modulo store_x_interval does not equal zero" << std::endl;
    }
    this->_ensure_local_md_initialized();
    try {
        local_md_pots_->setup_from_idxs(d_x_t_.data, d_box_t_.data, local_idxs, seed, radius, k, stream_);
    } catch (...) {
        local_md_pots_->reset_potentials(); // a setup that threw half-way may already have narrowed the all-pairs potential
        throw;
    }
    this->_run_local_steps(n_steps, n_samples, h_x, h_box);
}

void Context::multiple_steps_local_selection(
    const int n_steps, const int reference_idx, const std::vector<int> &selection_idxs, const int n_samples, const double radius,
    const double k, double *h_x, double *h_box) {
    if (n_samples < 0) {
        throw std::runtime_error("n_samples < 0");
    }
    const int store_x_interval = n_samples > 0 ? n_steps / n_samples : n_steps + 1;
    if (n_steps % store_x_interval != 0) {
        std::cout << "warning:: n_steps modulo store_x_interval does not equal zero" << std::endl;
    }
    this->_ensure_local_md_initialized();
    try {
        local_md_pots_->setup_from_selection(reference_idx, selection_idxs, radius, k, stream_);
    } catch (...) {
        local_md_pots_->reset_potentials();
        throw;
    }
    this->_run_local_steps(n_steps, n_samples, h_x, h_box);
}

void Context::step() {
    this->_step(stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));
    for (auto &mover : movers_) {
        mover->after_wait();
    }
}

void Context::initialize() {
    intg_->initialize(bps_, d_x_t_.data, d_v_t_.data, d_box_t_.data, nullptr, stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));
}

void Context::finalize() {
    intg_->finalize(bps_, d_x_t_.data, d_v_t_.data, d_box_t_.data, nullptr, stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));
}

void Context::set_x_t(const double *in) {
    d_x_t_.copy_from(in);
    this->invalidate_potential_inputs();
}
void Context::set_v_t(const double *in) {
    d_v_t_.copy_from(in);
    intg_->invalidate_state_cache();
}
void Context::set_box(const double *in) {
    d_box_t_.copy_from(in);
    this->invalidate_potential_inputs();
}
void Context::get_x_t(double *out) const { d_x_t_.copy_to(out); }
void Context::get_v_t(double *out) const { d_v_t_.copy_to(out); }
void Context::get_box(double *out) const { d_box_t_.copy_to(out); }

} // namespace tmamd
