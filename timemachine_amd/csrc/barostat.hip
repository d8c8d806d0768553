// MonteCarloBarostat for gfx950.  reference: cpp/src/barostat.cu:19-259, kernels/k_barostat.cuh:10-189,
// mol_utils.cpp:8-85 (group validation / flattening), mover.cu:7-23.
//
// One attempt = memcpy x -> x_proposed, four small kernels, two energy-only evaluations of every bound potential and
// two 128-bit reductions, all stream-ordered; no host synchronisation.  Differences from the reference, on purpose:
// the two uniforms come from a counter-based Philox4x32-10 keyed on (seed; attempt) inside the kernels instead of a
// cuRAND batch buffer (statistically equivalent, reproducible per seed on this implementation only).
#include "engine.hpp"
#include "fixed_point.hip.hpp"
#include "nb_snapshot_test.hip.hpp"
#include "philox.hip.hpp"

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <set>

namespace tmamd {

static const double BOLTZ_KJ = 0.008314462618; // cpp/src/constants.hpp:5
static const double AVOGADRO = 6.0221367e23;   // cpp/src/constants.hpp:6

// (volume, volume_delta, length_scale, second uniform) of this attempt + the proposed box.
template <typename Real>
__global__ void k_barostat_propose(
    const int adaptive, const unsigned long long seed, const unsigned long long attempt, const double *__restrict__ box,
    double *__restrict__ volume_scale, Real *__restrict__ mv, double *__restrict__ box_proposed,
    // the rest of the grid prepares the attempt's buffers in the same launch: x_proposed = x, molecule centroid sums = 0
    const int n_x, const double *__restrict__ x, double *__restrict__ x_proposed, const int n_centroids, u64 *__restrict__ centroids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_x) {
        x_proposed[i] = x[i];
    }
    if (i < n_centroids) {
        centroids[i] = 0;
    }
    if (i != 0) {
        return;
    }
    unsigned int r[4];
    philox4x32_10(static_cast<unsigned int>(attempt), static_cast<unsigned int>(attempt >> 32), 0x4241524fu, 0x53544154u,
                  static_cast<unsigned int>(seed), static_cast<unsigned int>(seed >> 32), r);
    // uniforms in (0, 1]: the convention of curandGenerateUniform the reference draws from
    const Real u1 = static_cast<Real>((static_cast<double>(r[0]) + 1.0) * (1.0 / 4294967296.0));
    const Real u2 = static_cast<Real>((static_cast<double>(r[1]) + 1.0) * (1.0 / 4294967296.0));
    const Real volume = static_cast<Real>(box[0] * box[4] * box[8]);
    if (adaptive && *volume_scale == 0.0) {
        *volume_scale = 0.01 * volume; // k_barostat.cuh:110-112: first attempt, 1 % of the box volume
    }
    const Real delta = static_cast<Real>(*volume_scale * 2 * (u1 - static_cast<Real>(0.5)));
    const Real new_volume = volume + delta;
    const Real scale = cbrt(new_volume / volume);
    mv[0] = volume;
    mv[1] = delta;
    mv[2] = scale;
    mv[3] = u2;
    for (int k = 0; k < 9; k++) {
        box_proposed[k] = box[k];
    }
    box_proposed[0] *= scale;
    box_proposed[4] *= scale;
    box_proposed[8] *= scale;
}

// Fixed-point sums of the molecules' coordinates.  A molecule's atoms are consecutive in the flattened group list, so the
// lanes of a wave that belong to one molecule form a run: the run's sum is formed in registers (a segmented scan over the
// lanes, six shuffle steps) and its LAST lane issues the atomics -- one per run and component instead of one per atom.
// (One atomic per atom made the 83 atoms of a solute chain queue on a single cache line: 18.9 us for this kernel at 23.5k
// atoms, memory-side atomics being served one after the other per line.)  Integer sums: the same bits in any order.
template <typename Real>
__global__ void k_barostat_centroids(
    const int n_grouped, const double *__restrict__ x, const int *__restrict__ atom_idxs, const int *__restrict__ mol_idxs,
    u64 *__restrict__ centroids) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = idx < n_grouped;
    const int m = valid ? mol_idxs[idx] : -1 - lane; // invalid lanes: runs of their own
    u64 v[3] = {0, 0, 0};
    if (valid) {
        const int a = atom_idxs[idx];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            v[d] = float_to_fixed<Real>(static_cast<Real>(x[a * 3 + d]));
        }
    }
    // segmented inclusive scan: after step o a lane holds the sum of its run's members among the 2 o lanes ending at it
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int m_up = __shfl_up(m, o, 64);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const u64 up = __shfl_up(v[d], o, 64);
            if (lane >= o && m_up == m) {
                v[d] += up;
            }
        }
    }
    const int m_next = __shfl_down(m, 1, 64);
    if (valid && (lane == 63 || m_next != m)) { // the last lane of a run carries its sum
#pragma unroll
        for (int d = 0; d < 3; d++) {
            atomicAdd(centroids + m * 3 + d, v[d]);
        }
    }
}

// Every grouped atom follows its molecule's centroid: the centroid is scaled about the box centre and wrapped into
// the scaled home box (k_barostat.cuh:10-69); intramolecular geometry is untouched.
template <typename Real>
__global__ void k_barostat_rescale(
    const int n_grouped, double *__restrict__ x_proposed, const Real *__restrict__ mv, const double *__restrict__ box,
    const int *__restrict__ atom_idxs, const int *__restrict__ mol_idxs, const int *__restrict__ mol_offsets,
    const u64 *__restrict__ centroids) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_grouped) {
        return;
    }
    const Real scale = mv[2];
    const int a = atom_idxs[idx], m = mol_idxs[idx];
    const Real n_atoms = static_cast<Real>(mol_offsets[m + 1] - mol_offsets[m]);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const Real edge = static_cast<Real>(box[d * 4]);
        const Real centre = edge * static_cast<Real>(0.5);
        Real c = fixed_to_float<Real>(centroids[m * 3 + d]) / n_atoms;
        const Real displacement = ((c - centre) * scale) + centre - c;
        c += displacement;
        const Real scaled_edge = edge * scale;
        const Real home = scaled_edge * floor(c / scaled_edge);
        x_proposed[a * 3 + d] += static_cast<double>(displacement - home);
    }
}

__device__ __forceinline__ bool energy_overflowed(const i128 v) { return fixed_point_overflow(v); }

// Metropolis test + bookkeeping + (on acceptance) x <- x_proposed, box <- box_proposed  (k_barostat.cuh:125-189)
template <typename Real>
__global__ void k_barostat_decide(
    const int N, const int adaptive, const int num_molecules, const double kT, const double pressure, const Real *__restrict__ mv,
    double *__restrict__ volume_scale, const i128 *__restrict__ u_init, const i128 *__restrict__ u_final, double *__restrict__ box,
    const double *__restrict__ box_proposed, double *__restrict__ x, const double *__restrict__ x_proposed, int *__restrict__ counters) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const Real volume = mv[0], delta = mv[1], u2 = mv[3];
    const Real new_volume = volume + delta;
    Real energy_delta = INFINITY;
    if (!energy_overflowed(u_final[0]) && !energy_overflowed(u_init[0])) {
        energy_delta = static_cast<Real>(static_cast<double>(static_cast<long long>(u_final[0] - u_init[0])) / static_cast<double>(TM_FIXED_EXPONENT));
    }
    const Real w = static_cast<Real>(energy_delta + pressure * delta - num_molecules * kT * log(new_volume / volume));
    const bool rejected = w > 0 && u2 > static_cast<Real>(exp(-w / kT));
    if (idx == 0) {
        if (!rejected) {
            counters[0]++;
        }
        counters[1]++;
        if (adaptive && counters[1] >= 10) {
            if (counters[0] < 0.25 * counters[1]) {
                volume_scale[0] /= 1.1;
                counters[0] = 0;
                counters[1] = 0;
            } else if (counters[0] > 0.75 * counters[1]) {
                volume_scale[0] = fmin(volume_scale[0] * 1.1, static_cast<double>(volume) * 0.3);
                counters[0] = 0;
                counters[1] = 0;
            }
        }
    }
    if (rejected || idx >= N) {
        return;
    }
    if (idx < 9) {
        box[idx] = box_proposed[idx];
    }
#pragma unroll
    for (int d = 0; d < 3; d++) {
        x[idx * 3 + d] = x_proposed[idx * 3 + d];
    }
}

// =============================================================================================================
// The fast path (round 5): both energies of an attempt on the nonbonded potential's CURRENT neighbor list and sorted records.
//
// After an MD step the integrator's update kernel has left the nonbonded potential's next gather done (engine.hpp:
// PregatherTarget): sorted records of x, the rebuild flag, the block bounds.  The reference-shaped attempt above throws that away
// twice (the proposal's evaluation gathers x' over it, the next MD step gathers x or x' again): fifteen small launches around the
// two energy launches.  Here the attempt is FOUR launches + the list launch:
//   [k_barostat_centroids]     only when a molecule has more than BAROSTAT_INLINE_MOL atoms: the centroid sums of the large ones
//   k_barostat_propose_probe   the proposal in one pass -- scale, molecule centroids (summed in-thread for the small
//                              molecules), x' in atom order and as a second set of sorted records next to the
//                              potential's own (ProbeTarget::gathered2), and the list-validity test of x' (raising the flag the list
//                              launch reads: a proposal the current list cannot vouch for rebuilds it, from x, first)
//   [k_find_ixns]              the list launch every evaluation makes (exits unless flagged)
//   k_nonbonded_tiles x 2      energy-only, on (gathered, box) and on (gathered2, box'); the plan's bonded terms and exclusions ride
//                              along; one partial sum per workgroup
//   k_barostat_decide_commit   every workgroup adds up the partial sums for itself, makes the Metropolis decision (the arithmetic of
//                              k_barostat_decide) and, on acceptance, commits the proposal INTO the pre-gathered state: x, box, the
//                              sorted records, the next call's rebuild flag, the block bounds -- so that the next MD step finds its
//                              inputs exactly as after an ordinary step (rejected: nothing was touched).
// Same energies bit for bit (the same per-pair / per-term functions on the same operands, integer sums), hence the same decisions
// and the same trajectories as the path above (tests/test_gpu_parity.py::test_barostat_follows_model_attempt_by_attempt,
// tests/test_gpu_barostat_cases.py run on both paths).
bool g_barostat_fast_path = std::getenv("TM_AMD_BAROSTAT_SLOW_PATH") == nullptr;
// both geometries in one tile launch (k_nonbonded_tiles<..., DUAL>) or one launch each (A/B switch, TM_AMD_BAROSTAT_TWO_LAUNCHES)
static const bool g_barostat_dual_launch = std::getenv("TM_AMD_BAROSTAT_TWO_LAUNCHES") == nullptr;
// molecules up to this size (waters, ions) have their centroid summed by each of their atoms' threads; larger ones by the
// segmented-scan kernel in a launch of its own (measured: 83-atom chains summed in-thread made the proposal kernel a 29 us chain of
// dependent loads and conversions per thread)
static const int BAROSTAT_INLINE_MOL = 8;

// what the proposal is, from the attempt's Philox draw and the box: the arithmetic of k_barostat_propose
template <typename Real> struct Proposal {
    Real volume, delta, scale, u2;
};
template <typename Real>
__device__ __forceinline__ Proposal<Real> draw_proposal(const int adaptive, const unsigned long long seed, const unsigned long long attempt, const double *__restrict__ box, const double volume_scale_now) {
    unsigned int r[4];
    philox4x32_10(static_cast<unsigned int>(attempt), static_cast<unsigned int>(attempt >> 32), 0x4241524fu, 0x53544154u,
                  static_cast<unsigned int>(seed), static_cast<unsigned int>(seed >> 32), r);
    const Real u1 = static_cast<Real>((static_cast<double>(r[0]) + 1.0) * (1.0 / 4294967296.0));
    Proposal<Real> p;
    p.u2 = static_cast<Real>((static_cast<double>(r[1]) + 1.0) * (1.0 / 4294967296.0));
    p.volume = static_cast<Real>(box[0] * box[4] * box[8]);
    const double vs = (adaptive && volume_scale_now == 0.0) ? 0.01 * p.volume : volume_scale_now; // first attempt: 1 % of the box volume
    p.delta = static_cast<Real>(vs * 2 * (u1 - static_cast<Real>(0.5)));
    const Real new_volume = p.volume + p.delta;
    p.scale = cbrt(new_volume / p.volume);
    return p;
}

template <typename Real, typename GReal>
__global__ __launch_bounds__(256) void k_barostat_propose_probe(
    const int N, const int adaptive, const unsigned long long seed, const unsigned long long attempt, const double *__restrict__ box,
    double *__restrict__ volume_scale, Real *__restrict__ mv, double *__restrict__ box_proposed, const double *__restrict__ x,
    double *__restrict__ x_proposed, const int4 *__restrict__ mol_of_atom, // per atom {molecule or -1, its first entry in atom_idxs, its size, its first atom if its atoms are consecutive else -1}
    const int *__restrict__ atom_idxs,
    const u64 *__restrict__ centroids, // sums of the molecules larger than BAROSTAT_INLINE_MOL (k_barostat_centroids ran first), else unused
    float *__restrict__ r2_blocks,     // [gridDim.x]: this block's largest |atom - own molecule's centroid|^2 (the DUAL tile launch's filter margin)
    unsigned int *__restrict__ overreach, // host-visible: set to attempt + 1 by the first proposal the list cannot vouch for however fresh
    const ProbeTarget t) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ float s_r2[4];
    float r2_mine = 0.0f;
    const double vs_now = *volume_scale;
    const Proposal<Real> p = draw_proposal<Real>(adaptive, seed, attempt, box, vs_now);
    double bp[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        bp[k] = box[k];
    }
    bp[0] *= p.scale;
    bp[4] *= p.scale;
    bp[8] *= p.scale;
    if (a == 0) {
        if (adaptive && vs_now == 0.0) {
            *volume_scale = 0.01 * p.volume; // (every thread has formed the same value for itself)
        }
        mv[0] = p.volume;
        mv[1] = p.delta;
        mv[2] = p.scale;
        mv[3] = p.u2;
        for (int k = 0; k < 9; k++) {
            box_proposed[k] = bp[k];
        }
    }
    if (a < 8) {
        static_cast<GReal *>(t.gathered2)[static_cast<size_t>(t.n) * 8 + a] = 0; // the sentinel record padded list slots point at
        if (t.second_records != 0) {
            static_cast<GReal *>(t.gathered2)[static_cast<size_t>(t.second_records + t.n) * 8 + a] = 0;
        }
    }
    if (a < N) {
    double xp[3] = {x[a * 3 + 0], x[a * 3 + 1], x[a * 3 + 2]};
    const int4 info = mol_of_atom[a]; // (one load: the molecule's extent arrives with its number)
    const int m = info.x;
    if (m >= 0) {
        const int first = info.y, last = info.y + info.z;
        u64 sum[3] = {0, 0, 0};
        if (last - first <= BAROSTAT_INLINE_MOL) {
            for (int k = first; k < last; k++) { // integer sums: the bits of k_barostat_centroids in any order
                const int b = info.w >= 0 ? info.w + (k - first) : atom_idxs[k];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    sum[d] += float_to_fixed<Real>(static_cast<Real>(x[b * 3 + d]));
                }
            }
        } else {
#pragma unroll
            for (int d = 0; d < 3; d++) {
                sum[d] = centroids[m * 3 + d];
            }
        }
        const Real n_atoms = static_cast<Real>(last - first);
#pragma unroll
        for (int d = 0; d < 3; d++) { // the arithmetic of k_barostat_rescale
            const Real edge = static_cast<Real>(box[d * 4]);
            const Real centre = edge * static_cast<Real>(0.5);
            Real c = fixed_to_float<Real>(sum[d]) / n_atoms;
            const float off = static_cast<float>(xp[d] - static_cast<double>(c));
            r2_mine = __builtin_fmaf(off, off, r2_mine);
            const Real displacement = ((c - centre) * p.scale) + centre - c;
            c += displacement;
            const Real scaled_edge = edge * p.scale;
            const Real home = scaled_edge * floor(c / scaled_edge);
            xp[d] += static_cast<double>(displacement - home);
        }
        r2_mine *= 1.0001f; // (f32 rounding of the three squares)
    }
    x_proposed[a * 3 + 0] = xp[0];
    x_proposed[a * 3 + 1] = xp[1];
    x_proposed[a * 3 + 2] = xp[2];
    // the proposal as the tile kernel reads it: a sorted record next to the current geometry's
    const int slot = t.slot_of_atom[a];
    const GReal *g = static_cast<const GReal *>(t.gathered) + static_cast<size_t>(slot) * 8;
    GReal *g2 = static_cast<GReal *>(t.gathered2) + static_cast<size_t>(slot) * 8;
    g2[0] = static_cast<GReal>(xp[0]);
    g2[1] = static_cast<GReal>(xp[1]);
    g2[2] = static_cast<GReal>(xp[2]);
    g2[3] = g[3];
    g2[4] = g[4];
    g2[5] = g[5];
    g2[6] = g[6];
    g2[7] = 0;
    if (t.second_records != 0) { // merged producers: the atom's second record (the group's parameters) follows the proposal too
        const GReal *gb = g + static_cast<size_t>(t.second_records) * 8;
        GReal *g2b = g2 + static_cast<size_t>(t.second_records) * 8;
        g2b[0] = g2[0];
        g2b[1] = g2[1];
        g2b[2] = g2[2];
        g2b[3] = gb[3];
        g2b[4] = gb[4];
        g2b[5] = gb[5];
        g2b[6] = gb[6];
        g2b[7] = 0;
    }
    // can the current list vouch for the proposal?  If not, the list launch that follows rebuilds it -- from the CURRENT geometry,
    // whose records, block bounds and snapshot source are all in place -- and the proposal then sits a proposal's displacement
    // from a fresh snapshot.
    if (t.scale_aware && snapshot_calls_for_rebuild(xp[0], xp[1], xp[2], t.snap_x + a * 3, bp, t.snap_box, t.threshold2)) {
        if (atomicExch(t.flag_probe, 1) == 0) { // the first to raise it resets the counters the build accumulates into (PregatherTarget)
            t.nbl_counters[0] = 0;
            t.nbl_counters[1] = 0;
            t.nbl_counters[2] = 0;
            for (int k = NB_COUNTER_CLASS0; k < NB_NUM_COUNTERS; k++) {
                t.nbl_counters[k] = 0;
            }
        }
    }
    } // a < N
    // the block's largest atom-to-centroid distance (every thread of the block gets here)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        r2_mine = fmaxf(r2_mine, __shfl_xor(r2_mine, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        s_r2[threadIdx.x >> 6] = r2_mine;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float r2 = fmaxf(fmaxf(s_r2[0], s_r2[1]), fmaxf(s_r2[2], s_r2[3]));
        r2_blocks[blockIdx.x] = r2;
        // A pair inside the cutoff in the proposal is at most (rc + 2 R |s - 1|) / (1 - |s - 1|) apart in the current geometry (the DUAL
        // launch's filter cutoff, with this block's R: the bound grows with R, so some block sees the violation iff the largest R does).
        // Beyond cutoff + padding even the list the probe has just rebuilt from the current geometry does not hold it: this attempt's
        // proposal energy may miss pairs.  Said out loud (MonteCarloBarostat::after_wait throws) instead of deciding on a wrong energy
        // in silence: only moves of several per cent of the box length get here (|s - 1| > padding / (rc + padding + 2 R), roughly).
        if (t.list_reach > 0.0) {
            const double ds = fabs(static_cast<double>(p.scale) - 1.0);
            const double need = ds < 0.5 ? (t.cutoff + 2.0 * sqrt(static_cast<double>(r2)) * ds * 1.001) / (1.0 - ds) + 1e-6 : 1e30;
            if (need > t.list_reach && *overreach == 0u) {
                *overreach = static_cast<unsigned int>(attempt) + 1u;
            }
        }
    }
}

// The decision of k_barostat_decide on energies that arrive as per-workgroup partial sums, and the commit of an accepted proposal
// into the potential's pre-gathered state.  One thread per sorted SLOT, 64-thread workgroups (two 32-atom blocks per wave: their
// bounds fall out of shuffles, as in the integrator's sorted update kernel).
template <typename Real, typename GReal>
__global__ __launch_bounds__(64) void k_barostat_decide_commit(
    const int N, const int adaptive, const int num_molecules, const double kT, const double pressure, const Real *__restrict__ mv,
    double *__restrict__ volume_scale, const i128 *__restrict__ u_init_partials, const int n_init, const i128 *__restrict__ u_final_partials,
    const int n_final, double *__restrict__ box, const double *__restrict__ box_proposed, double *__restrict__ x,
    const double *__restrict__ x_proposed, int *__restrict__ counters, u64 *__restrict__ centroids, const int n_centroids, const ProbeTarget t) {
    const int lane = threadIdx.x;
    i128 e0 = 0, e1 = 0;
    for (int k = lane; k < n_init; k += 64) {
        e0 += u_init_partials[k];
    }
    for (int k = lane; k < n_final; k += 64) {
        e1 += u_final_partials[k];
    }
    // (sums in every lane: butterfly instead of wave_sum_i128's lane-0 total)
    e0 = wave_sum_i128(e0);
    e1 = wave_sum_i128(e1);
    u64 lo0 = static_cast<u64>(e0), lo1 = static_cast<u64>(e1);
    long long hi0 = static_cast<long long>(e0 >> 64), hi1 = static_cast<long long>(e1 >> 64);
    lo0 = __shfl(lo0, 0, 64);
    hi0 = __shfl(hi0, 0, 64);
    lo1 = __shfl(lo1, 0, 64);
    hi1 = __shfl(hi1, 0, 64);
    const i128 u_init = (static_cast<i128>(hi0) << 64) | static_cast<i128>(lo0);
    const i128 u_final = (static_cast<i128>(hi1) << 64) | static_cast<i128>(lo1);
    const Real volume = mv[0], delta = mv[1], u2 = mv[3];
    const Real new_volume = volume + delta;
    Real energy_delta = INFINITY;
    if (!energy_overflowed(u_final) && !energy_overflowed(u_init)) {
        energy_delta = static_cast<Real>(static_cast<double>(static_cast<long long>(u_final - u_init)) / static_cast<double>(TM_FIXED_EXPONENT));
    }
    const Real w = static_cast<Real>(energy_delta + pressure * delta - num_molecules * kT * log(new_volume / volume));
    const bool rejected = w > 0 && u2 > static_cast<Real>(exp(-w / kT));
    const int slot = blockIdx.x * 64 + lane;
    if (slot == 0) {
        if (!rejected) {
            counters[0]++;
        }
        counters[1]++;
        if (adaptive && counters[1] >= 10) {
            if (counters[0] < 0.25 * counters[1]) {
                volume_scale[0] /= 1.1;
                counters[0] = 0;
                counters[1] = 0;
            } else if (counters[0] > 0.75 * counters[1]) {
                volume_scale[0] = fmin(volume_scale[0] * 1.1, static_cast<double>(volume) * 0.3);
                counters[0] = 0;
                counters[1] = 0;
            }
        }
        *t.flag_probe = 0; // consumed by the probe's list launch; the next call but one reads it again (the update kernel's flag_clear)
    }
    for (int k = slot; k < n_centroids; k += gridDim.x * 64) {
        centroids[k] = 0; // for the next attempt's k_barostat_centroids (molecules beyond BAROSTAT_INLINE_MOL)
    }
    if (rejected) {
        return; // (uniform across the launch) x, box, the sorted records, the flags, the bounds: all as the last MD step left them
    }
    // (N counts the SLOTS of the potential's order: a merged order has holes, perm == 0xffffffff)
    const int a = slot < N ? static_cast<int>(t.perm[slot]) : -1;
    const bool valid = a >= 0;
    GReal p[3] = {0, 0, 0};
    if (valid) {
        const double xp[3] = {x_proposed[a * 3 + 0], x_proposed[a * 3 + 1], x_proposed[a * 3 + 2]};
        x[a * 3 + 0] = xp[0];
        x[a * 3 + 1] = xp[1];
        x[a * 3 + 2] = xp[2];
        const GReal *g2 = static_cast<const GReal *>(t.gathered2) + static_cast<size_t>(slot) * 8;
        GReal *g = static_cast<GReal *>(t.gathered) + static_cast<size_t>(slot) * 8;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            p[d] = g2[d];
            g[d] = p[d];
        }
        if (t.second_records != 0) { // merged producers: the atom's second record follows
            GReal *gb = static_cast<GReal *>(t.gathered) + (static_cast<size_t>(t.second_records) + slot) * 8;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                gb[d] = p[d];
            }
        }
        // the committed geometry against the list's snapshot as it is NOW (the probe's list launch may have rebuilt it)
        if (t.scale_aware && snapshot_calls_for_rebuild(xp[0], xp[1], xp[2], t.snap_x + a * 3, box_proposed, t.snap_box, t.threshold2)) {
            if (atomicExch(t.flag_next, 1) == 0) {
                t.nbl_counters[0] = 0;
                t.nbl_counters[1] = 0;
                t.nbl_counters[2] = 0;
                for (int k = NB_COUNTER_CLASS0; k < NB_NUM_COUNTERS; k++) {
                    t.nbl_counters[k] = 0;
                }
            }
        }
    }
    // the new box (every thread reads box_proposed, never box: thread 0 rewrites it here)
    if (slot < 9) {
        box[slot] = box_proposed[slot];
    }
    // bounding boxes of the wave's two 32-slot blocks in the NEW box (the arithmetic of k_update_forward_baoab_sorted / k_block_bounds)
    const GReal half = static_cast<GReal>(0.5);
    const int first = lane & 32;
    GReal lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const GReal b = static_cast<GReal>(box_proposed[d * 4]);
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
    if (sub < 3 && (slot - sub) < N) {
        const int blk = slot >> 5;
        const GReal l = sub == 0 ? lo[0] : (sub == 1 ? lo[1] : lo[2]);
        const GReal h = sub == 0 ? hi[0] : (sub == 1 ? hi[1] : hi[2]);
        static_cast<GReal *>(t.blk_ctr)[blk * 3 + sub] = half * (h + l);
        static_cast<GReal *>(t.blk_ext)[blk * 3 + sub] = half * (h - l);
    }
}

// ---- host ---------------------------------------------------------------------------------------------------
static void verify_group_idxs(const int N, const std::vector<std::vector<int>> &group_idxs) { // mol_utils.cpp:8-27
    size_t n = 0;
    std::set<int> seen;
    for (const auto &atoms : group_idxs) {
        n += atoms.size();
        for (int a : atoms) {
            if (a < 0 || a >= N) {
                throw std::runtime_error("Grouped indices must be between 0 and N");
            }
            seen.insert(a);
        }
    }
    if (seen.size() != n) {
        throw std::runtime_error("All grouped indices must be unique");
    }
}

void Mover::move_host(const int N, const double *h_x, const double *h_box, double *h_x_out, double *h_box_out) {
    DeviceBuffer<double> d_x(static_cast<size_t>(N) * 3), d_box(9);
    d_x.copy_from(h_x);
    d_box.copy_from(h_box);
    this->move(N, d_x.data, d_box.data, 0);
    HIP_CHECK(hipStreamSynchronize(0));
    d_x.copy_to(h_x_out);
    d_box.copy_to(h_box_out);
}

template <typename Real>
MonteCarloBarostat<Real>::MonteCarloBarostat(
    const int N, const double pressure, const double temperature, const std::vector<std::vector<int>> &group_idxs, const int interval,
    const std::vector<std::shared_ptr<BoundPotential>> &bps, const int seed, const bool adaptive_scaling_enabled,
    const double initial_volume_scale_factor)
    : Mover(interval), N_(N), adaptive_(adaptive_scaling_enabled), bps_(bps), pressure_(static_cast<Real>(pressure)),
      temperature_(static_cast<Real>(temperature)), seed_(static_cast<unsigned long long>(static_cast<long long>(seed))),
      num_mols_(static_cast<int>(group_idxs.size())), num_grouped_atoms_(0), attempt_(0) {
    this->set_interval(interval); // validates
    for (auto &bp : bps_) {
        bp->potential->expect_box_scaling();
    }
    if (temperature < 100.0) {
        std::cout << "warning temperature less than 100K" << std::endl;
    }
    if (pressure > 10.0) {
        std::cout << "warning pressure more than 10bar" << std::endl;
    }
    verify_group_idxs(N, group_idxs);
    std::vector<int> atom_idxs, mol_idxs, mol_offsets(num_mols_ + 1, 0);
    for (int m = 0; m < num_mols_; m++) {
        std::vector<int> atoms = group_idxs[m];
        std::sort(atoms.begin(), atoms.end());
        mol_offsets[m] = static_cast<int>(atom_idxs.size());
        for (int a : atoms) {
            atom_idxs.push_back(a);
            mol_idxs.push_back(m);
        }
    }
    mol_offsets[num_mols_] = static_cast<int>(atom_idxs.size());
    num_grouped_atoms_ = static_cast<int>(atom_idxs.size());

    d_x_proposed_.realloc(static_cast<size_t>(N_) * 3);
    d_box_proposed_.realloc(9);
    d_volume_scale_.realloc(1);
    d_volume_scale_.copy_from(&initial_volume_scale_factor);
    d_move_.realloc(4);
    d_u_buffer_.realloc(std::max<size_t>(bps_.size(), 1));
    d_u_init_.realloc(1);
    d_u_final_.realloc(1);
    d_centroids_.realloc(static_cast<size_t>(std::max(num_mols_, 1)) * 3);
    d_atom_idxs_.realloc(num_grouped_atoms_);
    d_mol_idxs_.realloc(num_grouped_atoms_);
    d_mol_offsets_.realloc(num_mols_ + 1);
    if (num_grouped_atoms_ > 0) {
        d_atom_idxs_.copy_from(atom_idxs.data());
        d_mol_idxs_.copy_from(mol_idxs.data());
    }
    d_mol_offsets_.copy_from(mol_offsets.data());
    d_counters_.realloc(2);
    this->reset_counters();
    // the fast path's view of the groups: the molecule of every atom
    std::vector<int4> mol_of_atom(static_cast<size_t>(N_), make_int4(-1, 0, 0, -1));
    for (int m = 0; m < num_mols_; m++) {
        const int first = mol_offsets[m], size = mol_offsets[m + 1] - mol_offsets[m];
        max_mol_size_ = std::max(max_mol_size_, size);
        bool consecutive = true; // (atom_idxs is sorted within a molecule)
        for (int k = first + 1; k < first + size; k++) {
            consecutive = consecutive && atom_idxs[k] == atom_idxs[k - 1] + 1;
        }
        for (int k = first; k < first + size; k++) {
            mol_of_atom[atom_idxs[k]] = make_int4(m, first, size, consecutive ? atom_idxs[first] : -1);
        }
    }
    d_mol_of_atom_.realloc(std::max(N_, 1));
    if (N_ > 0) {
        d_mol_of_atom_.copy_from(mol_of_atom.data());
    }
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h_overreach_), sizeof(unsigned int), hipHostMallocDefault));
    *h_overreach_ = 0u;
}

template <typename Real> MonteCarloBarostat<Real>::~MonteCarloBarostat() {
    if (h_overreach_ != nullptr) {
        (void)hipHostFree(h_overreach_);
    }
}

// (the Context has waited for the stream: what the proposal kernels wrote is visible)
template <typename Real> void MonteCarloBarostat<Real>::after_wait() {
    if (h_overreach_ != nullptr && *h_overreach_ != 0u) {
        const unsigned int first = *h_overreach_ - 1u;
        *h_overreach_ = 0u;
        throw std::runtime_error(
            "MonteCarloBarostat: attempt " + std::to_string(first) + " proposed a box scaled further than the nonbonded potential's neighbor list reaches "
            "(cutoff + padding), so its energy -- and every decision since -- may be wrong.  Moves this large need the reference-shaped attempt: "
            "tm_debug_set_barostat_fast_path(0) / TM_AMD_BAROSTAT_SLOW_PATH=1, a larger nblist_padding, or a smaller volume scale factor");
    }
}

// One attempt on the nonbonded potential's current list (see the comment above k_barostat_propose_probe).  false: the state is
// not the one the fast path needs (nothing has been launched or changed: the caller runs the reference-shaped attempt).
template <typename Real> bool MonteCarloBarostat<Real>::move_on_current_list(double *d_x, double *d_box, hipStream_t stream) {
    if (!g_barostat_fast_path) { // A/B switch (tm_debug_set_barostat_fast_path, TM_AMD_BAROSTAT_SLOW_PATH): always the reference-shaped attempt
        return false;
    }
    // The DUAL launch's filter margin and the scale-aware list test both assume that EVERY atom moves rigidly with its molecule
    // (|s - 1| (|v| + 2 R) bounds the change of a pair distance).  The reference accepts partial group_idxs (mol_utils.cpp checks
    // range and uniqueness only): atoms outside every group keep x while the box and their neighbours move -- up to |s - 1| L of
    // relative displacement.  Those attempts take the reference-shaped path, whose evaluations list x' for themselves.
    if (num_grouped_atoms_ != N_) {
        return false;
    }
    const int n_bps = static_cast<int>(bps_.size());
    plan_.clear();
    for (int i = 0; i < n_bps; i++) {
        bps_[i]->potential->plan_forces(N_, bps_[i]->size, bps_[i]->size > 0 ? bps_[i]->d_p.data : nullptr, plan_);
    }
    plan_.merge_producers(); // an all-pairs potential + an interaction group on its atoms: one carrier (the reference's RBFE states)
    // exactly one potential with a kernel of its own -- an all-pairs nonbonded potential (or a merged carrier) whose sorted
    // pre-gathered state describes (d_x, d_box) -- and everything else in the plan's table, in that potential's precision
    if (plan_.rest().size() != 1) {
        return false;
    }
    const ForcePlan::Rest carrier = plan_.rest()[0];
    NonbondedAllPairsBase *nb = dynamic_cast<NonbondedAllPairsBase *>(carrier.pot);
    if (nb == nullptr || !nb->probe_ready(N_, carrier.P, d_x, carrier.d_p, d_box)) {
        return false;
    }
    const FusedTable *tables[2];
    int blocks[2];
    plan_.prepare_tables(N_, stream, tables, blocks);
    const int prec = nb->precision_bytes() == 8 ? 1 : 0;
    if (tables[prec ^ 1] != nullptr) {
        return false; // terms of the other precision: nobody to carry their table
    }
    const ProbeTarget t = nb->probe_begin();
    const int tpb = DEFAULT_TPB;
    if (max_mol_size_ > BAROSTAT_INLINE_MOL) { // large molecules (a protein): their centroid sums by the segmented-scan kernel, first
        if (!centroids_clean_) {
            HIP_CHECK(hipMemsetAsync(d_centroids_.data, 0, d_centroids_.size(), stream));
        }
        k_barostat_centroids<Real><<<ceil_divide(num_grouped_atoms_, tpb), tpb, 0, stream>>>(num_grouped_atoms_, d_x, d_atom_idxs_.data, d_mol_idxs_.data, d_centroids_.data);
        HIP_CHECK(hipGetLastError());
    }
    const double pressure = static_cast<double>(pressure_) * AVOGADRO * 1e-25; // bar -> kJ/mol/nm^3
    const double kT = BOLTZ_KJ * static_cast<double>(temperature_);
    const i128 *p0 = nullptr, *p1 = nullptr;
    int n0 = 0, n1 = 0;
    const int n_prop_blocks = ceil_divide(std::max(N_, 1), 256);
    d_r2_blocks_.reserve(n_prop_blocks);
#define TM_BAROSTAT_FAST(GREAL)                                                                                        \
    k_barostat_propose_probe<Real, GREAL><<<n_prop_blocks, 256, 0, stream>>>(                                            \
        N_, adaptive_ ? 1 : 0, seed_, attempt_, d_box, d_volume_scale_.data, d_move_.data, d_box_proposed_.data, d_x, d_x_proposed_.data, \
        d_mol_of_atom_.data, d_atom_idxs_.data, d_centroids_.data, d_r2_blocks_.data, h_overreach_, t);                  \
    HIP_CHECK(hipGetLastError());                                                                                      \
    if (g_barostat_dual_launch) {                                                                                      \
        nb->probe_energy_dual(d_box_proposed_.data, tables[prec], blocks[prec], d_x, d_x_proposed_.data, d_r2_blocks_.data, n_prop_blocks, stream, p0, p1, n0); \
        n1 = n0;                                                                                                       \
    } else {                                                                                                           \
        nb->probe_energy(0, d_box, tables[prec], blocks[prec], d_x, stream, p0, n0);                                   \
        nb->probe_energy(1, d_box_proposed_.data, tables[prec], blocks[prec], d_x_proposed_.data, stream, p1, n1);     \
    }                                                                                                                  \
    k_barostat_decide_commit<Real, GREAL><<<ceil_divide(std::max(t.n, 9), 64), 64, 0, stream>>>(                       \
        t.n, adaptive_ ? 1 : 0, num_mols_, kT, pressure, d_move_.data, d_volume_scale_.data, p0, n0, p1, n1, d_box, d_box_proposed_.data, \
        d_x, d_x_proposed_.data, d_counters_.data, d_centroids_.data, num_mols_ * 3, t);                                \
    HIP_CHECK(hipGetLastError())
    if (t.real_bytes == 8) {
        TM_BAROSTAT_FAST(double);
    } else {
        TM_BAROSTAT_FAST(float);
    }
#undef TM_BAROSTAT_FAST
    attempt_++;
    fast_attempts_++;
    centroids_clean_ = true;
    return true;
}

template <typename Real> void MonteCarloBarostat<Real>::reset_counters() { HIP_CHECK(hipMemset(d_counters_.data, 0, 2 * sizeof(int))); }

template <typename Real> void MonteCarloBarostat<Real>::get_counters(int *accepted, int *attempted) {
    int h[2];
    HIP_CHECK(hipDeviceSynchronize());
    d_counters_.copy_to(h);
    this->after_wait();
    *accepted = h[0];
    *attempted = h[1];
}

template <typename Real> double MonteCarloBarostat<Real>::get_volume_scale_factor() {
    double h;
    HIP_CHECK(hipDeviceSynchronize());
    d_volume_scale_.copy_to(&h);
    return h;
}

template <typename Real> void MonteCarloBarostat<Real>::set_volume_scale_factor(const double volume_scale_factor) {
    HIP_CHECK(hipDeviceSynchronize());
    d_volume_scale_.copy_from(&volume_scale_factor);
    this->reset_counters();
}

template <typename Real> void MonteCarloBarostat<Real>::set_pressure(const double pressure) {
    pressure_ = static_cast<Real>(pressure);
    this->reset_counters(); // barostat.cu:249-254
}

template <typename Real> void MonteCarloBarostat<Real>::move(const int N, double *d_x, double *d_box, hipStream_t stream) {
    if (N != N_) {
        throw std::runtime_error("N != N_");
    }
    this->step_++;
    this->acted_ = this->step_ % this->interval_ == 0;
    this->kept_inputs_ = false;
    if (!this->acted_) {
        return;
    }
    if (this->move_on_current_list(d_x, d_box, stream)) {
        this->kept_inputs_ = true; // accepted or not, the potentials' pre-gathered inputs describe (d_x, d_box)
        return;
    }
    centroids_clean_ = false; // (this path's centroid sums stay in the buffer)
    const int tpb = DEFAULT_TPB;
    const int n_prep = std::max(N_, num_mols_) * 3;
    k_barostat_propose<Real><<<ceil_divide(std::max(n_prep, 1), tpb), tpb, 0, stream>>>(
        adaptive_ ? 1 : 0, seed_, attempt_, d_box, d_volume_scale_.data, d_move_.data, d_box_proposed_.data, N_ * 3, d_x,
        d_x_proposed_.data, num_mols_ * 3, d_centroids_.data);
    HIP_CHECK(hipGetLastError());
    attempt_++;
    if (num_grouped_atoms_ > 0) {
        const int blocks = ceil_divide(num_grouped_atoms_, tpb);
        k_barostat_centroids<Real><<<blocks, tpb, 0, stream>>>(num_grouped_atoms_, d_x_proposed_.data, d_atom_idxs_.data, d_mol_idxs_.data, d_centroids_.data);
        k_barostat_rescale<Real><<<blocks, tpb, 0, stream>>>(
            num_grouped_atoms_, d_x_proposed_.data, d_move_.data, d_box, d_atom_idxs_.data, d_mol_idxs_.data, d_mol_offsets_.data,
            d_centroids_.data);
        HIP_CHECK(hipGetLastError());
    }
    const int n_bps = static_cast<int>(bps_.size());
    // every bound potential describes itself to one plan (as for the integrator's force evaluation): all short terms in
    // one energy launch, the nonbonded tile kernel's partial sums left where they are, one reduction for the total
    auto total_energy = [&](const double *x, const double *box, i128 *out) {
        plan_.clear();
        for (int i = 0; i < n_bps; i++) {
            bps_[i]->potential->plan_forces(N_, bps_[i]->size, bps_[i]->size > 0 ? bps_[i]->d_p.data : nullptr, plan_);
        }
        plan_.run_energy(N_, x, box, out, stream);
    };
    total_energy(d_x, d_box, d_u_init_.data);
    total_energy(d_x_proposed_.data, d_box_proposed_.data, d_u_final_.data);
    const double pressure = static_cast<double>(pressure_) * AVOGADRO * 1e-25; // bar -> kJ/mol/nm^3
    const double kT = BOLTZ_KJ * static_cast<double>(temperature_);
    k_barostat_decide<Real><<<ceil_divide(std::max(N_, 9), tpb), tpb, 0, stream>>>(
        N_, adaptive_ ? 1 : 0, num_mols_, kT, pressure, d_move_.data, d_volume_scale_.data, d_u_init_.data, d_u_final_.data, d_box,
        d_box_proposed_.data, d_x, d_x_proposed_.data, d_counters_.data);
    HIP_CHECK(hipGetLastError());
}

template class MonteCarloBarostat<float>;
template class MonteCarloBarostat<double>;

} // namespace tmamd
