// MonteCarloBarostat for gfx950.  reference: cpp/src/barostat.cu:19-259, kernels/k_barostat.cuh:10-189,
// mol_utils.cpp:8-85 (group validation / flattening), mover.cu:7-23.
//
// One attempt = memcpy x -> x_proposed, four small kernels, two energy-only evaluations of every bound potential and
// two 128-bit reductions, all stream-ordered; no host synchronisation.  Differences from the reference, on purpose:
// the two uniforms come from a counter-based Philox4x32-10 keyed on (seed; attempt) inside the kernels instead of a
// cuRAND batch buffer (statistically equivalent, reproducible per seed on this implementation only).
#include "engine.hpp"
#include "fixed_point.hip.hpp"
#include "philox.hip.hpp"

#include <algorithm>
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
}

template <typename Real> void MonteCarloBarostat<Real>::reset_counters() { HIP_CHECK(hipMemset(d_counters_.data, 0, 2 * sizeof(int))); }

template <typename Real> void MonteCarloBarostat<Real>::get_counters(int *accepted, int *attempted) {
    int h[2];
    HIP_CHECK(hipDeviceSynchronize());
    d_counters_.copy_to(h);
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
    if (!this->acted_) {
        return;
    }
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
