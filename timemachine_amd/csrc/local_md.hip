// Local MD: the driver that lets Context move a neighbourhood of the system while everything else stands still.
// reference: cpp/src/local_md_potentials.cu:27-338, local_md_utils.cu:13-137, kernels/k_local_md.cuh:5-47,
// kernels/k_flat_bottom_bond.cuh:22-80 (k_log_probability_selection), context.cu:90-213.
// See LocalMDPotentials in engine.hpp for how the work is split between one selection kernel and the host.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <sstream>

#include "engine.hpp"
#include "philox.hip.hpp"

namespace tmamd {

static const double BOLTZ = 0.008314462618; // kJ/mol/K, cpp/src/constants.hpp:5

// The uniform a free-particle selection compares atom `atom`'s acceptance probability with: (0, 1], one Philox block per
// atom keyed on the call's seed.  (The reference draws from cuRAND XORWOW seeded the same way -- third party, unpinned;
// oracle/local_md.py restates this generator so tests can predict the selection.)
__device__ __forceinline__ float local_md_uniform(const unsigned int seed, const unsigned int atom) {
    unsigned int r[4];
    philox4x32_10(atom, 0u, 0x4c4f4341u, 0x4c4d4421u, seed, 0x53454c45u, r);
    return (static_cast<float>(r[0] >> 8) + 1.0f) * 5.9604644775390625e-08f; // (r >> 8 + 1) / 2^24: exact in float
}

// One thread per atom: free (its own index) or frozen (N).  Arithmetic as k_log_probability_selection<float>:
// displacement to the reference in f64, imaged and squared in f32, U = k/4 (r - radius)^4 beyond the radius,
// p = exp(-U / kT) with the division and the exponential in f64, rounded to f32 for the comparison p >= u.
__global__ __launch_bounds__(256) void k_local_md_select(
    const int N, const double kBT, const float radius, const float k, const unsigned int reference_idx, const unsigned int seed,
    const double *__restrict__ coords, const double *__restrict__ box, unsigned int *__restrict__ selected) {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= static_cast<unsigned int>(N)) {
        return;
    }
    const float radius_sq = radius * radius;
    float d2 = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float b = static_cast<float>(box[d * 3 + d]);
        const float inv_b = 1 / b;
        float delta = static_cast<float>(coords[idx * 3 + d] - coords[reference_idx * 3 + d]);
        delta -= b * nearbyintf(delta * inv_b);
        d2 += delta * delta;
    }
    float prob = 1.0f;
    if (d2 >= radius_sq) {
        const float dr = sqrtf(d2) - radius;
        const float dr2 = dr * dr;
        const float energy = (k / 4.0f) * (dr2 * dr2);
        prob = static_cast<float>(exp(-static_cast<double>(energy) / kBT));
    }
    // the reference atom itself is never "selected": frozen unless the caller asked for a free reference
    selected[idx] = (idx != reference_idx && prob >= local_md_uniform(seed, idx)) ? idx : static_cast<unsigned int>(N);
}

void verify_local_md_parameters(const double radius, const double k) {
    // reference: local_md_utils.cu:114-135 (messages matched by tests/test_md.py:306-313)
    const double min_radius = 0.1;
    if (radius < min_radius) {
        throw std::runtime_error("radius must be greater or equal to " + std::to_string(min_radius));
    }
    if (k < 1.0) {
        throw std::runtime_error("k must be at least one");
    }
    const double max_k = 1e6;
    if (k > max_k) {
        std::ostringstream oss;
        oss << "k must be less than than " << max_k;
        throw std::runtime_error(oss.str());
    }
}

namespace {

struct FoundAllPairs {
    std::shared_ptr<NonbondedAllPairsBase> pot;
    std::vector<double> params;
};

// every NonbondedAllPairs (not interaction groups) reachable through Summed / Fanout wrappers, with its parameter slice
// (reference: get_nonbonded_all_pair_potentials, nonbonded_common.cpp:77-124)
void find_all_pairs(const std::shared_ptr<Potential> &pot, const std::vector<double> &params, std::vector<FoundAllPairs> &out) {
    if (auto nb = std::dynamic_pointer_cast<NonbondedAllPairsBase>(pot)) {
        if (!nb->is_interaction_group()) {
            out.push_back({nb, params});
        }
    } else if (auto f = std::dynamic_pointer_cast<FanoutSummedPotential>(pot)) {
        for (auto &c : f->get_potentials()) {
            find_all_pairs(c, params, out);
        }
    } else if (auto s = std::dynamic_pointer_cast<SummedPotential>(pot)) {
        size_t offset = 0;
        const std::vector<int> &sizes = s->get_parameter_sizes();
        const auto &children = s->get_potentials();
        for (size_t i = 0; i < children.size(); i++) {
            const std::vector<double> slice(params.begin() + offset, params.begin() + offset + sizes[i]);
            find_all_pairs(children[i], slice, out);
            offset += sizes[i];
        }
    }
}

} // namespace

LocalMDPotentials::LocalMDPotentials(
    const int N, const std::vector<std::shared_ptr<BoundPotential>> &bps, const bool freeze_reference, const double temperature)
    : freeze_reference(freeze_reference), temperature(temperature), N_(N), all_potentials_(bps), d_free_idxs_(N), h_free_(N) {
    if (temperature <= 0.0) {
        throw std::runtime_error("temperature must be greater than 0");
    }
    std::vector<FoundAllPairs> found;
    for (auto &bp : bps) {
        std::vector<double> h_params(bp->size);
        if (bp->size > 0) {
            bp->d_p.copy_to(h_params.data(), bp->size);
        }
        find_all_pairs(bp->potential, h_params, found);
    }
    if (found.size() > 1) {
        throw std::runtime_error("found multiple NonbondedAllPairs potentials");
    }
    if (found.size() != 1) {
        throw std::runtime_error("unable to find a NonbondedAllPairs potential");
    }
    if (N < 2) {
        throw std::runtime_error("N must be greater than 1");
    }
    all_pairs_ = found[0].pot;
    all_pairs_idxs_ = all_pairs_->current_atom_idxs();
    in_all_pairs_.assign(N, 0);
    for (int a : all_pairs_idxs_) {
        in_all_pairs_[a] = 1;
    }
    // restraints start out with no bonds; every setup re-targets them
    free_restraint_.reset(new FlatBottomBond<float, false>(std::vector<int>(), 0.0));
    // parameter buffers sized for the largest possible restraint list once (reference: N default bonds, :56-64)
    bound_free_restraint_.reset(new BoundPotential(free_restraint_, std::vector<double>(static_cast<size_t>(N) * 3)));
    // free x frozen interactions: an interaction group with the all-pairs potential's parameters, precision, beta, cutoff
    // (placeholder groups until the first setup; reference: construct_ixn_group_potential, local_md_utils.cu:75-104)
    const std::vector<int> row_dummy{0}, col_dummy{1};
    std::shared_ptr<Potential> ixn;
    if (all_pairs_->precision_bytes() == 8) {
        ixn.reset(new NonbondedInteractionGroup<double>(
            N, row_dummy, col_dummy, all_pairs_->get_beta(), all_pairs_->get_cutoff(), false, all_pairs_->get_nblist_padding()));
    } else {
        ixn.reset(new NonbondedInteractionGroup<float>(
            N, row_dummy, col_dummy, all_pairs_->get_beta(), all_pairs_->get_cutoff(), false, all_pairs_->get_nblist_padding()));
    }
    ixn_group_ = std::dynamic_pointer_cast<NonbondedAllPairsBase>(ixn);
    all_potentials_.push_back(bound_free_restraint_);
    all_potentials_.push_back(std::shared_ptr<BoundPotential>(new BoundPotential(ixn, found[0].params)));
    if (!freeze_reference) {
        frozen_restraint_.reset(new FlatBottomBond<float, true>(std::vector<int>(), 1.0 / (temperature * BOLTZ)));
        bound_frozen_restraint_.reset(new BoundPotential(frozen_restraint_, std::vector<double>(static_cast<size_t>(N) * 3)));
        all_potentials_.push_back(bound_frozen_restraint_);
    }
}

void LocalMDPotentials::setup_from_idxs(
    const double *d_x_t, const double *d_box_t, const std::vector<int> &local_idxs, const int seed, const double radius,
    const double k, hipStream_t stream) {
    // the reference atom: same generator, same distribution object as the reference (local_md_potentials.cu:128-132)
    std::mt19937 rng;
    rng.seed(seed);
    std::uniform_int_distribution<unsigned int> random_dist(0, local_idxs.size() - 1);
    const unsigned int reference_idx = local_idxs[random_dist(rng)];
    const double kBT = BOLTZ * temperature;
    k_local_md_select<<<ceil_divide(N_, 256), 256, 0, stream>>>(
        N_, kBT, static_cast<float>(radius), static_cast<float>(k), reference_idx, static_cast<unsigned int>(seed), d_x_t, d_box_t,
        d_free_idxs_.data);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(h_free_.data(), d_free_idxs_.data, static_cast<size_t>(N_) * sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    this->setup_given_free_flags(static_cast<int>(reference_idx), radius, k, stream);
}

void LocalMDPotentials::setup_from_selection(
    const int reference_idx, const std::vector<int> &selection_idxs, const double radius, const double k, hipStream_t stream) {
    std::fill(h_free_.begin(), h_free_.end(), static_cast<unsigned int>(N_));
    for (int a : selection_idxs) {
        h_free_[a] = static_cast<unsigned int>(a);
    }
    this->setup_given_free_flags(reference_idx, radius, k, stream);
}

void LocalMDPotentials::setup_given_free_flags(const int reference_idx, const double radius, const double k, hipStream_t stream) {
    last_reference_ = reference_idx;
    if (!freeze_reference) {
        h_free_[reference_idx] = static_cast<unsigned int>(reference_idx); // moves with the free atoms
    }
    // rows: free atoms the all-pairs potential covers; columns: frozen atoms it covers (reference:
    // local_md_potentials.cu:198-301 -- intersections + partitions on the device)
    std::vector<int> rows, cols;
    for (int i = 0; i < N_; i++) {
        if (!in_all_pairs_[i]) {
            continue;
        }
        (h_free_[i] < static_cast<unsigned int>(N_) ? rows : cols).push_back(i);
    }
    const int num_row_idxs = static_cast<int>(rows.size());
    if (num_row_idxs == 0) {
        throw std::runtime_error("LocalMDPotentials setup has no free particles selected");
    }
    if (num_row_idxs == N_ - 1 || (!freeze_reference && num_row_idxs == N_)) {
        fprintf(stderr, "LocalMDPotentials setup has entire system selected\n");
    }
    // what the integrator walks: every atom flagged free, covered by the all-pairs potential or not
    HIP_CHECK(hipMemcpyAsync(d_free_idxs_.data, h_free_.data(), static_cast<size_t>(N_) * sizeof(unsigned int), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));

    all_pairs_->narrow_to(rows);
    if (auto g = std::dynamic_pointer_cast<NonbondedInteractionGroup<double>>(ixn_group_)) {
        g->set_atom_idxs(rows, cols);
    } else {
        std::dynamic_pointer_cast<NonbondedInteractionGroup<float>>(ixn_group_)->set_atom_idxs(rows, cols);
    }
    // flat-bottom restraint (k, r_min = 0, r_max = radius) from the reference atom to every free atom.  A free reference is
    // itself a row: the reference's kernel then evaluates a bond of the atom with itself (0 * 0 / 0, saved by the two
    // contributions wrapping to zero in fixed point); here that pair is simply not listed.
    auto restrain = [&](const std::vector<int> &atoms, std::vector<int> &bonds, std::vector<double> &params) {
        if (std::isinf(radius)) {
            return; // "the entire system" (tests/test_md.py:546-583): a restraint that never acts; (r - inf)^3 * 0 is not evaluated
        }
        for (int a : atoms) {
            if (a == reference_idx) {
                continue;
            }
            bonds.push_back(reference_idx);
            bonds.push_back(a);
            params.push_back(k);
            params.push_back(0.0);
            params.push_back(radius);
        }
    };
    std::vector<int> bonds;
    std::vector<double> params;
    restrain(rows, bonds, params);
    free_restraint_->set_bonds(bonds);
    bound_free_restraint_->set_params_prefix(params);
    if (!freeze_reference) {
        // the moving reference stays tied to the frozen atoms it was selected among (log flat-bottom restraint)
        bonds.clear();
        params.clear();
        restrain(cols, bonds, params);
        frozen_restraint_->set_bonds(bonds);
        bound_frozen_restraint_->set_params_prefix(params);
    }
}

void LocalMDPotentials::reset_potentials() { all_pairs_->narrow_to(all_pairs_idxs_); }

} // namespace tmamd
