// HarmonicBond / HarmonicAngle / PeriodicTorsion host classes (device code: kernels_bonded.hip.hpp).
// reference: cpp/src/harmonic_bond.cu, harmonic_angle.cu, periodic_torsion.cu
#include "kernels_bonded.hip.hpp"

namespace tmamd {

// ---- host classes ------------------------------------------------------------------------------------------

template <typename Real> HarmonicBond<Real>::HarmonicBond(const std::vector<int> &bond_idxs) : B_(bond_idxs.size() / 2) {
    if (bond_idxs.size() % 2 != 0) {
        throw std::runtime_error("bond_idxs.size() must be exactly 2*k!");
    }
    for (int b = 0; b < B_; b++) {
        if (bond_idxs[b * 2 + 0] == bond_idxs[b * 2 + 1]) {
            throw std::runtime_error("src == dst");
        }
    }
    d_idxs_.realloc(B_ * 2);
    if (B_ > 0) {
        d_idxs_.copy_from(bond_idxs.data());
    }
    this->note_term_atoms(bond_idxs);
    d_u_partials_.realloc(ceil_divide(B_, 256) * 4 + 1);
}

template <typename Real> void HarmonicBond<Real>::plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) {
    if (P != 2 * B_) {
        throw std::runtime_error(
            "HarmonicBond::execute_device(): expected P == 2*B, got P=" + std::to_string(P) + ", 2*B=" + std::to_string(2 * B_));
    }
    if (B_ > 0) {
        plan.add_segment(sizeof(Real), FusedSegment{FUSED_BOND, B_, d_idxs_.data, d_p, nullptr, 0.0, 0.0}, this, P, d_p);
    }
}

template <typename Real>
void HarmonicBond<Real>::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    if (P != 2 * B_) {
        throw std::runtime_error(
            "HarmonicBond::execute_device(): expected P == 2*B, got P=" + std::to_string(P) + ", 2*B=" + std::to_string(2 * B_));
    }
    if (B_ > 0) {
        const int blocks = ceil_divide(B_, 256);
        k_harmonic_bond<Real><<<blocks, 256, 0, stream>>>(B_, d_x, d_p, d_idxs_.data, d_du_dx, d_du_dp, d_u ? d_u_partials_.data : nullptr);
        HIP_CHECK(hipGetLastError());
        if (d_u)
            reduce_i128_device(d_u_partials_.data, blocks * 4, d_u, stream);
    }
}

template <typename Real> HarmonicAngle<Real>::HarmonicAngle(const std::vector<int> &angle_idxs) : A_(angle_idxs.size() / 3) {
    if (angle_idxs.size() % 3 != 0) {
        throw std::runtime_error("angle_idxs.size() must be exactly 3*A");
    }
    for (int a = 0; a < A_; a++) {
        const int i = angle_idxs[a * 3 + 0], j = angle_idxs[a * 3 + 1], k = angle_idxs[a * 3 + 2];
        if (i == j || j == k || i == k) {
            throw std::runtime_error("angle triplets must be unique");
        }
    }
    d_idxs_.realloc(A_ * 3);
    if (A_ > 0) {
        d_idxs_.copy_from(angle_idxs.data());
    }
    this->note_term_atoms(angle_idxs);
    d_u_partials_.realloc(ceil_divide(A_, 256) * 4 + 1);
}

template <typename Real> void HarmonicAngle<Real>::plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) {
    if (P != 3 * A_) {
        throw std::runtime_error(
            "HarmonicAngle::execute_device(): expected P == 3*A, got P=" + std::to_string(P) + ", 3*A=" + std::to_string(3 * A_));
    }
    if (A_ > 0) {
        plan.add_segment(sizeof(Real), FusedSegment{FUSED_ANGLE, A_, d_idxs_.data, d_p, nullptr, 0.0, 0.0}, this, P, d_p);
    }
}

template <typename Real>
void HarmonicAngle<Real>::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    if (P != 3 * A_) {
        throw std::runtime_error(
            "HarmonicAngle::execute_device(): expected P == 3*A, got P=" + std::to_string(P) + ", 3*A=" + std::to_string(3 * A_));
    }
    if (A_ > 0) {
        const int blocks = ceil_divide(A_, 256);
        k_harmonic_angle<Real><<<blocks, 256, 0, stream>>>(A_, d_x, d_p, d_idxs_.data, d_du_dx, d_du_dp, d_u ? d_u_partials_.data : nullptr);
        HIP_CHECK(hipGetLastError());
        if (d_u)
            reduce_i128_device(d_u_partials_.data, blocks * 4, d_u, stream);
    }
}

template <typename Real> PeriodicTorsion<Real>::PeriodicTorsion(const std::vector<int> &torsion_idxs) : T_(torsion_idxs.size() / 4) {
    if (torsion_idxs.size() % 4 != 0) {
        throw std::runtime_error("torsion_idxs.size() must be exactly 4*k");
    }
    for (int a = 0; a < T_; a++) {
        const int i = torsion_idxs[a * 4 + 0], j = torsion_idxs[a * 4 + 1], k = torsion_idxs[a * 4 + 2], l = torsion_idxs[a * 4 + 3];
        if (i == j || i == k || i == l || j == k || j == l || k == l) {
            throw std::runtime_error("torsion quads must be unique");
        }
    }
    d_idxs_.realloc(T_ * 4);
    if (T_ > 0) {
        d_idxs_.copy_from(torsion_idxs.data());
    }
    this->note_term_atoms(torsion_idxs);
    d_u_partials_.realloc(ceil_divide(T_, 256) * 4 + 1);
}

template <typename Real> void PeriodicTorsion<Real>::plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) {
    if (P != 3 * T_) {
        throw std::runtime_error(
            "PeriodicTorsion::execute_device(): expected P == 3*T, got P=" + std::to_string(P) + ", 3*T=" + std::to_string(3 * T_));
    }
    if (T_ > 0) {
        plan.add_segment(sizeof(Real), FusedSegment{FUSED_TORSION, T_, d_idxs_.data, d_p, nullptr, 0.0, 0.0}, this, P, d_p);
    }
}

template <typename Real>
void PeriodicTorsion<Real>::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    if (P != 3 * T_) {
        throw std::runtime_error(
            "PeriodicTorsion::execute_device(): expected P == 3*T, got P=" + std::to_string(P) + ", 3*T=" + std::to_string(3 * T_));
    }
    if (T_ > 0) {
        const int blocks = ceil_divide(T_, 256);
        k_periodic_torsion<Real><<<blocks, 256, 0, stream>>>(T_, d_x, d_p, d_idxs_.data, d_du_dx, d_du_dp, d_u ? d_u_partials_.data : nullptr);
        HIP_CHECK(hipGetLastError());
        if (d_u)
            reduce_i128_device(d_u_partials_.data, blocks * 4, d_u, stream);
    }
}

// ---- chiral restraints ---------------------------------------------------------------------------------------
template <typename Real> ChiralAtomRestraint<Real>::ChiralAtomRestraint(const std::vector<int> &idxs) : R_(idxs.size() / 4) {
    if (idxs.size() % 4 != 0) {
        throw std::runtime_error("idxs.size() must be exactly 4*k!");
    }
    d_idxs_.realloc(R_ * 4);
    if (R_ > 0) {
        d_idxs_.copy_from(idxs.data());
    }
    this->note_term_atoms(idxs);
    d_u_partials_.realloc(ceil_divide(R_, 256) * 4 + 1);
}

template <typename Real> void ChiralAtomRestraint<Real>::check_size(const int P) const {
    if (P != R_) {
        throw std::runtime_error(
            "ChiralAtomRestraint::execute_device(): expected P == R, got P=" + std::to_string(P) + ", R=" + std::to_string(R_));
    }
}

template <typename Real> void ChiralAtomRestraint<Real>::plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) {
    this->check_size(P);
    if (R_ > 0) {
        plan.add_segment(sizeof(Real), FusedSegment{FUSED_CHIRAL_ATOM, R_, d_idxs_.data, d_p, nullptr, 0.0, 0.0, nullptr}, this, P, d_p);
    }
}

template <typename Real>
void ChiralAtomRestraint<Real>::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    this->check_size(P);
    if (R_ > 0) {
        const int blocks = ceil_divide(R_, 256);
        k_chiral_atom_restraint<Real><<<blocks, 256, 0, stream>>>(R_, d_x, d_p, d_idxs_.data, d_du_dx, d_du_dp, d_u ? d_u_partials_.data : nullptr);
        HIP_CHECK(hipGetLastError());
        if (d_u)
            reduce_i128_device(d_u_partials_.data, blocks * 4, d_u, stream);
    }
}

template <typename Real>
ChiralBondRestraint<Real>::ChiralBondRestraint(const std::vector<int> &idxs, const std::vector<int> &signs) : R_(idxs.size() / 4) {
    if (idxs.size() % 4 != 0) {
        throw std::runtime_error("idxs.size() must be exactly 4*R!");
    }
    if (static_cast<size_t>(R_) != signs.size()) {
        throw std::runtime_error("signs.size() must be exactly R!");
    }
    for (auto sgn : signs) {
        if (sgn != -1 && sgn != 1) {
            throw std::runtime_error("signs must be comprised exclusively of 1 or -1");
        }
    }
    d_idxs_.realloc(R_ * 4);
    d_signs_.realloc(R_);
    if (R_ > 0) {
        d_idxs_.copy_from(idxs.data());
        this->note_term_atoms(idxs);
        d_signs_.copy_from(signs.data());
    }
    d_u_partials_.realloc(ceil_divide(R_, 256) * 4 + 1);
}

template <typename Real> void ChiralBondRestraint<Real>::check_size(const int P) const {
    if (P != R_) {
        throw std::runtime_error(
            "ChiralBondRestraint::execute_device(): expected P == R, got P=" + std::to_string(P) + ", R=" + std::to_string(R_));
    }
}

template <typename Real> void ChiralBondRestraint<Real>::plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) {
    this->check_size(P);
    if (R_ > 0) {
        plan.add_segment(sizeof(Real), FusedSegment{FUSED_CHIRAL_BOND, R_, d_idxs_.data, d_p, nullptr, 0.0, 0.0, d_signs_.data}, this, P, d_p);
    }
}

template <typename Real>
void ChiralBondRestraint<Real>::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    this->check_size(P);
    if (R_ > 0) {
        const int blocks = ceil_divide(R_, 256);
        k_chiral_bond_restraint<Real><<<blocks, 256, 0, stream>>>(
            R_, d_x, d_p, d_idxs_.data, d_signs_.data, d_du_dx, d_du_dp, d_u ? d_u_partials_.data : nullptr);
        HIP_CHECK(hipGetLastError());
        if (d_u)
            reduce_i128_device(d_u_partials_.data, blocks * 4, d_u, stream);
    }
}

// ---- flat-bottom restraints, centroid restraint -----------------------------------------------------------------
template <typename Real, bool Log>
FlatBottomBond<Real, Log>::FlatBottomBond(const std::vector<int> &bond_idxs, const double beta) : B_(bond_idxs.size() / 2), beta_(beta) {
    if (Log && beta <= 0) {
        throw std::runtime_error("beta must be positive");
    }
    if (bond_idxs.size() % 2 != 0) {
        throw std::runtime_error("bond_idxs.size() must be exactly 2*k!");
    }
    for (int b = 0; b < B_; b++) {
        const int src = bond_idxs[b * 2 + 0], dst = bond_idxs[b * 2 + 1];
        if (src == dst) {
            throw std::runtime_error("src == dst");
        }
        if (src < 0 || dst < 0) {
            throw std::runtime_error("idxs must be non-negative");
        }
    }
    d_idxs_.realloc(B_ * 2);
    if (B_ > 0) {
        d_idxs_.copy_from(bond_idxs.data());
    }
    this->note_term_atoms(bond_idxs);
    d_u_partials_.realloc(ceil_divide(B_, 256) * 4 + 1);
}

template <typename Real, bool Log> void FlatBottomBond<Real, Log>::set_bonds(const std::vector<int> &bond_idxs) {
    if (bond_idxs.size() % 2 != 0) {
        throw std::runtime_error("bond_idxs.size() must be exactly 2*k!");
    }
    HIP_CHECK(hipDeviceSynchronize()); // a launch still reading the old list must finish before it is replaced
    B_ = static_cast<int>(bond_idxs.size() / 2);
    d_idxs_.reserve(static_cast<size_t>(B_) * 2);
    if (B_ > 0) {
        d_idxs_.copy_from(bond_idxs.data(), static_cast<size_t>(B_) * 2);
    }
    d_u_partials_.reserve(ceil_divide(B_, 256) * 4 + 1);
}

template <typename Real, bool Log> void FlatBottomBond<Real, Log>::check_size(const int P) const {
    if (P != 3 * B_) {
        throw std::runtime_error(
            std::string(Log ? "LogFlatBottomBond" : "FlatBottomBond") + "::execute_device(): expected P == 3*B, got P=" +
            std::to_string(P) + ", 3*B=" + std::to_string(3 * B_));
    }
}

template <typename Real, bool Log> void FlatBottomBond<Real, Log>::plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) {
    this->check_size(P);
    if (B_ > 0) {
        plan.add_segment(
            sizeof(Real), FusedSegment{Log ? FUSED_LOG_FLAT_BOTTOM_BOND : FUSED_FLAT_BOTTOM_BOND, B_, d_idxs_.data, d_p, nullptr, beta_, 0.0, nullptr},
            this, P, d_p);
    }
}

template <typename Real, bool Log>
void FlatBottomBond<Real, Log>::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    this->check_size(P);
    if (B_ > 0) {
        const int blocks = ceil_divide(B_, 256);
        k_flat_bottom_bond<Real, Log><<<blocks, 256, 0, stream>>>(
            B_, d_x, d_box, d_p, d_idxs_.data, beta_, d_du_dx, d_du_dp, d_u ? d_u_partials_.data : nullptr);
        HIP_CHECK(hipGetLastError());
        if (d_u)
            reduce_i128_device(d_u_partials_.data, blocks * 4, d_u, stream);
    } else if (d_u) {
        // no bonds (local MD starts both of its restraints that way): the contract is that d_u is OVERWRITTEN
        HIP_CHECK(hipMemsetAsync(d_u, 0, sizeof(i128), stream));
    }
}

template <typename Real>
CentroidRestraint<Real>::CentroidRestraint(
    const std::vector<int> &group_a_idxs, const std::vector<int> &group_b_idxs, const double kb, const double b0)
    : NA_(group_a_idxs.size()), NB_(group_b_idxs.size()), kb_(kb), b0_(b0) {
    d_a_.realloc(NA_);
    d_b_.realloc(NB_);
    if (NA_ > 0)
        d_a_.copy_from(group_a_idxs.data());
    if (NB_ > 0)
        d_b_.copy_from(group_b_idxs.data());
    d_sums_.realloc(6);
}

template <typename Real>
void CentroidRestraint<Real>::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    if (NA_ + NB_ > 0) { // (no parameters: d_du_dp is left untouched, as in the reference)
        const int blocks = ceil_divide(NA_ + NB_, 256);
        d_sums_.zero_async(stream, 6);
        k_centroid_sums<Real><<<blocks, 256, 0, stream>>>(NA_, NB_, d_x, d_a_.data, d_b_.data, d_sums_.data);
        k_centroid_restraint<Real><<<blocks, 256, 0, stream>>>(NA_, NB_, d_a_.data, d_b_.data, d_sums_.data, kb_, b0_, d_du_dx, d_u);
        HIP_CHECK(hipGetLastError());
    }
}

template class FlatBottomBond<float, false>;
template class FlatBottomBond<double, false>;
template class FlatBottomBond<float, true>;
template class FlatBottomBond<double, true>;
template class CentroidRestraint<float>;
template class CentroidRestraint<double>;
template class ChiralAtomRestraint<float>;
template class ChiralAtomRestraint<double>;
template class ChiralBondRestraint<float>;
template class ChiralBondRestraint<double>;
template class HarmonicBond<float>;
template class HarmonicBond<double>;
template class HarmonicAngle<float>;
template class HarmonicAngle<double>;
template class PeriodicTorsion<float>;
template class PeriodicTorsion<double>;

} // namespace tmamd
