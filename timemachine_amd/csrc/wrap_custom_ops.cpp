// custom_ops -- the compiled pybind11 module `timemachine_amd.lib.custom_ops`.
//
// The reference's boundary is PYBIND11_MODULE(custom_ops, m) in timemachine/cpp/src/wrap_kernels.cpp:2143-2312, which binds
// its C++/CUDA classes directly.  This module offers the same Python surface for the force-evaluation + Langevin-step hot
// path, but is a HOST-ONLY translation unit (g++, no HIP): every method is a thin door onto the C ABI of
// libtimemachine_amd.so (include/timemachine_amd.h), which owns all device code.  What lives here is exactly what the
// reference keeps in its binding layer: argument conversion (py::array_t<T, c_style> WITHOUT forcecast -- safe casts only,
// wrap_kernels.cpp:51-78 and every constructor), the shape / box validation with the reference's messages, allocation of the
// returned arrays, fixed-point -> float conversion of the returned accumulators, object lifetimes (std::shared_ptr holders:
// a BoundPotential keeps its Potential, a Context its integrator / potentials / movers), and exception translation
// (std::runtime_error -> RuntimeError, InvalidHardware).  The GIL is released around every call that reaches the device.
//
// Built by timemachine_amd/csrc/build.py into timemachine_amd/lib/custom_ops<ext-suffix>.so, linked against
// libtimemachine_amd.so (RUNPATH $ORIGIN/../csrc).  There is no CPU fallback: without a GPU every method raises
// InvalidHardware, and without the library the import fails.
#include <pybind11/eval.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <climits>
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/timemachine_amd.h"

namespace py = pybind11;

namespace {

using arr_d = py::array_t<double, py::array::c_style>;
using arr_i = py::array_t<int32_t, py::array::c_style>;
using arr_u = py::array_t<uint32_t, py::array::c_style>;
using arr_u64 = py::array_t<uint64_t, py::array::c_style>;
using arr_i64 = py::array_t<int64_t, py::array::c_style>;

// reference: cpp/src/exceptions.hpp (InvalidHardware), registered at wrap_kernels.cpp:2311
struct InvalidHardwareError : public std::runtime_error {
    using std::runtime_error::runtime_error;
};

void check(const int code) {
    if (code == TM_OK) {
        return;
    }
    const std::string msg = tm_last_error();
    if (code == TM_ERR_INVALID_HARDWARE) {
        throw InvalidHardwareError(msg);
    }
    throw std::runtime_error(msg);
}

// wrap_kernels.cpp:51-59
void verify_coords(const arr_d &coords) {
    if (coords.ndim() != 2) {
        throw std::runtime_error("coords dimensions must be 2");
    }
    if (coords.shape(1) != 3) {
        throw std::runtime_error("coords must have a shape that is 3 dimensional");
    }
}

// wrap_kernels.cpp:62-78
void verify_coords_and_box(const arr_d &coords, const arr_d &box) {
    verify_coords(coords);
    if (box.ndim() != 2 || box.shape(0) != 3 || box.shape(1) != 3) {
        throw std::runtime_error("box must be 3x3");
    }
    const double *b = box.data();
    for (int i = 0; i < 9; i++) {
        if (i % 4 == 0) {
            if (b[i] <= 0.0) {
                throw std::runtime_error("box must have positive values along diagonal");
            }
        } else if (b[i] != 0.0) {
            throw std::runtime_error("box must be ortholinear");
        }
    }
}

// FIXED_TO_FLOAT<double> over an accumulator array (cpp/src/fixed_point.hpp:18-20)
arr_d fixed_to_float_array(const std::vector<uint64_t> &v, const std::vector<py::ssize_t> &shape) {
    arr_d out(shape);
    double *o = out.mutable_data();
    for (size_t i = 0; i < v.size(); i++) {
        o[i] = tm_fixed_to_float(v[i]);
    }
    return out;
}

py::object int128_to_python(const tm_int128 &e) {
    py::int_ hi(static_cast<long long>(e.hi)), lo(static_cast<unsigned long long>(e.lo));
    return hi.attr("__lshift__")(64).attr("__or__")(lo);
}

std::vector<py::ssize_t> shape_of(const py::array &a) { return std::vector<py::ssize_t>(a.shape(), a.shape() + a.ndim()); }

// ---- holders ---------------------------------------------------------------------------------------------------------
struct PyPotential {
    tm_potential_t h = nullptr;
    std::vector<std::shared_ptr<PyPotential>> children; // Summed / Fanout: the Python-visible children (same objects back)
    PyPotential() = default;
    PyPotential(const PyPotential &) = delete;
    PyPotential &operator=(const PyPotential &) = delete;
    virtual ~PyPotential() {
        if (h) {
            tm_potential_destroy(h);
        }
    }
};
enum PotKind {
    HarmonicBond, HarmonicAngle, PeriodicTorsion, NonbondedAllPairs, NonbondedPairList, NonbondedExclusions, NonbondedInteractionGroup,
    NonbondedPairListPrecomputed, FlatBottomBond, LogFlatBottomBond, CentroidRestraint, ChiralAtomRestraint, ChiralBondRestraint,
};
template <PotKind KIND, int PREC> struct Pot : PyPotential {};
struct PySummed : PyPotential {};
struct PyFanout : PyPotential {};

struct PyBound {
    tm_bound_potential_t h = nullptr;
    std::shared_ptr<PyPotential> potential; // tests/test_potentials.py:36-48: the bound potential keeps it alive
    ~PyBound() {
        if (h) {
            tm_bound_potential_destroy(h);
        }
    }
};

struct PyIntegrator {
    tm_integrator_t h = nullptr;
    virtual ~PyIntegrator() {
        if (h) {
            tm_integrator_destroy(h);
        }
    }
};
struct PyLangevin : PyIntegrator {};
struct PyVerlet : PyIntegrator {};

struct PyMover {
    tm_mover_t h = nullptr;
    virtual ~PyMover() {
        if (h) {
            tm_mover_destroy(h);
        }
    }
};
struct PyBarostat : PyMover {
    std::vector<std::shared_ptr<PyBound>> bps; // the barostat evaluates them
};

struct PyContext {
    tm_context_t h = nullptr;
    int N = 0;
    std::shared_ptr<PyIntegrator> integrator;
    std::vector<std::shared_ptr<PyBound>> bps;
    std::vector<std::shared_ptr<PyMover>> movers;
    ~PyContext() {
        if (h) {
            tm_context_destroy(h);
        }
    }
};

template <int PREC> struct PyNeighborlist {
    tm_neighborlist_t h = nullptr;
    ~PyNeighborlist() {
        if (h) {
            tm_neighborlist_destroy(h);
        }
    }
};

struct PyHilbertSort {
    tm_hilbert_sort_t h = nullptr;
    ~PyHilbertSort() {
        if (h) {
            tm_hilbert_sort_destroy(h);
        }
    }
};

std::vector<tm_potential_t> handles_of(const std::vector<std::shared_ptr<PyPotential>> &pots) {
    std::vector<tm_potential_t> hs;
    for (const auto &p : pots) {
        if (!p) {
            throw std::runtime_error("got nullptr instead of potential");
        }
        hs.push_back(p->h);
    }
    return hs;
}

std::vector<tm_bound_potential_t> handles_of(const std::vector<std::shared_ptr<PyBound>> &bps) {
    std::vector<tm_bound_potential_t> hs;
    for (const auto &b : bps) {
        if (!b) {
            throw std::runtime_error("got nullptr instead of bound potential");
        }
        hs.push_back(b->h);
    }
    return hs;
}

// per-column exponents for nonbonded terms, slices for Summed: the virtual Potential::du_dp_fixed_to_float
void du_dp_to_float(PyPotential &p, const int N, const int P, const uint64_t *fixed, double *out) {
    check(tm_potential_du_dp_fixed_to_float(p.h, N, P, fixed, out));
}

// ---- Potential (wrap_kernels.cpp:731-1131) ------------------------------------------------------------------------------
py::tuple potential_execute(PyPotential &pot, const arr_d &coords, const arr_d &params, const arr_d &box, const bool want_dx, const bool want_dp,
                            const bool want_u) {
    verify_coords_and_box(coords, box);
    const int N = static_cast<int>(coords.shape(0)), P = static_cast<int>(params.size());
    // (the conversions of wrap_kernels.cpp:1066-1101 happen on the device, behind tm_potential_execute_f64: same values)
    arr_d dx(std::vector<py::ssize_t>{want_dx ? N : 0, 3}), dp(want_dp ? shape_of(params) : std::vector<py::ssize_t>{0});
    double u = 0.0;
    double *p_dx = want_dx ? dx.mutable_data() : nullptr, *p_dp = want_dp ? dp.mutable_data() : nullptr;
    {
        py::gil_scoped_release nogil;
        check(tm_potential_execute_f64(pot.h, N, P, coords.data(), params.data(), box.data(), p_dx, p_dp, want_u ? &u : nullptr));
    }
    return py::make_tuple(want_dx ? py::object(dx) : py::none(), want_dp ? py::object(dp) : py::none(), want_u ? py::object(py::float_(u)) : py::none());
}

// the un-converted accumulators (uint64[N,3] | None, uint64[P] | None, python int | None): not part of the reference surface;
// the parity tests compare integers with it
py::tuple potential_execute_raw(PyPotential &pot, const arr_d &coords, const arr_d &params, const arr_d &box, const bool want_dx, const bool want_dp,
                                const bool want_u) {
    verify_coords_and_box(coords, box);
    const int N = static_cast<int>(coords.shape(0)), P = static_cast<int>(params.size());
    arr_u64 du_dx(std::vector<py::ssize_t>{want_dx ? N : 0, 3}), du_dp(std::vector<py::ssize_t>{want_dp ? P : 0});
    std::memset(du_dx.mutable_data(), 0, sizeof(uint64_t) * du_dx.size());
    std::memset(du_dp.mutable_data(), 0, sizeof(uint64_t) * du_dp.size());
    tm_int128 u{0, 0};
    uint64_t *p_dx = want_dx ? du_dx.mutable_data() : nullptr, *p_dp = want_dp ? du_dp.mutable_data() : nullptr;
    {
        py::gil_scoped_release nogil;
        check(tm_potential_execute(pot.h, N, P, coords.data(), params.data(), box.data(), p_dx, p_dp, want_u ? &u : nullptr));
    }
    return py::make_tuple(want_dx ? py::object(du_dx) : py::none(), want_dp ? py::object(du_dp) : py::none(), want_u ? int128_to_python(u) : py::none());
}

py::tuple potential_execute_batch(PyPotential &pot, const arr_d &coords, const arr_d &params, const arr_d &boxes, const bool want_dx, const bool want_dp,
                                  const bool want_u) {
    if (coords.ndim() != 3 || boxes.ndim() != 3) {
        throw std::runtime_error("coords and boxes must have 3 dimensions");
    }
    if (coords.shape(0) != boxes.shape(0)) {
        throw std::runtime_error("number of batches of coords and boxes don't match");
    }
    if (params.ndim() < 2) {
        throw std::runtime_error("parameters must have at least 2 dimensions");
    }
    const int C = static_cast<int>(coords.shape(0)), N = static_cast<int>(coords.shape(1)), Pb = static_cast<int>(params.shape(0));
    const int P = Pb ? static_cast<int>(params.size() / Pb) : 0;
    std::vector<py::ssize_t> dp_shape{want_dp ? C : 0, Pb};
    for (py::ssize_t d = 1; d < params.ndim(); d++) {
        dp_shape.push_back(params.shape(d));
    }
    arr_d dx(std::vector<py::ssize_t>{want_dx ? C : 0, Pb, N, 3}), dp(dp_shape), e(std::vector<py::ssize_t>{want_u ? C : 0, Pb});
    double *p_dx = want_dx ? dx.mutable_data() : nullptr, *p_dp = want_dp ? dp.mutable_data() : nullptr, *p_u = want_u ? e.mutable_data() : nullptr;
    {
        py::gil_scoped_release nogil;
        check(tm_potential_execute_batch_f64(pot.h, C, N, Pb, P, coords.data(), params.data(), boxes.data(), p_dx, p_dp, p_u));
    }
    return py::make_tuple(want_dx ? py::object(dx) : py::none(), want_dp ? py::object(dp) : py::none(), want_u ? py::object(e) : py::none());
}

py::tuple potential_execute_batch_sparse(PyPotential &pot, const arr_d &coords, const arr_d &params, const arr_d &boxes, const arr_u &coords_batch_idxs,
                                         const arr_u &params_batch_idxs, const bool want_dx, const bool want_dp, const bool want_u) {
    if (coords.ndim() != 3 || boxes.ndim() != 3) {
        throw std::runtime_error("coords and boxes must have 3 dimensions");
    }
    if (coords.shape(0) != boxes.shape(0)) {
        throw std::runtime_error("number of coord arrays and boxes don't match");
    }
    if (params.ndim() < 2) {
        throw std::runtime_error("parameters must have at least 2 dimensions");
    }
    if (coords_batch_idxs.ndim() != 1 || params_batch_idxs.ndim() != 1) {
        throw std::runtime_error("coords_batch_idxs and params_batch_idxs must be one-dimensional arrays");
    }
    if (coords_batch_idxs.size() != params_batch_idxs.size()) {
        throw std::runtime_error("coords_batch_idxs and params_batch_idxs must have the same length");
    }
    const int B = static_cast<int>(coords_batch_idxs.size());
    const int Cs = static_cast<int>(coords.shape(0)), N = static_cast<int>(coords.shape(1)), Ps = static_cast<int>(params.shape(0));
    for (int i = 0; i < B; i++) {
        if (coords_batch_idxs.data()[i] >= static_cast<uint32_t>(Cs)) {
            throw std::runtime_error("coords_batch_idxs contains an index that is out of bounds");
        }
        if (params_batch_idxs.data()[i] >= static_cast<uint32_t>(Ps)) {
            throw std::runtime_error("params_batch_idxs contains an index that is out of bounds");
        }
    }
    const int P = Ps ? static_cast<int>(params.size() / Ps) : 0;
    std::vector<py::ssize_t> dp_shape{want_dp ? B : 0};
    for (py::ssize_t d = 1; d < params.ndim(); d++) {
        dp_shape.push_back(params.shape(d));
    }
    arr_d dx(std::vector<py::ssize_t>{want_dx ? B : 0, N, 3}), dp(dp_shape), e(std::vector<py::ssize_t>{want_u ? B : 0});
    double *p_dx = want_dx ? dx.mutable_data() : nullptr, *p_dp = want_dp ? dp.mutable_data() : nullptr, *p_u = want_u ? e.mutable_data() : nullptr;
    {
        py::gil_scoped_release nogil;
        check(tm_potential_execute_batch_sparse_f64(pot.h, Cs, N, Ps, P, B, coords_batch_idxs.data(), params_batch_idxs.data(), coords.data(), params.data(),
                                                    boxes.data(), p_dx, p_dp, p_u));
    }
    return py::make_tuple(want_dx ? py::object(dx) : py::none(), want_dp ? py::object(dp) : py::none(), want_u ? py::object(e) : py::none());
}

void declare_potential(py::module &m) {
    py::class_<PyPotential, std::shared_ptr<PyPotential>>(m, "Potential", py::dynamic_attr())
        .def("execute", &potential_execute, py::arg("coords"), py::arg("params"), py::arg("box"), py::arg("compute_du_dx") = true, py::arg("compute_du_dp") = true,
             py::arg("compute_u") = true, "-> (du_dx[N,3] | None, du_dp[params.shape] | None, u | None); wrap_kernels.cpp:1039-1105")
        .def("execute_raw", &potential_execute_raw, py::arg("coords"), py::arg("params"), py::arg("box"), py::arg("compute_du_dx") = true,
             py::arg("compute_du_dp") = true, py::arg("compute_u") = true,
             "the fixed-point accumulators as they are: (uint64[N,3] | None, uint64[P] | None, int | None); not in the reference surface")
        .def(
            "execute_du_dx",
            [](PyPotential &pot, const arr_d &coords, const arr_d &params, const arr_d &box) -> py::object {
                return potential_execute(pot, coords, params, box, true, false, false)[0];
            },
            py::arg("coords"), py::arg("params"), py::arg("box"), "wrap_kernels.cpp:1106-1130")
        .def("execute_batch", &potential_execute_batch, py::arg("coords"), py::arg("params"), py::arg("boxes"), py::arg("compute_du_dx"), py::arg("compute_du_dp"),
             py::arg("compute_u"), "-> (du_dx[C,Pb,N,3], du_dp[C,Pb,*params.shape[1:]], u[C,Pb]); wrap_kernels.cpp:738-862")
        .def("execute_batch_sparse", &potential_execute_batch_sparse, py::arg("coords"), py::arg("params"), py::arg("boxes"), py::arg("coords_batch_idxs"),
             py::arg("params_batch_idxs"), py::arg("compute_du_dx"), py::arg("compute_du_dp"), py::arg("compute_u"),
             "-> (du_dx[B,N,3], du_dp[B,*params.shape[1:]], u[B]); wrap_kernels.cpp:863-1038");
}

template <typename C> using pot_class = py::class_<C, std::shared_ptr<C>, PyPotential>;
template <PotKind KIND, int PREC> pot_class<Pot<KIND, PREC>> declare_pot(py::module &m, const char *base_name) {
    const std::string name = std::string(base_name) + (PREC == TM_F64 ? "_f64" : "_f32");
    return pot_class<Pot<KIND, PREC>>(m, name.c_str(), py::dynamic_attr());
}

std::vector<int32_t> to_i32(const std::vector<int> &v) { return std::vector<int32_t>(v.begin(), v.end()); }
template <typename T> struct type_tag {
    using type = T;
};

// the per-potential constructors (wrap_kernels.cpp:1311-1589); each template is declared twice like the reference's (:2186-2216)
template <int PREC> void declare_potentials(py::module &m) {
    declare_pot<HarmonicBond, PREC>(m, "HarmonicBond")
        .def(py::init([](const arr_i &bond_idxs) {
                 if (bond_idxs.size() % 2 != 0) {
                     throw std::runtime_error("bond_idxs.size() must be exactly 2*k!");
                 }
                 auto p = std::make_shared<Pot<HarmonicBond, PREC>>();
                 check(tm_harmonic_bond_create(PREC, bond_idxs.data(), static_cast<int>(bond_idxs.size() / 2), &p->h));
                 return p;
             }),
             py::arg("bond_idxs"));
    declare_pot<HarmonicAngle, PREC>(m, "HarmonicAngle")
        .def(py::init([](const arr_i &angle_idxs) {
                 if (angle_idxs.size() % 3 != 0) {
                     throw std::runtime_error("angle_idxs.size() must be exactly 3*A");
                 }
                 auto p = std::make_shared<Pot<HarmonicAngle, PREC>>();
                 check(tm_harmonic_angle_create(PREC, angle_idxs.data(), static_cast<int>(angle_idxs.size() / 3), &p->h));
                 return p;
             }),
             py::arg("angle_idxs"));
    declare_pot<PeriodicTorsion, PREC>(m, "PeriodicTorsion")
        .def(py::init([](const arr_i &angle_idxs) { // the keyword really is angle_idxs (wrap_kernels.cpp:1432-1444)
                 if (angle_idxs.size() % 4 != 0) {
                     throw std::runtime_error("torsion_idxs.size() must be exactly 4*k");
                 }
                 auto p = std::make_shared<Pot<PeriodicTorsion, PREC>>();
                 check(tm_periodic_torsion_create(PREC, angle_idxs.data(), static_cast<int>(angle_idxs.size() / 4), &p->h));
                 return p;
             }),
             py::arg("angle_idxs"));
    declare_pot<FlatBottomBond, PREC>(m, "FlatBottomBond")
        .def(py::init([](const arr_i &bond_idxs) {
                 if (bond_idxs.size() % 2 != 0) {
                     throw std::runtime_error("bond_idxs.size() must be exactly 2*k!");
                 }
                 auto p = std::make_shared<Pot<FlatBottomBond, PREC>>();
                 check(tm_flat_bottom_bond_create(PREC, 0, bond_idxs.data(), static_cast<int>(bond_idxs.size() / 2), 0.0, &p->h));
                 return p;
             }),
             py::arg("bond_idxs"));
    declare_pot<LogFlatBottomBond, PREC>(m, "LogFlatBottomBond")
        .def(py::init([](const arr_i &bond_idxs, const double beta) {
                 if (bond_idxs.size() % 2 != 0) {
                     throw std::runtime_error("bond_idxs.size() must be exactly 2*k!");
                 }
                 auto p = std::make_shared<Pot<LogFlatBottomBond, PREC>>();
                 check(tm_flat_bottom_bond_create(PREC, 1, bond_idxs.data(), static_cast<int>(bond_idxs.size() / 2), beta, &p->h));
                 return p;
             }),
             py::arg("bond_idxs"), py::arg("beta"));
    declare_pot<CentroidRestraint, PREC>(m, "CentroidRestraint")
        .def(py::init([](const arr_i &group_a_idxs, const arr_i &group_b_idxs, const double kb, const double b0) {
                 auto p = std::make_shared<Pot<CentroidRestraint, PREC>>();
                 check(tm_centroid_restraint_create(PREC, group_a_idxs.data(), static_cast<int>(group_a_idxs.size()), group_b_idxs.data(),
                                                    static_cast<int>(group_b_idxs.size()), kb, b0, &p->h));
                 return p;
             }),
             py::arg("group_a_idxs"), py::arg("group_b_idxs"), py::arg("kb"), py::arg("b0"));
    declare_pot<ChiralAtomRestraint, PREC>(m, "ChiralAtomRestraint")
        .def(py::init([](const arr_i &idxs) {
                 if (idxs.size() % 4 != 0) {
                     throw std::runtime_error("idxs.size() must be exactly 4*k!");
                 }
                 auto p = std::make_shared<Pot<ChiralAtomRestraint, PREC>>();
                 check(tm_chiral_atom_restraint_create(PREC, idxs.data(), static_cast<int>(idxs.size() / 4), &p->h));
                 return p;
             }),
             py::arg("idxs"));
    declare_pot<ChiralBondRestraint, PREC>(m, "ChiralBondRestraint")
        .def(py::init([](const arr_i &idxs, const arr_i &signs) {
                 if (idxs.size() % 4 != 0) {
                     throw std::runtime_error("idxs.size() must be exactly 4*R!");
                 }
                 auto p = std::make_shared<Pot<ChiralBondRestraint, PREC>>();
                 check(tm_chiral_bond_restraint_create(PREC, idxs.data(), static_cast<int>(idxs.size() / 4), signs.data(), static_cast<int>(signs.size()), &p->h));
                 return p;
             }),
             py::arg("idxs"), py::arg("signs"));
    declare_pot<NonbondedPairListPrecomputed, PREC>(m, "NonbondedPairListPrecomputed")
        .def(py::init([](const arr_i &pair_idxs, const double beta, const double cutoff) {
                 if (pair_idxs.size() % 2 != 0) {
                     throw std::runtime_error("idxs.size() must be exactly 2*B!");
                 }
                 auto p = std::make_shared<Pot<NonbondedPairListPrecomputed, PREC>>();
                 check(tm_nonbonded_pair_list_precomputed_create(PREC, pair_idxs.data(), static_cast<int>(pair_idxs.size() / 2), beta, cutoff, &p->h));
                 return p;
             }),
             py::arg("pair_idxs"), py::arg("beta"), py::arg("cutoff"));
    auto pair_list_init = [](auto tag, const int negated) {
        using C = typename decltype(tag)::type;
        return py::init([negated](const arr_i &pair_idxs_i, const arr_d &scales_i, const double beta, const double cutoff) {
            if (pair_idxs_i.size() % 2 != 0) {
                throw std::runtime_error("pair_idxs.size() must be even, but got " + std::to_string(pair_idxs_i.size()));
            }
            auto p = std::make_shared<C>();
            check(tm_nonbonded_pair_list_create(PREC, negated, pair_idxs_i.data(), static_cast<int>(pair_idxs_i.size() / 2), scales_i.data(),
                                                static_cast<int>(scales_i.size() / 2), beta, cutoff, &p->h));
            return p;
        });
    };
    declare_pot<NonbondedPairList, PREC>(m, "NonbondedPairList")
        .def(pair_list_init(type_tag<Pot<NonbondedPairList, PREC>>{}, 0), py::arg("pair_idxs_i"), py::arg("scales_i"), py::arg("beta"),
             py::arg("cutoff"));
    declare_pot<NonbondedExclusions, PREC>(m, "NonbondedExclusions")
        .def(pair_list_init(type_tag<Pot<NonbondedExclusions, PREC>>{}, 1), py::arg("pair_idxs_i"), py::arg("scales_i"), py::arg("beta"),
             py::arg("cutoff"));

    using AllPairs = Pot<NonbondedAllPairs, PREC>;
    declare_pot<NonbondedAllPairs, PREC>(m, "NonbondedAllPairs")
        .def(py::init([](const int num_atoms, const double beta, const double cutoff, const std::optional<arr_i> &atom_idxs_i, const bool disable_hilbert_sort,
                         const double nblist_padding) {
                 auto p = std::make_shared<AllPairs>();
                 check(tm_nonbonded_all_pairs_create(PREC, num_atoms, beta, cutoff, atom_idxs_i ? atom_idxs_i->data() : nullptr,
                                                     atom_idxs_i ? static_cast<int>(atom_idxs_i->size()) : 0, disable_hilbert_sort ? 1 : 0, nblist_padding, &p->h));
                 return p;
             }),
             py::arg("num_atoms"), py::arg("beta"), py::arg("cutoff"), py::arg("atom_idxs_i") = py::none(), py::arg("disable_hilbert_sort") = false,
             py::arg("nblist_padding") = 0.1)
        .def(
            "set_atom_idxs",
            [](AllPairs &p, const std::vector<int> &atom_idxs) {
                const std::vector<int32_t> v = to_i32(atom_idxs);
                check(tm_nonbonded_all_pairs_set_atom_idxs(p.h, v.data(), static_cast<int>(v.size())));
            },
            py::arg("atom_idxs"))
        .def("get_atom_idxs",
             [](AllPairs &p) {
                 int n = 0;
                 check(tm_nonbonded_all_pairs_get_num_atom_idxs(p.h, &n));
                 std::vector<int32_t> v(n);
                 check(tm_nonbonded_all_pairs_get_atom_idxs(p.h, v.data(), n));
                 return std::vector<int>(v.begin(), v.end());
             })
        .def("get_num_atom_idxs",
             [](AllPairs &p) {
                 int n = 0;
                 check(tm_nonbonded_all_pairs_get_num_atom_idxs(p.h, &n));
                 return n;
             })
        // diagnostics (not in the reference surface): 32 x 32 tiles in the current list; list builds since construction;
        // per-wave cycle counters of the last tile launch (-DTM_TIMING builds)
        .def("get_tile_ixn_count",
             [](AllPairs &p) {
                 unsigned int n = 0;
                 check(tm_nonbonded_all_pairs_get_tile_count(p.h, &n));
                 return n;
             })
        .def("get_build_count",
             [](AllPairs &p) {
                 unsigned int n = 0;
                 check(tm_nonbonded_all_pairs_get_build_count(p.h, &n));
                 return n;
             })
        .def("get_same_frame_skips", [](AllPairs &p) { // diagnostic: evaluations that launched no list kernel on the batch entry point's word
            long long skips = 0;
            check(tm_nonbonded_all_pairs_get_same_frame_skips(p.h, &skips));
            return skips;
        })
        .def("get_memo_stats", [](AllPairs &p) { // diagnostic: (energy-only evaluations remembered on the device, of which the all-pairs launch was empty)
            long long evals = 0, skipped = 0;
            check(tm_nonbonded_all_pairs_get_memo_stats(p.h, &evals, &skipped));
            return py::make_tuple(evals, skipped);
        })
        .def("get_merged_stats", [](AllPairs &p) { // diagnostic: (evaluations made as the carrier of an interaction group, its list's tiles, its list's builds)
            long long calls = 0;
            unsigned int tiles = 0, builds = 0;
            check(tm_nonbonded_all_pairs_get_merged_stats(p.h, &calls, &tiles, &builds));
            return py::make_tuple(calls, tiles, builds);
        })
        .def(
            "debug_timing",
            [](AllPairs &p, const int max_waves) {
                py::array_t<long long, py::array::c_style> out(std::vector<py::ssize_t>{max_waves, 8});
                std::memset(out.mutable_data(), 0, sizeof(long long) * out.size());
                int n = 0;
                check(tm_nonbonded_all_pairs_debug_timing(p.h, out.mutable_data(), static_cast<int>(out.size()), &n));
                return py::make_tuple(out, n);
            },
            py::arg("max_waves") = 8192);

    using Group = Pot<NonbondedInteractionGroup, PREC>;
    declare_pot<NonbondedInteractionGroup, PREC>(m, "NonbondedInteractionGroup")
        .def(py::init([](const int num_atoms, const arr_i &row_atom_idxs_i, const double beta, const double cutoff, const std::optional<arr_i> &col_atom_idxs_i,
                         const bool disable_hilbert_sort, const double nblist_padding) {
                 auto p = std::make_shared<Group>();
                 check(tm_nonbonded_interaction_group_create(PREC, num_atoms, row_atom_idxs_i.data(), static_cast<int>(row_atom_idxs_i.size()),
                                                             col_atom_idxs_i ? col_atom_idxs_i->data() : nullptr,
                                                             col_atom_idxs_i ? static_cast<int>(col_atom_idxs_i->size()) : 0, beta, cutoff,
                                                             disable_hilbert_sort ? 1 : 0, nblist_padding, &p->h));
                 return p;
             }),
             py::arg("num_atoms"), py::arg("row_atom_idxs_i"), py::arg("beta"), py::arg("cutoff"), py::arg("col_atom_idxs_i") = py::none(),
             py::arg("disable_hilbert_sort") = false, py::arg("nblist_padding") = 0.1)
        .def(
            "set_atom_idxs",
            [](Group &p, const std::vector<int> &row_atom_idxs, const std::vector<int> &col_atom_idxs) {
                const std::vector<int32_t> r = to_i32(row_atom_idxs), c = to_i32(col_atom_idxs);
                check(tm_nonbonded_interaction_group_set_atom_idxs(p.h, r.data(), static_cast<int>(r.size()), c.data(), static_cast<int>(c.size())));
            },
            py::arg("row_atom_idxs"), py::arg("col_atom_idxs"));
}

void declare_summed_potentials(py::module &m) {
    py::class_<PySummed, std::shared_ptr<PySummed>, PyPotential>(m, "SummedPotential", py::dynamic_attr())
        .def(py::init([](const std::vector<std::shared_ptr<PyPotential>> &potentials, const std::vector<int> &params_sizes, const bool parallel) {
                 auto p = std::make_shared<PySummed>();
                 const std::vector<tm_potential_t> hs = handles_of(potentials);
                 const std::vector<int32_t> sizes = to_i32(params_sizes);
                 check(tm_summed_potential_create(hs.data(), static_cast<int>(hs.size()), sizes.data(), static_cast<int>(sizes.size()), parallel ? 1 : 0, &p->h));
                 p->children = potentials;
                 return p;
             }),
             py::arg("potentials"), py::arg("params_sizes"), py::arg("parallel") = true)
        .def("get_potentials", [](PySummed &p) { return p.children; });
    py::class_<PyFanout, std::shared_ptr<PyFanout>, PyPotential>(m, "FanoutSummedPotential", py::dynamic_attr())
        .def(py::init([](const std::vector<std::shared_ptr<PyPotential>> &potentials, const bool parallel) {
                 auto p = std::make_shared<PyFanout>();
                 const std::vector<tm_potential_t> hs = handles_of(potentials);
                 check(tm_fanout_summed_potential_create(hs.data(), static_cast<int>(hs.size()), parallel ? 1 : 0, &p->h));
                 p->children = potentials;
                 return p;
             }),
             py::arg("potentials"), py::arg("parallel") = true)
        .def("get_potentials", [](PyFanout &p) { return p.children; });
}

// ---- BoundPotential (wrap_kernels.cpp:1133-1309) ------------------------------------------------------------------------
void declare_bound_potential(py::module &m) {
    py::class_<PyBound, std::shared_ptr<PyBound>>(m, "BoundPotential", py::dynamic_attr())
        .def(py::init([](std::shared_ptr<PyPotential> potential, const arr_d &params) {
                 if (!potential) {
                     throw std::runtime_error("got nullptr instead of potential");
                 }
                 auto b = std::make_shared<PyBound>();
                 check(tm_bound_potential_create(potential->h, params.data(), static_cast<int>(params.size()), &b->h));
                 b->potential = std::move(potential);
                 return b;
             }),
             py::arg("potential"), py::arg("params"))
        .def("get_potential", [](const PyBound &b) { return b.potential; })
        .def(
            "set_params", [](PyBound &b, const arr_d &params) { check(tm_bound_potential_set_params(b.h, params.data(), static_cast<int>(params.size()))); },
            py::arg("params"))
        .def("size",
             [](PyBound &b) {
                 int n = 0;
                 check(tm_bound_potential_size(b.h, &n));
                 return n;
             })
        .def(
            "execute",
            [](PyBound &b, const arr_d &coords, const arr_d &box, const bool want_dx, const bool want_u) -> py::tuple {
                verify_coords_and_box(coords, box);
                const int N = static_cast<int>(coords.shape(0));
                arr_d dx(std::vector<py::ssize_t>{want_dx ? N : 0, 3});
                double u = 0.0;
                double *p_dx = want_dx ? dx.mutable_data() : nullptr;
                {
                    py::gil_scoped_release nogil;
                    check(tm_bound_potential_execute_f64(b.h, N, coords.data(), box.data(), p_dx, want_u ? &u : nullptr));
                }
                return py::make_tuple(want_dx ? py::object(dx) : py::none(), want_u ? py::object(py::float_(u)) : py::none());
            },
            py::arg("coords"), py::arg("box"), py::arg("compute_du_dx") = true, py::arg("compute_u") = true, "-> (du_dx | None, u | None); wrap_kernels.cpp:1149-1186")
        .def(
            "execute_batch",
            [](PyBound &b, const arr_d &coords, const arr_d &boxes, const bool want_dx, const bool want_u) -> py::tuple {
                if (coords.ndim() != 3 && boxes.ndim() != 3) { // (the reference's condition, wrap_kernels.cpp:1193)
                    throw std::runtime_error("coords and boxes must have 3 dimensions");
                }
                if (coords.shape(0) != boxes.shape(0)) {
                    throw std::runtime_error("number of batches of coords and boxes don't match");
                }
                const int C = static_cast<int>(coords.shape(0)), N = static_cast<int>(coords.shape(1));
                arr_d dx(std::vector<py::ssize_t>{want_dx ? C : 0, N, 3}), e(std::vector<py::ssize_t>{want_u ? C : 0});
                double *p_dx = want_dx ? dx.mutable_data() : nullptr, *p_u = want_u ? e.mutable_data() : nullptr;
                {
                    py::gil_scoped_release nogil;
                    check(tm_bound_potential_execute_batch_f64(b.h, C, N, coords.data(), boxes.data(), p_dx, p_u));
                }
                return py::make_tuple(want_dx ? py::object(dx) : py::none(), want_u ? py::object(e) : py::none());
            },
            py::arg("coords"), py::arg("boxes"), py::arg("compute_du_dx"), py::arg("compute_u"), "-> (du_dx[C,N,3] | None, u[C] | None); wrap_kernels.cpp:1187-1274")
        .def(
            "execute_fixed",
            [](PyBound &b, const arr_d &coords, const arr_d &box) {
                verify_coords_and_box(coords, box);
                const int N = static_cast<int>(coords.shape(0));
                tm_int128 u{0, 0};
                {
                    py::gil_scoped_release nogil;
                    check(tm_bound_potential_execute(b.h, N, coords.data(), box.data(), nullptr, &u));
                }
                arr_u64 out(std::vector<py::ssize_t>{1});
                out.mutable_data()[0] = tm_energy_overflowed(&u) ? static_cast<uint64_t>(LLONG_MAX) : u.lo;
                return out;
            },
            py::arg("coords"), py::arg("box"), "-> uint64[1]: the fixed-point energy, LLONG_MAX when it overflowed; wrap_kernels.cpp:1275-1308");
}

// ---- integrators, movers (wrap_kernels.cpp:691-729, 1591-1659) ----------------------------------------------------------
void declare_integrators_and_movers(py::module &m) {
    py::class_<PyIntegrator, std::shared_ptr<PyIntegrator>>(m, "Integrator", py::dynamic_attr());
    py::class_<PyLangevin, std::shared_ptr<PyLangevin>, PyIntegrator>(m, "LangevinIntegrator", py::dynamic_attr())
        .def(py::init([](const arr_d &masses, const double temperature, const double dt, const double friction, const int seed) {
                 auto p = std::make_shared<PyLangevin>();
                 check(tm_langevin_integrator_create(masses.data(), static_cast<int>(masses.size()), temperature, dt, friction, seed, &p->h));
                 return p;
             }),
             py::arg("masses"), py::arg("temperature"), py::arg("dt"), py::arg("friction"), py::arg("seed"));
    py::class_<PyVerlet, std::shared_ptr<PyVerlet>, PyIntegrator>(m, "VelocityVerletIntegrator", py::dynamic_attr())
        .def(py::init([](const double dt, const arr_d &cbs) {
                 auto p = std::make_shared<PyVerlet>();
                 check(tm_velocity_verlet_integrator_create(dt, cbs.data(), static_cast<int>(cbs.size()), &p->h));
                 return p;
             }),
             py::arg("dt"), py::arg("cbs"));

    py::class_<PyMover, std::shared_ptr<PyMover>>(m, "Mover", py::dynamic_attr())
        .def(
            "set_interval", [](PyMover &mv, const int interval) { check(tm_mover_set_interval(mv.h, interval)); }, py::arg("interval"))
        .def("get_interval",
             [](PyMover &mv) {
                 int n = 0;
                 check(tm_mover_get_interval(mv.h, &n));
                 return n;
             })
        .def(
            "set_step", [](PyMover &mv, const int step) { check(tm_mover_set_step(mv.h, step)); }, py::arg("step"))
        .def(
            "move",
            [](PyMover &mv, const arr_d &coords, const arr_d &box) -> py::tuple {
                verify_coords_and_box(coords, box);
                const int N = static_cast<int>(coords.shape(0));
                arr_d x_out(std::vector<py::ssize_t>{N, 3}), box_out(std::vector<py::ssize_t>{3, 3});
                double *px = x_out.mutable_data(), *pb = box_out.mutable_data();
                {
                    py::gil_scoped_release nogil;
                    check(tm_mover_move(mv.h, N, coords.data(), box.data(), px, pb));
                }
                return py::make_tuple(x_out, box_out);
            },
            py::arg("coords"), py::arg("box"));
    py::class_<PyBarostat, std::shared_ptr<PyBarostat>, PyMover>(m, "MonteCarloBarostat", py::dynamic_attr())
        .def(py::init([](const int N, const double pressure, const double temperature, const std::vector<std::vector<int>> &group_idxs, const int interval,
                         const std::vector<std::shared_ptr<PyBound>> &bps, const int seed, const bool adaptive_scaling_enabled,
                         const double initial_volume_scale_factor) {
                 std::vector<int32_t> flat, offsets{0};
                 for (const auto &g : group_idxs) {
                     flat.insert(flat.end(), g.begin(), g.end());
                     offsets.push_back(static_cast<int32_t>(flat.size()));
                 }
                 const std::vector<tm_bound_potential_t> hs = handles_of(bps);
                 auto p = std::make_shared<PyBarostat>();
                 check(tm_monte_carlo_barostat_create(N, pressure, temperature, flat.data(), offsets.data(), static_cast<int>(group_idxs.size()), interval, hs.data(),
                                                      static_cast<int>(hs.size()), seed, adaptive_scaling_enabled ? 1 : 0, initial_volume_scale_factor, &p->h));
                 p->bps = bps;
                 return p;
             }),
             py::arg("N"), py::arg("pressure"), py::arg("temperature"), py::arg("group_idxs"), py::arg("interval"), py::arg("bps"), py::arg("seed"),
             py::arg("adaptive_scaling_enabled"), py::arg("initial_volume_scale_factor"))
        .def(
            "set_volume_scale_factor", [](PyBarostat &b, const double f) { check(tm_barostat_set_volume_scale_factor(b.h, f)); }, py::arg("volume_scale_factor"))
        .def("get_volume_scale_factor",
             [](PyBarostat &b) {
                 double f = 0;
                 check(tm_barostat_get_volume_scale_factor(b.h, &f));
                 return f;
             })
        .def(
            "set_adaptive_scaling", [](PyBarostat &b, const bool on) { check(tm_barostat_set_adaptive_scaling(b.h, on ? 1 : 0)); },
            py::arg("adaptive_scaling_enabled"))
        .def("get_adaptive_scaling",
             [](PyBarostat &b) {
                 int on = 0;
                 check(tm_barostat_get_adaptive_scaling(b.h, &on));
                 return on != 0;
             })
        .def(
            "set_pressure", [](PyBarostat &b, const double pressure) { check(tm_barostat_set_pressure(b.h, pressure)); }, py::arg("pressure"))
        .def("get_counters", [](PyBarostat &b) { // diagnostic (not in the reference surface): (accepted, attempted)
            int acc = 0, att = 0;
            check(tm_barostat_get_counters(b.h, &acc, &att));
            return py::make_tuple(acc, att);
        })
        .def("get_attempt_paths", [](PyBarostat &b) { // diagnostic: (attempts since construction, of which on the potential's current list)
            long long att = 0, fast = 0;
            check(tm_barostat_get_attempt_paths(b.h, &att, &fast));
            return py::make_tuple(att, fast);
        });
}

// ---- Context (wrap_kernels.cpp:296-689) ---------------------------------------------------------------------------------
int local_num_samples(const int n_steps, const int store_x_interval) {
    // sizes only: the C ABI validates (with the binding's messages) before anything is written
    if (n_steps <= 0 || store_x_interval < 0) {
        return 0;
    }
    return n_steps / (store_x_interval == 0 ? n_steps : store_x_interval);
}

void declare_context(py::module &m) {
    py::class_<PyContext, std::shared_ptr<PyContext>>(m, "Context", py::dynamic_attr())
        .def(py::init([](const arr_d &x0, const arr_d &v0, const arr_d &box, std::shared_ptr<PyIntegrator> integrator,
                         const std::vector<std::shared_ptr<PyBound>> &bps, const std::optional<std::vector<std::shared_ptr<PyMover>>> &movers) {
                 verify_coords_and_box(x0, box);
                 if (x0.shape(0) != v0.shape(0)) {
                     throw std::runtime_error("v0 N != x0 N");
                 }
                 if (v0.ndim() != 2 || x0.shape(1) != v0.shape(1)) {
                     throw std::runtime_error("v0 D != x0 D");
                 }
                 if (!integrator) {
                     throw std::runtime_error("got nullptr instead of integrator");
                 }
                 auto c = std::make_shared<PyContext>();
                 c->N = static_cast<int>(x0.shape(0));
                 if (movers) {
                     c->movers = *movers;
                 }
                 std::vector<tm_mover_t> mh;
                 for (const auto &mv : c->movers) {
                     if (!mv) {
                         throw std::runtime_error("got nullptr instead of mover");
                     }
                     mh.push_back(mv->h);
                 }
                 const std::vector<tm_bound_potential_t> bh = handles_of(bps);
                 check(tm_context_create_with_movers(x0.data(), v0.data(), box.data(), c->N, integrator->h, bh.data(), static_cast<int>(bh.size()), mh.data(),
                                                     static_cast<int>(mh.size()), &c->h));
                 c->integrator = std::move(integrator);
                 c->bps = bps;
                 return c;
             }),
             py::arg("x0"), py::arg("v0"), py::arg("box"), py::arg("integrator"), py::arg("bps"), py::arg("movers") = py::none())
        .def("step", [](PyContext &c) { check(tm_context_step(c.h)); }, py::call_guard<py::gil_scoped_release>())
        .def("initialize", [](PyContext &c) { check(tm_context_initialize(c.h)); }, py::call_guard<py::gil_scoped_release>())
        .def("finalize", [](PyContext &c) { check(tm_context_finalize(c.h)); }, py::call_guard<py::gil_scoped_release>())
        .def(
            "multiple_steps",
            [](PyContext &c, const int n_steps, const int store_x_interval) -> py::tuple {
                if (store_x_interval < 0) {
                    throw std::runtime_error("store_x_interval must be greater than or equal to zero");
                }
                const int x_interval = store_x_interval == 0 ? n_steps : store_x_interval;
                const int n_samples = x_interval > 0 ? n_steps / x_interval : 0;
                arr_d xs(std::vector<py::ssize_t>{n_samples, c.N, 3}), boxes(std::vector<py::ssize_t>{n_samples, 3, 3});
                double *px = xs.mutable_data(), *pb = boxes.mutable_data();
                {
                    py::gil_scoped_release nogil;
                    check(tm_context_multiple_steps(c.h, n_steps, n_samples, px, pb));
                }
                return py::make_tuple(xs, boxes);
            },
            py::arg("n_steps"), py::arg("store_x_interval") = 0,
            "-> (xs[F,N,3], boxes[F,3,3]), F = n_steps // (store_x_interval or n_steps); wrap_kernels.cpp:347-369")
        .def("last_multiple_steps_ms",
             [](PyContext &c) { // measurement aid (not in the reference surface): device time of the last multiple_steps call's steps
                 double ms = 0;
                 check(tm_context_last_multiple_steps_ms(c.h, &ms));
                 return ms;
             })
        .def(
            "setup_local_md",
            [](PyContext &c, const double temperature, const bool freeze_reference) { check(tm_context_setup_local_md(c.h, temperature, freeze_reference ? 1 : 0)); },
            py::arg("temperature"), py::arg("freeze_reference"))
        .def(
            "multiple_steps_local",
            [](PyContext &c, const int n_steps, const arr_i &local_idxs, const int store_x_interval, const double radius, const double k, const int seed) -> py::tuple {
                const int n_samples = local_num_samples(n_steps, store_x_interval);
                arr_d xs(std::vector<py::ssize_t>{n_samples, c.N, 3}), boxes(std::vector<py::ssize_t>{n_samples, 3, 3});
                double *px = xs.mutable_data(), *pb = boxes.mutable_data();
                {
                    py::gil_scoped_release nogil;
                    check(tm_context_multiple_steps_local(c.h, n_steps, local_idxs.data(), static_cast<int>(local_idxs.size()), store_x_interval, radius, k, seed, px, pb));
                }
                return py::make_tuple(xs, boxes);
            },
            py::arg("n_steps"), py::arg("local_idxs"), py::arg("store_x_interval") = 0, py::arg("radius") = 1.2, py::arg("k") = 10000.0, py::arg("seed") = 2022)
        .def(
            "multiple_steps_local_selection",
            [](PyContext &c, const int n_steps, const int reference_idx, const arr_i &selection_idxs, const int store_x_interval, const double radius,
               const double k) -> py::tuple {
                const int n_samples = local_num_samples(n_steps, store_x_interval);
                arr_d xs(std::vector<py::ssize_t>{n_samples, c.N, 3}), boxes(std::vector<py::ssize_t>{n_samples, 3, 3});
                double *px = xs.mutable_data(), *pb = boxes.mutable_data();
                {
                    py::gil_scoped_release nogil;
                    check(tm_context_multiple_steps_local_selection(c.h, n_steps, reference_idx, selection_idxs.data(), static_cast<int>(selection_idxs.size()),
                                                                    store_x_interval, radius, k, px, pb));
                }
                return py::make_tuple(xs, boxes);
            },
            py::arg("n_steps"), py::arg("reference_idx"), py::arg("selection_idxs"), py::arg("store_x_interval") = 0, py::arg("radius") = 1.2, py::arg("k") = 10000.0)
        .def("local_md_last_selection",
             [](PyContext &c) { // diagnostic: (reference atom, atoms that moved) of the last local-MD call; (-1, empty) before the first
                 int ref = -1;
                 std::vector<unsigned int> free_idxs(c.N, static_cast<unsigned int>(c.N));
                 check(tm_context_local_md_last_selection(c.h, &ref, free_idxs.data()));
                 std::vector<int32_t> moved;
                 if (ref >= 0) {
                     for (int i = 0; i < c.N; i++) {
                         if (free_idxs[i] < static_cast<unsigned int>(c.N)) {
                             moved.push_back(i);
                         }
                     }
                 }
                 return py::make_tuple(ref, arr_i(static_cast<py::ssize_t>(moved.size()), moved.data()));
             })
        .def(
            "set_x_t",
            [](PyContext &c, const arr_d &coords) {
                if (coords.ndim() < 1 || coords.shape(0) != c.N) {
                    throw std::runtime_error("number of new coords disagree with current coords");
                }
                check(tm_context_set_x_t(c.h, coords.data()));
            },
            py::arg("coords"))
        .def(
            "set_v_t",
            [](PyContext &c, const arr_d &velocities) {
                if (velocities.ndim() < 1 || velocities.shape(0) != c.N) {
                    throw std::runtime_error("number of new velocities disagree with current coords");
                }
                check(tm_context_set_v_t(c.h, velocities.data()));
            },
            py::arg("velocities"))
        .def(
            "set_box",
            [](PyContext &c, const arr_d &box) {
                if (box.size() != 9 || box.ndim() < 1 || box.shape(0) != 3) {
                    throw std::runtime_error("box must be 3x3");
                }
                check(tm_context_set_box(c.h, box.data()));
            },
            py::arg("box"))
        .def("get_x_t",
             [](PyContext &c) {
                 arr_d out(std::vector<py::ssize_t>{c.N, 3});
                 check(tm_context_get_x_t(c.h, out.mutable_data()));
                 return out;
             })
        .def("get_v_t",
             [](PyContext &c) {
                 arr_d out(std::vector<py::ssize_t>{c.N, 3});
                 check(tm_context_get_v_t(c.h, out.mutable_data()));
                 return out;
             })
        .def("get_box",
             [](PyContext &c) {
                 arr_d out(std::vector<py::ssize_t>{3, 3});
                 check(tm_context_get_box(c.h, out.mutable_data()));
                 return out;
             })
        .def("get_integrator", [](PyContext &c) { return c.integrator; })
        .def("get_potentials", [](PyContext &c) { return c.bps; })
        .def("get_movers", [](PyContext &c) { return c.movers; })
        .def("get_barostat", [](PyContext &c) -> py::object { // the first MonteCarloBarostat among the movers, else None (wrap_kernels.cpp:671-680)
            for (const auto &mv : c.movers) {
                if (auto b = std::dynamic_pointer_cast<PyBarostat>(mv)) {
                    return py::cast(b);
                }
            }
            return py::none();
        });
}

// ---- Neighborlist_f32/_f64, HilbertSort (wrap_kernels.cpp:113-194) ------------------------------------------------------
template <int PREC> void declare_neighborlist(py::module &m, const char *name) {
    using NL = PyNeighborlist<PREC>;
    py::class_<NL, std::shared_ptr<NL>>(m, name, py::dynamic_attr())
        .def(py::init([](const int N) {
                 auto p = std::make_shared<NL>();
                 check(tm_neighborlist_create(PREC, N, &p->h));
                 return p;
             }),
             py::arg("N"))
        .def(
            "compute_block_bounds",
            [](NL &nl, const arr_d &coords, const arr_d &box, const int block_size) -> py::tuple {
                if (block_size != 32) {
                    throw std::runtime_error("Block size must be 32.");
                }
                verify_coords_and_box(coords, box);
                const int N = static_cast<int>(coords.shape(0)), B = (N + block_size - 1) / block_size;
                arr_d ctrs(std::vector<py::ssize_t>{B, 3}), exts(std::vector<py::ssize_t>{B, 3});
                check(tm_neighborlist_compute_block_bounds(nl.h, N, coords.data(), box.data(), block_size, ctrs.mutable_data(), exts.mutable_data()));
                return py::make_tuple(ctrs, exts);
            },
            py::arg("coords"), py::arg("box"), py::arg("block_size"))
        .def(
            "get_nblist",
            [](NL &nl, const arr_d &coords, const arr_d &box, const double cutoff) {
                verify_coords_and_box(coords, box);
                int nrb = 0, total = 0;
                check(tm_neighborlist_get_nblist(nl.h, static_cast<int>(coords.shape(0)), coords.data(), box.data(), cutoff, &nrb, &total));
                std::vector<int32_t> offsets(nrb + 1, 0), atoms(total > 0 ? total : 1, 0);
                check(tm_neighborlist_copy_nblist(nl.h, offsets.data(), atoms.data()));
                std::vector<std::vector<int>> out(nrb);
                for (int r = 0; r < nrb; r++) {
                    out[r].assign(atoms.begin() + offsets[r], atoms.begin() + offsets[r + 1]);
                }
                return out;
            },
            py::arg("coords"), py::arg("box"), py::arg("cutoff"))
        .def(
            "set_row_idxs", [](NL &nl, const arr_u &idxs) { check(tm_neighborlist_set_row_idxs(nl.h, idxs.data(), static_cast<int>(idxs.size()))); },
            py::arg("idxs"))
        .def("reset_row_idxs", [](NL &nl) { check(tm_neighborlist_reset_row_idxs(nl.h)); })
        .def(
            "resize", [](NL &nl, const int size) { check(tm_neighborlist_resize(nl.h, size)); }, py::arg("size"))
        .def("get_tile_ixn_count",
             [](NL &nl) {
                 unsigned int n = 0;
                 check(tm_neighborlist_get_tile_ixn_count(nl.h, &n));
                 return n;
             })
        .def("get_max_ixn_count",
             [](NL &nl) {
                 int n = 0;
                 check(tm_neighborlist_get_max_ixn_count(nl.h, &n));
                 return n;
             })
        .def("get_num_row_idxs", [](NL &nl) {
            int n = 0;
            check(tm_neighborlist_get_num_row_idxs(nl.h, &n));
            return n;
        });
}

void declare_hilbert_sort(py::module &m) {
    py::class_<PyHilbertSort, std::shared_ptr<PyHilbertSort>>(m, "HilbertSort", py::dynamic_attr())
        .def(py::init([](const int size) {
                 auto p = std::make_shared<PyHilbertSort>();
                 check(tm_hilbert_sort_create(size, &p->h));
                 return p;
             }),
             py::arg("size"))
        .def(
            "sort",
            [](PyHilbertSort &hs, const arr_d &coords, const arr_d &box) {
                verify_coords_and_box(coords, box);
                const int N = static_cast<int>(coords.shape(0));
                arr_u perm(std::vector<py::ssize_t>{N});
                check(tm_hilbert_sort_sort(hs.h, N, coords.data(), box.data(), perm.mutable_data()));
                return perm;
            },
            py::arg("coords"), py::arg("box"));
}

// ---- module-level functions ---------------------------------------------------------------------------------------------
void declare_functions(py::module &m) {
    m.def("cuda_device_reset", []() { check(tm_device_reset()); },
          "Destroy all allocations and reset all state on the current device in the current process (wrap_kernels.cpp:2222-2225; the name is the "
          "reference's, the device is the HIP device).");
    m.def("device_count", []() {
        int n = 0;
        check(tm_device_count(&n));
        return n;
    });
    m.def("set_device", [](const int idx) { check(tm_set_device(idx)); }, py::arg("idx"), "one process per GPU: call with LOCAL_RANK before creating objects");
    m.def("device_synchronize", []() { check(tm_device_synchronize()); }, py::call_guard<py::gil_scoped_release>());
    m.def("device_name", []() {
        char buf[256] = {0};
        check(tm_device_name(buf, sizeof(buf)));
        return std::string(buf);
    });
    m.def("version", []() { return std::string(tm_version()); });
    m.def("hilbert_lut", []() { // host only: the 128^3 bin -> Hilbert index table (cpp/src/hilbert_sort.cu:18-31)
        arr_u out(std::vector<py::ssize_t>{128 * 128 * 128});
        check(tm_hilbert_lut(out.mutable_data()));
        return out;
    });
    m.def(
        "es_force_table",
        [](const double beta) { // host only: [256, 6] coefficients of the f64 kernels' electrostatic force factor F(d^2)
            arr_d out(std::vector<py::ssize_t>{256, 6});
            check(tm_es_force_table(beta, out.mutable_data()));
            return out;
        },
        py::arg("beta"));
    m.def(
        "es_energy_table",
        [](const double beta) { // host only: [256, 6] coefficients of the f64 kernels' electrostatic energy factor G(d^2)
            arr_d out(std::vector<py::ssize_t>{256, 6});
            check(tm_es_energy_table(beta, out.mutable_data()));
            return out;
        },
        py::arg("beta"));
    m.def(
        "hrex_run_neighbor_swaps",
        [](const arr_i64 &replica_idx_by_state, const arr_i64 &neighbor_pairs, const arr_d &log_q_kl, const arr_i64 &pair_idxs, const arr_d &uniform_samples) {
            // the swap chain of one HREX exchange step, natively (timemachine/md/hrex.py:50-130 is a jitted lax.scan)
            const int n_states = static_cast<int>(replica_idx_by_state.size()), n_pairs = static_cast<int>(neighbor_pairs.size() / 2);
            if (log_q_kl.ndim() != 2 || log_q_kl.shape(1) != n_states || pair_idxs.size() != uniform_samples.size()) {
                throw std::runtime_error("run_neighbor_swaps: inconsistent shapes");
            }
            arr_i64 out(std::vector<py::ssize_t>{n_states});
            arr_u proposed(std::vector<py::ssize_t>{n_pairs}), accepted(std::vector<py::ssize_t>{n_pairs});
            std::memset(proposed.mutable_data(), 0, sizeof(uint32_t) * n_pairs);
            std::memset(accepted.mutable_data(), 0, sizeof(uint32_t) * n_pairs);
            check(tm_hrex_run_neighbor_swaps(static_cast<int>(log_q_kl.shape(0)), n_states, replica_idx_by_state.data(), n_pairs, neighbor_pairs.data(),
                                             log_q_kl.data(), static_cast<int>(pair_idxs.size()), pair_idxs.data(), uniform_samples.data(), out.mutable_data(),
                                             proposed.mutable_data(), accepted.mutable_data()));
            return py::make_tuple(out, proposed, accepted);
        },
        py::arg("replica_idx_by_state"), py::arg("neighbor_pairs"), py::arg("log_q_kl"), py::arg("pair_idxs"), py::arg("uniform_samples"));
    m.def(
        "debug_float_to_fixed",
        [](const py::array_t<double, py::array::c_style | py::array::forcecast> &values, const py::object &precision, const int kind) {
            // the DEVICE's fixed-point conversion of `values` (tm_debug_float_to_fixed); kind 3 takes [n, 2] (prefactor, delta) pairs
            const int n = kind == 3 ? static_cast<int>(values.shape(0)) : static_cast<int>(values.size());
            const bool f64 = py::dtype::from_args(precision).is(py::dtype::of<double>()) || py::dtype::from_args(precision).itemsize() == 8;
            arr_u64 out(std::vector<py::ssize_t>{n});
            check(tm_debug_float_to_fixed(f64 ? TM_F64 : TM_F32, kind, values.data(), n, out.mutable_data()));
            return out;
        },
        py::arg("values"), py::arg("precision"), py::arg("kind") = 0);
    m.def(
        "debug_float_to_fixed_energy",
        [](const py::array_t<double, py::array::c_style | py::array::forcecast> &values, const py::object &precision) {
            const int n = static_cast<int>(values.size());
            const bool f64 = py::dtype::from_args(precision).itemsize() == 8;
            std::vector<tm_int128> out(n, tm_int128{0, 0});
            check(tm_debug_float_to_fixed_energy(f64 ? TM_F64 : TM_F32, values.data(), n, out.data()));
            py::list result;
            for (const tm_int128 &e : out) {
                result.append(int128_to_python(e));
            }
            return result;
        },
        py::arg("values"), py::arg("precision"));
    m.def("debug_set_box_scaling_reuse", [](const bool enabled) { check(tm_debug_set_box_scaling_reuse(enabled ? 1 : 0)); }, py::arg("enabled"));
    m.def(
        "debug_set_static_list_max_k",
        [](const int max_atoms) { // A/B aid: static complete lists for potentials over <= max_atoms atoms (0: off); -> the old value
            int previous = 0;
            check(tm_debug_set_static_list_max_k(max_atoms, &previous));
            return previous;
        },
        py::arg("max_atoms"));
    m.def(
        "multiple_steps_group",
        [](const std::vector<std::shared_ptr<PyContext>> &ctxts, const int n_steps) {
            // not in the reference surface: n_steps of several distinct Contexts interleaved on their own streams (windows or HREX
            // replicas that share a GPU: one context's list / update kernels run underneath another's force kernel)
            std::vector<tm_context_t> hs;
            for (const auto &c : ctxts) {
                if (!c) {
                    throw std::runtime_error("multiple_steps_group: None in the context list");
                }
                hs.push_back(c->h);
            }
            py::gil_scoped_release nogil;
            check(tm_context_multiple_steps_group(hs.data(), static_cast<int>(hs.size()), n_steps));
        },
        py::arg("contexts"), py::arg("n_steps"));
    m.def(
        "debug_set_rowblock_min_k",
        [](const int min_atoms) { // A/B aid: forces-only launches over >= min_atoms atoms run the row-block kernel; -> the old value
            int previous = 0;
            check(tm_debug_set_rowblock_min_k(min_atoms, &previous));
            return previous;
        },
        py::arg("min_atoms"));
    m.def("debug_last_host_call_device_ms", []() { // device time of the evaluations of the last execute / execute_batch[_sparse] call (diagnostic)
        double ms = 0.0;
        check(tm_debug_last_host_call_device_ms(&ms));
        return ms;
    });
    m.def(
        "debug_set_same_frame_hint",
        [](const bool enabled) {
            int previous = 0;
            check(tm_debug_set_same_frame_hint(enabled ? 1 : 0, &previous));
            return previous != 0;
        },
        py::arg("enabled"));
    m.def(
        "debug_set_energy_memo",
        [](const bool enabled) {
            int previous = 0;
            check(tm_debug_set_energy_memo(enabled ? 1 : 0, &previous));
            return previous != 0;
        },
        py::arg("enabled"));
    m.def(
        "debug_set_merge_producers",
        [](const bool enabled) {
            int previous = 0;
            check(tm_debug_set_merge_producers(enabled ? 1 : 0, &previous));
            return previous != 0;
        },
        py::arg("enabled"));
    m.def(
        "debug_set_barostat_fast_path",
        [](const bool enabled) { // A/B aid: barostat attempts on the potential's current list (true) or reference-shaped (false); -> the old value
            int previous = 0;
            check(tm_debug_set_barostat_fast_path(enabled ? 1 : 0, &previous));
            return previous != 0;
        },
        py::arg("enabled"));
    m.def("debug_rowblock_available", []() { // does the loaded library carry the row-block kernel (the variant library of the parity tests)?
        int yes = 0;
        check(tm_debug_rowblock_available(&yes));
        return yes != 0;
    });
    m.def("debug_check_guards", []() { // -DTM_GUARD builds: violated guard zones so far; -1 in product builds
        int n = 0;
        check(tm_debug_check_guards(&n));
        return n;
    });
    m.def("profile_set_enabled", [](const bool enabled) { check(tm_profile_set_enabled(enabled ? 1 : 0)); }, py::arg("enabled"));
    m.def(
        "profile_read",
        [](const std::string &name) {
            double ms = 0;
            long long n = 0;
            check(tm_profile_read(name.c_str(), &ms, &n));
            return py::make_tuple(ms, n);
        },
        py::arg("name") = "nonbonded_tiles");
    m.def("profile_reset", []() { check(tm_profile_reset()); });
}

// Entries of the reference module outside the MI355X hot path (DESIGN.md "out of scope"): the names
// exist and fail loudly, by name.
const char *OUT_OF_SCOPE = R"py(
def _not_on_hot_path(name):
    class _Missing:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"custom_ops.{name} is outside the MI355X hot path of timemachine_amd (see DESIGN.md, 'out of scope')")
    _Missing.__name__ = _Missing.__qualname__ = name
    return _Missing

for _name in (
    "BDExchangeMove_f32", "BDExchangeMove_f64", "TIBDExchangeMove_f32", "TIBDExchangeMove_f64",
    "NonbondedMolEnergyPotential_f32", "NonbondedMolEnergyPotential_f64", "SegmentedSumExp_f32", "SegmentedSumExp_f64",
    "SegmentedWeightedRandomSampler_f32", "SegmentedWeightedRandomSampler_f64",
    "atom_by_atom_energies_f32", "atom_by_atom_energies_f64", "inner_and_outer_mols_f32", "inner_and_outer_mols_f64", "rmsd_align",
    "rotate_and_translate_mol_f32", "rotate_and_translate_mol_f64", "rotate_coords_f32", "rotate_coords_f64",
    "translations_inside_and_outside_sphere_host_f32", "translations_inside_and_outside_sphere_host_f64",
):
    globals()[_name] = _not_on_hot_path(_name)
del _name
)py";

} // namespace

PYBIND11_MODULE(custom_ops, m) {
    m.doc() = "timemachine_amd.lib.custom_ops: the reference's timemachine.lib.custom_ops surface (wrap_kernels.cpp) for the force-evaluation + "
              "Langevin-step hot path, bound onto the C ABI of libtimemachine_amd.so (hand-written HIP for gfx950). No CPU fallback.";
    py::register_exception<InvalidHardwareError>(m, "InvalidHardware");

    declare_potential(m);
    declare_bound_potential(m);
    declare_summed_potentials(m);
    declare_potentials<TM_F32>(m);
    declare_potentials<TM_F64>(m);
    declare_integrators_and_movers(m);
    declare_neighborlist<TM_F32>(m, "Neighborlist_f32");
    declare_neighborlist<TM_F64>(m, "Neighborlist_f64");
    declare_hilbert_sort(m);
    declare_context(m);
    declare_functions(m);
    py::exec(OUT_OF_SCOPE, m.attr("__dict__"));

    m.attr("FIXED_EXPONENT") = py::int_(static_cast<unsigned long long>(TM_FIXED_EXPONENT_VALUE)); // wrap_kernels.cpp:2144
    m.attr("BINDING") = "pybind11";
}
