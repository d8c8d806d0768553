// extern "C" surface of libtimemachine_amd.so (declared in include/timemachine_amd.h).
// Thin: argument marshalling + exception -> error-code translation.  No torch, no Python types.
#include "../../include/timemachine_amd.h"
#include "engine.hpp"
#include "profiler.hpp"

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <set>

using namespace tmamd;

struct tm_potential_s {
    std::shared_ptr<Potential> p;
};
struct tm_bound_potential_s {
    std::shared_ptr<BoundPotential> p;
};
struct tm_integrator_s {
    std::shared_ptr<Integrator> p;
};
struct tm_context_s {
    std::unique_ptr<Context> p;
};
struct tm_mover_s {
    std::shared_ptr<Mover> p;
};
struct tm_neighborlist_s {
    int precision;
    std::unique_ptr<Neighborlist<float>> f32;
    std::unique_ptr<Neighborlist<double>> f64;
    std::vector<std::vector<int>> last;
};
struct tm_hilbert_sort_s {
    std::unique_ptr<HilbertSort> p;
};

static thread_local std::string g_last_error;

// Streams of a stepped-together group of contexts (tm_context_multiple_steps_group) need hardware queues of their own; the HIP
// runtime creates GPU_MAX_HW_QUEUES (default 4) per process and reads the variable when it first touches the device.  Exported
// when the library is loaded -- before any HIP call made through this library -- unless the user chose a number; a C-ABI
// consumer that never imports the Python package gets the same behaviour as the Python binding (include/timemachine_amd.h).
__attribute__((constructor)) static void tm_export_hw_queue_count() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

// Threading contract of the C ABI: every entry point runs under the lock of the calling thread's CURRENT DEVICE.  The objects
// behind the handles keep host-side state between calls (pre-gathered inputs, piggy-backed tables, launch parities, uploaded plans)
// and live on one device; the reference's own contract is per object ("*Not* guaranteed to be thread-safe", cpp/src/potential.hpp:7)
// and its pybind11 layer holds the GIL through every call.  The compiled binding here releases the GIL around device calls, so
// this lock is what keeps two host threads from entering the same Potential / Context meanwhile -- while a process that drives
// SEVERAL GPUs from threads (one thread per device, hipSetDevice / tm_set_device in each) is not serialised across devices: a
// tm_context_multiple_steps call on device 0 no longer holds up device 1 (round 5: one process-wide lock).  Objects of one device
// must still not be entered from two threads at once -- which this lock enforces -- and a thread must have the object's device
// current, as HIP itself demands.  Recursive: entry points call each other.
static const int TM_MAX_LOCK_DEVICES = 64;
static std::recursive_mutex g_api_mutex[TM_MAX_LOCK_DEVICES + 1]; // [TM_MAX_LOCK_DEVICES]: no device (host-only entry points on a box without one)
static thread_local int g_lock_device_override = -1;               // tm_debug_set_thread_lock_device (CPU tests: no device to make current)
static std::recursive_mutex &api_mutex() {
    int dev = g_lock_device_override;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) {
            (void)hipGetLastError();
            dev = TM_MAX_LOCK_DEVICES;
        }
    }
    return g_api_mutex[(dev >= 0 && dev < TM_MAX_LOCK_DEVICES) ? dev : TM_MAX_LOCK_DEVICES];
}
#define TM_TRY                                                                                                         \
    std::lock_guard<std::recursive_mutex> tm_api_lock_(api_mutex());                                                   \
    try {
#define TM_CATCH                                                                                                       \
    }                                                                                                                  \
    catch (const InvalidHardware &e) {                                                                                 \
        g_last_error = e.what();                                                                                       \
        return TM_ERR_INVALID_HARDWARE;                                                                                \
    }                                                                                                                  \
    catch (const std::exception &e) {                                                                                  \
        g_last_error = e.what();                                                                                       \
        return TM_ERR_RUNTIME;                                                                                         \
    }                                                                                                                  \
    catch (...) {                                                                                                      \
        g_last_error = "unknown error";                                                                                \
        return TM_ERR_RUNTIME;                                                                                         \
    }                                                                                                                  \
    return TM_OK;

static void require(bool cond, const char *msg) {
    if (!cond)
        throw std::runtime_error(msg);
}

static i128 *as_i128(tm_int128 *u) { return reinterpret_cast<i128 *>(u); }
static_assert(sizeof(tm_int128) == sizeof(i128), "tm_int128 must be layout compatible with __int128");

template <template <typename> class Cls, typename... Args> static std::shared_ptr<Potential> make_by_precision(int precision, Args &&...args) {
    if (precision == TM_F32)
        return std::make_shared<Cls<float>>(std::forward<Args>(args)...);
    if (precision == TM_F64)
        return std::make_shared<Cls<double>>(std::forward<Args>(args)...);
    throw std::runtime_error("invalid precision");
}

template <typename F> static void with_all_pairs(tm_potential_t pot, F f) {
    if (auto a = std::dynamic_pointer_cast<NonbondedAllPairs<float>>(pot->p))
        f(*a);
    else if (auto b = std::dynamic_pointer_cast<NonbondedAllPairs<double>>(pot->p))
        f(*b);
    else
        throw std::runtime_error("unable to cast potential to NonbondedAllPairs");
}

extern "C" {

const char *tm_last_error(void) { return g_last_error.c_str(); }
const char *tm_version(void) { return "timemachine_amd 0.1 (gfx950)"; }

int tm_device_count(int *count) {
    TM_TRY
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        n = 0;
        (void)hipGetLastError();
    }
    *count = n;
    TM_CATCH
}

int tm_set_device(int device) {
    TM_TRY
    HIP_CHECK(hipSetDevice(device));
    TM_CATCH
}

int tm_device_synchronize(void) {
    TM_TRY
    HIP_CHECK(hipDeviceSynchronize());
    TM_CATCH
}

int tm_device_reset(void) {
    TM_TRY
    HIP_CHECK(hipDeviceReset());
    TM_CATCH
}

int tm_device_name(char *buf, size_t cap) {
    TM_TRY
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    TM_CATCH
}

double tm_fixed_to_float(uint64_t v) { return static_cast<double>(static_cast<long long>(v)) / static_cast<double>(TM_FIXED_EXPONENT_VALUE); }

int tm_energy_overflowed(const tm_int128 *u) {
    const i128 v = *reinterpret_cast<const i128 *>(u);
    return (v >= static_cast<i128>(LLONG_MAX) || v <= static_cast<i128>(LLONG_MIN)) ? 1 : 0;
}

double tm_energy_to_float(const tm_int128 *u) {
    if (tm_energy_overflowed(u))
        return std::numeric_limits<double>::quiet_NaN();
    const i128 v = *reinterpret_cast<const i128 *>(u);
    return static_cast<double>(static_cast<long long>(v)) / static_cast<double>(TM_FIXED_EXPONENT_VALUE);
}

// ---------------------------------------------------------------------------------------------------------
int tm_harmonic_bond_create(int precision, const int32_t *idxs, int n, tm_potential_t *out) {
    TM_TRY
    std::vector<int> v(idxs, idxs + static_cast<size_t>(n) * 2);
    *out = new tm_potential_s{make_by_precision<HarmonicBond>(precision, v)};
    TM_CATCH
}

int tm_harmonic_angle_create(int precision, const int32_t *idxs, int n, tm_potential_t *out) {
    TM_TRY
    std::vector<int> v(idxs, idxs + static_cast<size_t>(n) * 3);
    *out = new tm_potential_s{make_by_precision<HarmonicAngle>(precision, v)};
    TM_CATCH
}

int tm_periodic_torsion_create(int precision, const int32_t *idxs, int n, tm_potential_t *out) {
    TM_TRY
    std::vector<int> v(idxs, idxs + static_cast<size_t>(n) * 4);
    *out = new tm_potential_s{make_by_precision<PeriodicTorsion>(precision, v)};
    TM_CATCH
}

int tm_nonbonded_all_pairs_create(
    int precision, int num_atoms, double beta, double cutoff, const int32_t *atom_idxs, int num_atom_idxs, int disable_hilbert_sort,
    double nblist_padding, tm_potential_t *out) {
    TM_TRY
    std::optional<std::vector<int>> idxs;
    if (atom_idxs != nullptr) {
        idxs.emplace(atom_idxs, atom_idxs + num_atom_idxs);
    }
    *out = new tm_potential_s{make_by_precision<NonbondedAllPairs>(precision, num_atoms, beta, cutoff, idxs, disable_hilbert_sort != 0, nblist_padding)};
    TM_CATCH
}

int tm_nonbonded_pair_list_create(
    int precision, int negated, const int32_t *pair_idxs, int num_pairs, const double *scales, int num_scales, double beta, double cutoff,
    tm_potential_t *out) {
    TM_TRY
    std::vector<int> p(pair_idxs, pair_idxs + static_cast<size_t>(num_pairs) * 2);
    std::vector<double> s(scales, scales + static_cast<size_t>(num_scales) * 2);
    std::shared_ptr<Potential> pot;
    if (precision == TM_F32) {
        if (negated)
            pot = std::make_shared<NonbondedPairList<float, true>>(p, s, beta, cutoff);
        else
            pot = std::make_shared<NonbondedPairList<float, false>>(p, s, beta, cutoff);
    } else if (precision == TM_F64) {
        if (negated)
            pot = std::make_shared<NonbondedPairList<double, true>>(p, s, beta, cutoff);
        else
            pot = std::make_shared<NonbondedPairList<double, false>>(p, s, beta, cutoff);
    } else {
        throw std::runtime_error("invalid precision");
    }
    *out = new tm_potential_s{pot};
    TM_CATCH
}

int tm_nonbonded_interaction_group_create(
    int precision, int num_atoms, const int32_t *row_atom_idxs, int num_rows, const int32_t *col_atom_idxs, int num_cols, double beta,
    double cutoff, int disable_hilbert_sort, double nblist_padding, tm_potential_t *out) {
    TM_TRY
    std::vector<int> rows(row_atom_idxs, row_atom_idxs + num_rows);
    std::vector<int> cols;
    if (col_atom_idxs != nullptr) {
        cols.assign(col_atom_idxs, col_atom_idxs + num_cols);
    } else { // every atom that is not a row atom (wrap_kernels.cpp:1519-1522)
        const std::set<int> row_set(rows.begin(), rows.end());
        for (int i = 0; i < num_atoms; i++) {
            if (!row_set.count(i)) {
                cols.push_back(i);
            }
        }
    }
    *out = new tm_potential_s{make_by_precision<NonbondedInteractionGroup>(precision, num_atoms, rows, cols, beta, cutoff, disable_hilbert_sort != 0, nblist_padding)};
    TM_CATCH
}

int tm_nonbonded_interaction_group_set_atom_idxs(
    tm_potential_t pot, const int32_t *row_atom_idxs, int num_rows, const int32_t *col_atom_idxs, int num_cols) {
    TM_TRY
    std::vector<int> rows(row_atom_idxs, row_atom_idxs + num_rows), cols(col_atom_idxs, col_atom_idxs + num_cols);
    if (auto a = std::dynamic_pointer_cast<NonbondedInteractionGroup<float>>(pot->p))
        a->set_atom_idxs(rows, cols);
    else if (auto b = std::dynamic_pointer_cast<NonbondedInteractionGroup<double>>(pot->p))
        b->set_atom_idxs(rows, cols);
    else
        throw std::runtime_error("unable to cast potential to NonbondedInteractionGroup");
    TM_CATCH
}

int tm_nonbonded_pair_list_precomputed_create(
    int precision, const int32_t *pair_idxs, int num_pairs, double beta, double cutoff, tm_potential_t *out) {
    TM_TRY
    std::vector<int> p(pair_idxs, pair_idxs + static_cast<size_t>(num_pairs) * 2);
    *out = new tm_potential_s{make_by_precision<NonbondedPairListPrecomputed>(precision, p, beta, cutoff)};
    TM_CATCH
}

int tm_flat_bottom_bond_create(int precision, int log_form, const int32_t *bond_idxs, int n, double beta, tm_potential_t *out) {
    TM_TRY
    std::vector<int> v(bond_idxs, bond_idxs + static_cast<size_t>(n) * 2);
    std::shared_ptr<Potential> pot;
    if (precision == TM_F32) {
        if (log_form)
            pot = std::make_shared<FlatBottomBond<float, true>>(v, beta);
        else
            pot = std::make_shared<FlatBottomBond<float, false>>(v, beta);
    } else if (precision == TM_F64) {
        if (log_form)
            pot = std::make_shared<FlatBottomBond<double, true>>(v, beta);
        else
            pot = std::make_shared<FlatBottomBond<double, false>>(v, beta);
    } else {
        throw std::runtime_error("invalid precision");
    }
    *out = new tm_potential_s{pot};
    TM_CATCH
}

int tm_centroid_restraint_create(
    int precision, const int32_t *group_a_idxs, int num_a, const int32_t *group_b_idxs, int num_b, double kb, double b0,
    tm_potential_t *out) {
    TM_TRY
    std::vector<int> a(group_a_idxs, group_a_idxs + num_a), b(group_b_idxs, group_b_idxs + num_b);
    *out = new tm_potential_s{make_by_precision<CentroidRestraint>(precision, a, b, kb, b0)};
    TM_CATCH
}

int tm_chiral_atom_restraint_create(int precision, const int32_t *idxs, int n, tm_potential_t *out) {
    TM_TRY
    std::vector<int> v(idxs, idxs + static_cast<size_t>(n) * 4);
    *out = new tm_potential_s{make_by_precision<ChiralAtomRestraint>(precision, v)};
    TM_CATCH
}

int tm_chiral_bond_restraint_create(int precision, const int32_t *idxs, int n, const int32_t *signs, int num_signs, tm_potential_t *out) {
    TM_TRY
    std::vector<int> v(idxs, idxs + static_cast<size_t>(n) * 4), sg(signs, signs + num_signs);
    *out = new tm_potential_s{make_by_precision<ChiralBondRestraint>(precision, v, sg)};
    TM_CATCH
}

int tm_summed_potential_create(
    const tm_potential_t *potentials, int n, const int32_t *params_sizes, int n_sizes, int parallel, tm_potential_t *out) {
    TM_TRY
    std::vector<std::shared_ptr<Potential>> pots;
    for (int i = 0; i < n; i++)
        pots.push_back(potentials[i]->p);
    std::vector<int> sizes(params_sizes, params_sizes + n_sizes);
    *out = new tm_potential_s{std::make_shared<SummedPotential>(pots, sizes, parallel != 0)};
    TM_CATCH
}

int tm_fanout_summed_potential_create(const tm_potential_t *potentials, int n, int parallel, tm_potential_t *out) {
    TM_TRY
    std::vector<std::shared_ptr<Potential>> pots;
    for (int i = 0; i < n; i++)
        pots.push_back(potentials[i]->p);
    *out = new tm_potential_s{std::make_shared<FanoutSummedPotential>(pots, parallel != 0)};
    TM_CATCH
}

int tm_potential_destroy(tm_potential_t pot) {
    TM_TRY
    delete pot;
    TM_CATCH
}

int tm_potential_get_children(tm_potential_t pot, tm_potential_t *out, int cap, int *count) {
    TM_TRY
    const std::vector<std::shared_ptr<Potential>> *kids = nullptr;
    if (auto s = std::dynamic_pointer_cast<SummedPotential>(pot->p))
        kids = &s->get_potentials();
    else if (auto f = std::dynamic_pointer_cast<FanoutSummedPotential>(pot->p))
        kids = &f->get_potentials();
    else
        throw std::runtime_error("potential has no children");
    *count = kids->size();
    for (int i = 0; i < cap && i < static_cast<int>(kids->size()); i++)
        out[i] = new tm_potential_s{(*kids)[i]};
    TM_CATCH
}

int tm_nonbonded_all_pairs_set_atom_idxs(tm_potential_t pot, const int32_t *atom_idxs, int n) {
    TM_TRY
    std::vector<int> v(atom_idxs, atom_idxs + n);
    with_all_pairs(pot, [&](auto &p) { p.set_atom_idxs(v); });
    TM_CATCH
}

int tm_nonbonded_all_pairs_get_num_atom_idxs(tm_potential_t pot, int *count) {
    TM_TRY
    with_all_pairs(pot, [&](auto &p) { *count = p.get_num_atom_idxs(); });
    TM_CATCH
}

int tm_nonbonded_all_pairs_get_atom_idxs(tm_potential_t pot, int32_t *out, int cap) {
    TM_TRY
    with_all_pairs(pot, [&](auto &p) {
        std::vector<int> v = p.get_atom_idxs();
        for (int i = 0; i < cap && i < static_cast<int>(v.size()); i++)
            out[i] = v[i];
    });
    TM_CATCH
}

int tm_nonbonded_all_pairs_debug_timing(tm_potential_t pot, long long *out, int cap, int *n) {
    TM_TRY
    with_all_pairs(pot, [&](auto &p) {
        std::vector<long long> v = p.debug_timing();
        *n = static_cast<int>(v.size());
        for (int i = 0; i < cap && i < static_cast<int>(v.size()); i++)
            out[i] = v[i];
    });
    TM_CATCH
}

int tm_nonbonded_all_pairs_get_tile_count(tm_potential_t pot, unsigned int *count) {
    TM_TRY
    with_all_pairs(pot, [&](auto &p) { *count = p.num_tile_ixns(); });
    TM_CATCH
}

int tm_nonbonded_all_pairs_get_build_count(tm_potential_t pot, unsigned int *count) {
    TM_TRY
    with_all_pairs(pot, [&](auto &p) { *count = p.num_builds(); });
    TM_CATCH
}

int tm_nonbonded_all_pairs_get_merged_stats(tm_potential_t pot, long long *calls, unsigned int *tiles, unsigned int *builds) {
    TM_TRY
    with_all_pairs(pot, [&](auto &p) { p.merged_stats(calls, tiles, builds); });
    TM_CATCH
}

int tm_nonbonded_all_pairs_get_memo_stats(tm_potential_t pot, long long *evaluations, long long *skipped) {
    TM_TRY
    with_all_pairs(pot, [&](auto &p) { p.memo_stats(evaluations, skipped); });
    TM_CATCH
}

int tm_nonbonded_all_pairs_get_same_frame_skips(tm_potential_t pot, long long *skips) {
    TM_TRY
    with_all_pairs(pot, [&](auto &p) { *skips = p.same_frame_skips(); });
    TM_CATCH
}

// ---------------------------------------------------------------------------------------------------------
int tm_potential_execute(
    tm_potential_t pot, int N, int P, const double *coords, const double *params, const double *box, uint64_t *du_dx, uint64_t *du_dp,
    tm_int128 *u) {
    TM_TRY
    pot->p->execute_host(N, P, coords, params, box, reinterpret_cast<u64 *>(du_dx), reinterpret_cast<u64 *>(du_dp), as_i128(u));
    TM_CATCH
}

int tm_potential_execute_batch(
    tm_potential_t pot, int C, int N, int Pb, int P, const double *coords, const double *params, const double *boxes, uint64_t *du_dx,
    uint64_t *du_dp, tm_int128 *u) {
    TM_TRY
    pot->p->execute_batch_host(C, N, Pb, P, coords, params, boxes, reinterpret_cast<u64 *>(du_dx), reinterpret_cast<u64 *>(du_dp), as_i128(u));
    TM_CATCH
}

int tm_potential_execute_batch_sparse(
    tm_potential_t pot, int coords_size, int N, int params_size, int P, int batch_size, const uint32_t *cidx, const uint32_t *pidx,
    const double *coords, const double *params, const double *boxes, uint64_t *du_dx, uint64_t *du_dp, tm_int128 *u) {
    TM_TRY
    for (int i = 0; i < batch_size; i++) {
        require(cidx[i] < static_cast<uint32_t>(coords_size), "coords_batch_idxs contains an index that is out of bounds");
        require(pidx[i] < static_cast<uint32_t>(params_size), "params_batch_idxs contains an index that is out of bounds");
    }
    pot->p->execute_batch_sparse_host(
        coords_size, N, params_size, P, batch_size, cidx, pidx, coords, params, boxes, reinterpret_cast<u64 *>(du_dx),
        reinterpret_cast<u64 *>(du_dp), as_i128(u));
    TM_CATCH
}

int tm_potential_execute_f64(
    tm_potential_t pot, int N, int P, const double *coords, const double *params, const double *box, double *du_dx, double *du_dp, double *u) {
    TM_TRY
    pot->p->execute_host_f64(1, N, 1, P, -1, nullptr, nullptr, coords, params, box, du_dx, du_dp, u);
    TM_CATCH
}

int tm_potential_execute_batch_f64(
    tm_potential_t pot, int C, int N, int Pb, int P, const double *coords, const double *params, const double *boxes, double *du_dx,
    double *du_dp, double *u) {
    TM_TRY
    pot->p->execute_host_f64(C, N, Pb, P, -1, nullptr, nullptr, coords, params, boxes, du_dx, du_dp, u);
    TM_CATCH
}

int tm_potential_execute_batch_sparse_f64(
    tm_potential_t pot, int coords_size, int N, int params_size, int P, int batch_size, const uint32_t *cidx, const uint32_t *pidx,
    const double *coords, const double *params, const double *boxes, double *du_dx, double *du_dp, double *u) {
    TM_TRY
    require(batch_size >= 0, "batch_size must not be negative");
    for (int i = 0; i < batch_size; i++) {
        require(cidx[i] < static_cast<uint32_t>(coords_size), "coords_batch_idxs contains an index that is out of bounds");
        require(pidx[i] < static_cast<uint32_t>(params_size), "params_batch_idxs contains an index that is out of bounds");
    }
    pot->p->execute_host_f64(coords_size, N, params_size, P, batch_size, cidx, pidx, coords, params, boxes, du_dx, du_dp, u);
    TM_CATCH
}

int tm_potential_du_dp_fixed_to_float(tm_potential_t pot, int N, int P, const uint64_t *du_dp, double *out) {
    TM_TRY
    pot->p->du_dp_fixed_to_float(N, P, reinterpret_cast<const u64 *>(du_dp), out);
    TM_CATCH
}

int tm_potential_execute_device(
    tm_potential_t pot, int N, int P, const double *d_x, const double *d_p, const double *d_box, uint64_t *d_du_dx, uint64_t *d_du_dp,
    tm_int128 *d_u, void *hip_stream) {
    TM_TRY
    pot->p->execute_device(
        N, P, d_x, d_p, d_box, reinterpret_cast<u64 *>(d_du_dx), reinterpret_cast<u64 *>(d_du_dp), as_i128(d_u),
        static_cast<hipStream_t>(hip_stream));
    TM_CATCH
}

// ---------------------------------------------------------------------------------------------------------
int tm_bound_potential_create(tm_potential_t pot, const double *params, int P, tm_bound_potential_t *out) {
    TM_TRY
    std::vector<double> v(params, params + P);
    *out = new tm_bound_potential_s{std::make_shared<BoundPotential>(pot->p, v)};
    TM_CATCH
}

int tm_bound_potential_destroy(tm_bound_potential_t bp) {
    TM_TRY
    delete bp;
    TM_CATCH
}

int tm_bound_potential_set_params(tm_bound_potential_t bp, const double *params, int P) {
    TM_TRY
    std::vector<double> v(params, params + P);
    bp->p->set_params(v);
    TM_CATCH
}

int tm_bound_potential_size(tm_bound_potential_t bp, int *size) {
    TM_TRY
    *size = bp->p->size;
    TM_CATCH
}

int tm_bound_potential_get_potential(tm_bound_potential_t bp, tm_potential_t *out) {
    TM_TRY
    *out = new tm_potential_s{bp->p->potential};
    TM_CATCH
}

int tm_bound_potential_execute(tm_bound_potential_t bp, int N, const double *coords, const double *box, uint64_t *du_dx, tm_int128 *u) {
    TM_TRY
    bp->p->execute_host(N, coords, box, reinterpret_cast<u64 *>(du_dx), as_i128(u));
    TM_CATCH
}

int tm_bound_potential_execute_f64(tm_bound_potential_t bp, int N, const double *coords, const double *box, double *du_dx, double *u) {
    TM_TRY
    bp->p->execute_host_f64(1, N, coords, box, du_dx, u);
    TM_CATCH
}

int tm_bound_potential_execute_batch_f64(tm_bound_potential_t bp, int C, int N, const double *coords, const double *boxes, double *du_dx, double *u) {
    TM_TRY
    bp->p->execute_host_f64(C, N, coords, boxes, du_dx, u);
    TM_CATCH
}

int tm_bound_potential_execute_batch(
    tm_bound_potential_t bp, int C, int N, const double *coords, const double *boxes, uint64_t *du_dx, tm_int128 *u) {
    TM_TRY
    bp->p->execute_batch_host(C, N, coords, boxes, reinterpret_cast<u64 *>(du_dx), as_i128(u));
    TM_CATCH
}

// ---------------------------------------------------------------------------------------------------------
int tm_langevin_integrator_create(
    const double *masses, int N, double temperature, double dt, double friction, int seed, tm_integrator_t *out) {
    TM_TRY
    *out = new tm_integrator_s{std::make_shared<LangevinIntegrator<float>>(N, masses, temperature, dt, friction, seed)};
    TM_CATCH
}

int tm_velocity_verlet_integrator_create(double dt, const double *cbs, int N, tm_integrator_t *out) {
    TM_TRY
    *out = new tm_integrator_s{std::make_shared<VelocityVerletIntegrator>(N, dt, cbs)};
    TM_CATCH
}

int tm_integrator_destroy(tm_integrator_t intg) {
    TM_TRY
    delete intg;
    TM_CATCH
}

int tm_context_create(
    const double *x0, const double *v0, const double *box, int N, tm_integrator_t intg, const tm_bound_potential_t *bps, int num_bps,
    tm_context_t *out) {
    TM_TRY
    std::vector<std::shared_ptr<BoundPotential>> v;
    for (int i = 0; i < num_bps; i++)
        v.push_back(bps[i]->p);
    std::vector<std::shared_ptr<Mover>> movers;
    *out = new tm_context_s{std::make_unique<Context>(N, x0, v0, box, intg->p, v, movers)};
    TM_CATCH
}

int tm_context_create_with_movers(
    const double *x0, const double *v0, const double *box, int N, tm_integrator_t intg, const tm_bound_potential_t *bps, int num_bps,
    const tm_mover_t *movers, int num_movers, tm_context_t *out) {
    TM_TRY
    std::vector<std::shared_ptr<BoundPotential>> v;
    for (int i = 0; i < num_bps; i++)
        v.push_back(bps[i]->p);
    std::vector<std::shared_ptr<Mover>> mv;
    for (int i = 0; i < num_movers; i++)
        mv.push_back(movers[i]->p);
    *out = new tm_context_s{std::make_unique<Context>(N, x0, v0, box, intg->p, v, mv)};
    TM_CATCH
}

// ---------------------------------------------------------------------------------------------------------
static MonteCarloBarostat<float> &as_barostat(tm_mover_t m) {
    auto b = std::dynamic_pointer_cast<MonteCarloBarostat<float>>(m->p);
    if (!b) {
        throw std::runtime_error("unable to cast mover to MonteCarloBarostat");
    }
    return *b;
}

int tm_monte_carlo_barostat_create(
    int N, double pressure, double temperature, const int32_t *group_atom_idxs, const int32_t *group_offsets, int num_groups,
    int interval, const tm_bound_potential_t *bps, int num_bps, int seed, int adaptive_scaling_enabled,
    double initial_volume_scale_factor, tm_mover_t *out) {
    TM_TRY
    std::vector<std::vector<int>> groups(num_groups);
    for (int g = 0; g < num_groups; g++) {
        groups[g].assign(group_atom_idxs + group_offsets[g], group_atom_idxs + group_offsets[g + 1]);
    }
    std::vector<std::shared_ptr<BoundPotential>> v;
    for (int i = 0; i < num_bps; i++)
        v.push_back(bps[i]->p);
    *out = new tm_mover_s{std::make_shared<MonteCarloBarostat<float>>(
        N, pressure, temperature, groups, interval, v, seed, adaptive_scaling_enabled != 0, initial_volume_scale_factor)};
    TM_CATCH
}

int tm_mover_destroy(tm_mover_t m) {
    TM_TRY
    delete m;
    TM_CATCH
}
int tm_mover_set_interval(tm_mover_t m, int interval) {
    TM_TRY
    m->p->set_interval(interval);
    TM_CATCH
}
int tm_mover_get_interval(tm_mover_t m, int *interval) {
    TM_TRY
    *interval = m->p->get_interval();
    TM_CATCH
}
int tm_mover_set_step(tm_mover_t m, int step) {
    TM_TRY
    m->p->set_step(step);
    TM_CATCH
}
int tm_mover_move(tm_mover_t m, int N, const double *x, const double *box, double *x_out, double *box_out) {
    TM_TRY
    m->p->move_host(N, x, box, x_out, box_out);
    TM_CATCH
}
int tm_barostat_set_volume_scale_factor(tm_mover_t m, double f) {
    TM_TRY
    as_barostat(m).set_volume_scale_factor(f);
    TM_CATCH
}
int tm_barostat_get_volume_scale_factor(tm_mover_t m, double *f) {
    TM_TRY
    *f = as_barostat(m).get_volume_scale_factor();
    TM_CATCH
}
int tm_barostat_set_adaptive_scaling(tm_mover_t m, int enabled) {
    TM_TRY
    as_barostat(m).set_adaptive_scaling(enabled != 0);
    TM_CATCH
}
int tm_barostat_get_adaptive_scaling(tm_mover_t m, int *enabled) {
    TM_TRY
    *enabled = as_barostat(m).get_adaptive_scaling() ? 1 : 0;
    TM_CATCH
}
int tm_barostat_set_pressure(tm_mover_t m, double pressure) {
    TM_TRY
    as_barostat(m).set_pressure(pressure);
    TM_CATCH
}
int tm_barostat_get_counters(tm_mover_t m, int *accepted, int *attempted) {
    TM_TRY
    as_barostat(m).get_counters(accepted, attempted);
    TM_CATCH
}
int tm_barostat_get_attempt_paths(tm_mover_t m, long long *attempts, long long *fast) {
    TM_TRY
    if (auto b = std::dynamic_pointer_cast<MonteCarloBarostat<float>>(m->p)) {
        b->get_attempt_paths(attempts, fast);
    } else if (auto c = std::dynamic_pointer_cast<MonteCarloBarostat<double>>(m->p)) {
        c->get_attempt_paths(attempts, fast);
    } else {
        throw std::runtime_error("not a MonteCarloBarostat");
    }
    TM_CATCH
}

int tm_context_destroy(tm_context_t ctxt) {
    TM_TRY
    delete ctxt;
    TM_CATCH
}

int tm_context_num_atoms(tm_context_t ctxt, int *N) {
    TM_TRY
    *N = ctxt->p->num_atoms();
    TM_CATCH
}

int tm_context_step(tm_context_t ctxt) {
    TM_TRY
    ctxt->p->step();
    TM_CATCH
}
int tm_context_initialize(tm_context_t ctxt) {
    TM_TRY
    ctxt->p->initialize();
    TM_CATCH
}
int tm_context_finalize(tm_context_t ctxt) {
    TM_TRY
    ctxt->p->finalize();
    TM_CATCH
}
int tm_context_multiple_steps(tm_context_t ctxt, int n_steps, int n_samples, double *xs, double *boxes) {
    TM_TRY
    ctxt->p->multiple_steps(n_steps, n_samples, xs, boxes);
    TM_CATCH
}
int tm_context_multiple_steps_group(const tm_context_t *ctxts, int n_ctxts, int n_steps) {
    TM_TRY
    require(n_ctxts >= 0 && (n_ctxts == 0 || ctxts != nullptr), "multiple_steps_group: bad context list");
    std::vector<Context *> group;
    for (int i = 0; i < n_ctxts; i++) {
        require(ctxts[i] != nullptr, "multiple_steps_group: null context");
        group.push_back(ctxts[i]->p.get());
    }
    Context::multiple_steps_group(group, n_steps);
    TM_CATCH
}
int tm_context_last_multiple_steps_ms(tm_context_t ctxt, double *ms) {
    TM_TRY
    *ms = ctxt->p->last_multiple_steps_ms();
    TM_CATCH
}
int tm_context_setup_local_md(tm_context_t ctxt, double temperature, int freeze_reference) {
    TM_TRY
    ctxt->p->setup_local_md(temperature, freeze_reference != 0);
    TM_CATCH
}
static int local_md_num_samples(int n_steps, int store_x_interval) {
    // wrap_kernels.cpp:408-426
    if (n_steps <= 0) {
        throw std::runtime_error("local steps must be at least one");
    }
    if (store_x_interval < 0) {
        throw std::runtime_error("store_x_interval must be greater than or equal to zero");
    }
    const int x_interval = store_x_interval == 0 ? n_steps : store_x_interval;
    return n_steps / x_interval;
}
int tm_context_multiple_steps_local(
    tm_context_t ctxt, int n_steps, const int *local_idxs, int num_local_idxs, int store_x_interval, double radius, double k, int seed,
    double *xs, double *boxes) {
    TM_TRY
    const int n_samples = local_md_num_samples(n_steps, store_x_interval);
    verify_local_md_parameters(radius, k);
    const std::vector<int> idxs(local_idxs, local_idxs + num_local_idxs);
    verify_atom_idxs(ctxt->p->num_atoms(), idxs);
    ctxt->p->multiple_steps_local(n_steps, idxs, n_samples, radius, k, seed, xs, boxes);
    TM_CATCH
}
int tm_context_multiple_steps_local_selection(
    tm_context_t ctxt, int n_steps, int reference_idx, const int *selection_idxs, int num_selection_idxs, int store_x_interval,
    double radius, double k, double *xs, double *boxes) {
    TM_TRY
    const int n_samples = local_md_num_samples(n_steps, store_x_interval);
    verify_local_md_parameters(radius, k);
    const int N = ctxt->p->num_atoms();
    if (reference_idx < 0 || reference_idx >= N) {
        throw std::runtime_error("reference idx must be at least 0 and less than " + std::to_string(N));
    }
    const std::vector<int> idxs(selection_idxs, selection_idxs + num_selection_idxs);
    verify_atom_idxs(N, idxs);
    if (std::find(idxs.begin(), idxs.end(), reference_idx) != idxs.end()) {
        throw std::runtime_error("reference idx must not be in selection idxs");
    }
    ctxt->p->multiple_steps_local_selection(n_steps, reference_idx, idxs, n_samples, radius, k, xs, boxes);
    TM_CATCH
}
int tm_context_local_md_last_selection(tm_context_t ctxt, int *reference_idx, unsigned int *free_idxs) {
    TM_TRY
    *reference_idx = ctxt->p->local_md_last_reference();
    const std::vector<unsigned int> f = ctxt->p->local_md_last_free_idxs();
    std::copy(f.begin(), f.end(), free_idxs);
    TM_CATCH
}
int tm_context_get_x_t(tm_context_t ctxt, double *out) {
    TM_TRY
    ctxt->p->get_x_t(out);
    TM_CATCH
}
int tm_context_get_v_t(tm_context_t ctxt, double *out) {
    TM_TRY
    ctxt->p->get_v_t(out);
    TM_CATCH
}
int tm_context_get_box(tm_context_t ctxt, double *out) {
    TM_TRY
    ctxt->p->get_box(out);
    TM_CATCH
}
int tm_context_set_x_t(tm_context_t ctxt, const double *in) {
    TM_TRY
    ctxt->p->set_x_t(in);
    TM_CATCH
}
int tm_context_set_v_t(tm_context_t ctxt, const double *in) {
    TM_TRY
    ctxt->p->set_v_t(in);
    TM_CATCH
}
int tm_context_set_box(tm_context_t ctxt, const double *in) {
    TM_TRY
    ctxt->p->set_box(in);
    TM_CATCH
}

// ---------------------------------------------------------------------------------------------------------
#define NB_DISPATCH(nb, expr)                                                                                          \
    do {                                                                                                               \
        if ((nb)->precision == TM_F32) {                                                                               \
            auto &L = *(nb)->f32;                                                                                      \
            expr;                                                                                                      \
        } else {                                                                                                       \
            auto &L = *(nb)->f64;                                                                                      \
            expr;                                                                                                      \
        }                                                                                                              \
    } while (0)

int tm_neighborlist_create(int precision, int N, tm_neighborlist_t *out) {
    TM_TRY
    std::unique_ptr<tm_neighborlist_s> h(new tm_neighborlist_s());
    h->precision = precision;
    if (precision == TM_F32)
        h->f32.reset(new Neighborlist<float>(N));
    else if (precision == TM_F64)
        h->f64.reset(new Neighborlist<double>(N));
    else
        throw std::runtime_error("invalid precision");
    *out = h.release();
    TM_CATCH
}

int tm_neighborlist_destroy(tm_neighborlist_t nb) {
    TM_TRY
    delete nb;
    TM_CATCH
}

int tm_neighborlist_get_nblist(
    tm_neighborlist_t nb, int N, const double *coords, const double *box, double cutoff, int *num_row_blocks, int *total_atoms) {
    TM_TRY
    NB_DISPATCH(nb, nb->last = L.get_nblist_host(N, coords, box, cutoff));
    *num_row_blocks = nb->last.size();
    int total = 0;
    for (auto &l : nb->last)
        total += l.size();
    *total_atoms = total;
    TM_CATCH
}

int tm_neighborlist_copy_nblist(tm_neighborlist_t nb, int32_t *offsets, int32_t *atoms) {
    TM_TRY
    int off = 0;
    for (size_t r = 0; r < nb->last.size(); r++) {
        offsets[r] = off;
        for (int a : nb->last[r])
            atoms[off++] = a;
    }
    offsets[nb->last.size()] = off;
    TM_CATCH
}

int tm_neighborlist_compute_block_bounds(
    tm_neighborlist_t nb, int N, const double *coords, const double *box, int block_size, double *ctrs, double *exts) {
    TM_TRY
    require(block_size == 32, "Block size must be 32.");
    NB_DISPATCH(nb, L.compute_block_bounds_host(N, coords, box, ctrs, exts));
    TM_CATCH
}

int tm_neighborlist_set_row_idxs(tm_neighborlist_t nb, const uint32_t *idxs, int count) {
    TM_TRY
    std::vector<unsigned int> v(idxs, idxs + count);
    NB_DISPATCH(nb, L.set_row_idxs(v));
    TM_CATCH
}
int tm_neighborlist_reset_row_idxs(tm_neighborlist_t nb) {
    TM_TRY
    NB_DISPATCH(nb, L.reset_row_idxs());
    TM_CATCH
}
int tm_neighborlist_resize(tm_neighborlist_t nb, int size) {
    TM_TRY
    NB_DISPATCH(nb, L.resize(size));
    TM_CATCH
}
int tm_neighborlist_get_tile_ixn_count(tm_neighborlist_t nb, unsigned int *count) {
    TM_TRY
    NB_DISPATCH(nb, *count = L.num_tile_ixns());
    TM_CATCH
}
int tm_neighborlist_get_max_ixn_count(tm_neighborlist_t nb, int *count) {
    TM_TRY
    NB_DISPATCH(nb, *count = L.max_ixn_count());
    TM_CATCH
}
int tm_neighborlist_get_num_row_idxs(tm_neighborlist_t nb, int *count) {
    TM_TRY
    NB_DISPATCH(nb, *count = L.get_num_row_idxs());
    TM_CATCH
}

// ---------------------------------------------------------------------------------------------------------
int tm_hilbert_sort_create(int size, tm_hilbert_sort_t *out) {
    TM_TRY
    *out = new tm_hilbert_sort_s{std::make_unique<HilbertSort>(size)};
    TM_CATCH
}
int tm_hilbert_sort_destroy(tm_hilbert_sort_t hs) {
    TM_TRY
    delete hs;
    TM_CATCH
}
int tm_hilbert_sort_sort(tm_hilbert_sort_t hs, int N, const double *coords, const double *box, uint32_t *perm) {
    TM_TRY
    std::vector<unsigned int> p = hs->p->sort_host(N, coords, box);
    for (int i = 0; i < N; i++)
        perm[i] = p[i];
    TM_CATCH
}
int tm_hilbert_lut(uint32_t *out) {
    TM_TRY
    const std::vector<unsigned int> &t = HilbertSort::lut();
    for (size_t i = 0; i < t.size(); i++)
        out[i] = t[i];
    TM_CATCH
}

// ---------------------------------------------------------------------------------------------------------
int tm_debug_set_box_scaling_reuse(int enabled) {
    TM_TRY
    g_box_scaling_reuse = enabled != 0;
    TM_CATCH
}
int tm_debug_set_static_list_max_k(int max_atoms, int *previous) {
    TM_TRY
    if (previous) {
        *previous = g_static_list_max_k;
    }
    g_static_list_max_k = max_atoms;
    TM_CATCH
}
int tm_debug_set_rowblock_min_k(int min_atoms, int *previous) {
    TM_TRY
    if (previous) {
        *previous = g_rowblock_min_k;
    }
    require(g_rowblock_built || min_atoms == std::numeric_limits<int>::max(),
            "the row-block kernel is not built into this library (load the variant libtimemachine_amd_rowblock.so: TM_AMD_LIB)");
    g_rowblock_min_k = min_atoms;
    TM_CATCH
}
int tm_debug_set_thread_lock_device(int device) {
    g_lock_device_override = device; // (no lock: it chooses the lock)
    return TM_OK;
}
int tm_debug_hold_api_lock(int milliseconds, int *max_concurrent) {
    static std::atomic<int> inside{0}, seen{0};
    TM_TRY
    const int now = ++inside;
    int prev = seen.load();
    while (now > prev && !seen.compare_exchange_weak(prev, now)) {
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(milliseconds));
    --inside;
    if (max_concurrent) {
        *max_concurrent = seen.load();
    }
    if (milliseconds < 0) {
        seen.store(0);
    }
    TM_CATCH
}
int tm_debug_last_host_call_device_ms(double *ms) {
    TM_TRY
    *ms = g_last_host_call_device_ms;
    TM_CATCH
}
int tm_debug_set_same_frame_hint(int enabled, int *previous) {
    TM_TRY
    if (previous) {
        *previous = g_same_frame_hint ? 1 : 0;
    }
    g_same_frame_hint = enabled != 0;
    TM_CATCH
}
int tm_debug_set_energy_memo(int enabled, int *previous) {
    TM_TRY
    if (previous) {
        *previous = g_energy_memo ? 1 : 0;
    }
    g_energy_memo = enabled != 0;
    TM_CATCH
}
int tm_debug_set_merge_producers(int enabled, int *previous) {
    TM_TRY
    if (previous) {
        *previous = g_merge_producers ? 1 : 0;
    }
    g_merge_producers = enabled != 0;
    TM_CATCH
}
int tm_debug_set_barostat_fast_path(int enabled, int *previous) {
    TM_TRY
    if (previous) {
        *previous = g_barostat_fast_path ? 1 : 0;
    }
    g_barostat_fast_path = enabled != 0;
    TM_CATCH
}
int tm_debug_rowblock_available(int *available) {
    TM_TRY
    *available = g_rowblock_built ? 1 : 0;
    TM_CATCH
}
int tm_profile_set_enabled(int enabled) {
    TM_TRY
    Profiler::get().set_enabled(enabled != 0);
    TM_CATCH
}
int tm_profile_read(const char *kernel_name, double *total_ms, long long *launches) {
    TM_TRY
    Profiler::get().read(kernel_name, total_ms, launches);
    TM_CATCH
}
int tm_profile_reset(void) {
    TM_TRY
    Profiler::get().reset();
    TM_CATCH
}
int tm_es_force_table(double beta, double *out) {
    TM_TRY
    es_force_table_host(beta, out);
    TM_CATCH
}
int tm_es_energy_table(double beta, double *out) {
    TM_TRY
    es_energy_table_host(beta, out);
    TM_CATCH
}
int tm_hrex_run_neighbor_swaps(
    int n_replicas, int n_states, const int64_t *replica_idx_by_state, int n_pairs, const int64_t *neighbor_pairs, const double *log_q_kl,
    int n_attempts, const int64_t *pair_idxs, const double *uniform_samples, int64_t *out_replica_idx_by_state, uint32_t *proposed,
    uint32_t *accepted) {
    TM_TRY
    require(n_replicas > 0 && n_states > 0 && n_pairs >= 0 && n_attempts >= 0, "run_neighbor_swaps: negative size");
    std::vector<int64_t> perm(replica_idx_by_state, replica_idx_by_state + n_states);
    for (int64_t r : perm) {
        require(r >= 0 && r < n_replicas, "run_neighbor_swaps: replica index out of range");
    }
    for (int k = 0; k < n_pairs; k++) {
        proposed[k] = 0;
        accepted[k] = 0;
        require(neighbor_pairs[2 * k] >= 0 && neighbor_pairs[2 * k] < n_states && neighbor_pairs[2 * k + 1] >= 0 && neighbor_pairs[2 * k + 1] < n_states,
                "run_neighbor_swaps: state index out of range");
    }
    for (int t = 0; t < n_attempts; t++) {
        const int64_t k = pair_idxs[t];
        require(k >= 0 && k < n_pairs, "run_neighbor_swaps: pair index out of range");
        const int64_t s_a = neighbor_pairs[2 * k], s_b = neighbor_pairs[2 * k + 1];
        proposed[k] += 1;
        const int64_t r_a = perm[s_a], r_b = perm[s_b];
        const double before = log_q_kl[r_a * n_states + s_a] + log_q_kl[r_b * n_states + s_b];
        const double after = log_q_kl[r_a * n_states + s_b] + log_q_kl[r_b * n_states + s_a];
        const double diff = after - before; // NaN for (-inf) - (-inf): the comparison below is then false, as in jnp
        const double acceptance_probability = std::exp(diff < 0.0 ? diff : (diff >= 0.0 ? 0.0 : diff));
        if (uniform_samples[t] < acceptance_probability) {
            perm[s_a] = r_b;
            perm[s_b] = r_a;
            accepted[k] += 1;
        }
    }
    for (int s = 0; s < n_states; s++) {
        out_replica_idx_by_state[s] = perm[s];
    }
    TM_CATCH
}
int tm_debug_float_to_fixed(int precision, int kind, const double *in, int n, uint64_t *out) {
    TM_TRY
    require(precision == TM_F32 || precision == TM_F64, "invalid precision");
    debug_float_to_fixed(precision == TM_F64 ? 8 : 4, kind, n, in, reinterpret_cast<u64 *>(out));
    TM_CATCH
}
int tm_debug_float_to_fixed_energy(int precision, const double *in, int n, tm_int128 *out) {
    TM_TRY
    require(precision == TM_F32 || precision == TM_F64, "invalid precision");
    debug_float_to_fixed_energy(precision == TM_F64 ? 8 : 4, n, in, as_i128(out));
    TM_CATCH
}
int tm_debug_check_guards(int *violations) {
    TM_TRY
#ifdef TM_GUARD
    *violations = GuardRegistry::get().check();
#else
    *violations = -1; // not a guard build
#endif
    TM_CATCH
}

} // extern "C"
