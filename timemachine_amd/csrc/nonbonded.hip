// NonbondedAllPairs / NonbondedPairList / Neighborlist / HilbertSort host classes + kernel instantiations.
// One translation unit on purpose: the tile kernel and the pair-list kernel must inline the same nb_pair() under
// the same compiler flags (exact fixed-point cancellation of exclusions).
#include "engine.hpp"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include "kernels_nblist.hip.hpp"
#ifdef TM_ROWBLOCK // the row-block kernel is a test / A-B artefact: built into the variant library libtimemachine_amd_rowblock.so only
#include "kernels_nonbonded_rowblock.hip.hpp"
#endif
#include "profiler.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <numeric>
#include <set>

#ifdef TM_GUARD
// debug builds: name every launch of the nonbonded pipeline and wait for it, so that a faulting kernel is the last one named
#define TM_DEBUG_SYNC(name, stream)                                                                                    \
    do {                                                                                                               \
        if (std::getenv("TM_DEBUG_SYNC")) {                                                                            \
            fprintf(stderr, "[sync] %s\n", name);                                                                      \
            fflush(stderr);                                                                                            \
            HIP_CHECK(hipStreamSynchronize(stream));                                                                   \
        }                                                                                                              \
    } while (0)
#else
#define TM_DEBUG_SYNC(name, stream)                                                                                    \
    do {                                                                                                               \
    } while (0)
#endif

namespace tmamd {

// =============================================================================================================
// Electrostatic force-factor table (nb_es_table.hip.hpp)
// =============================================================================================================
// F(s) continued smoothly through d = 1.2 (no clamp: the kernels select 0 there), in long double
static long double es_force_factor_reference(const long double beta, const long double s) {
    const long double pi = 3.14159265358979323846264338327950288L;
    const long double d = sqrtl(s), inv = 1.0L / d;
    const long double x = d / 1.2L;
    const long double x2 = x * x, x4 = x2 * x2, q = x4 * x4;
    const long double c = cosl(0.5L * pi * q), sn = sinl(0.5L * pi * q);
    const long double S = c * c * c;
    const long double d2 = d * d, d4 = d2 * d2, d7 = d4 * d2 * d;
    const long double k8 = powl(1.0L / 1.2L, 8);
    const long double dS = -12.0L * pi * k8 * d7 * sn * c * c;
    const long double e = erfcl(beta * d);
    const long double de = -2.0L * beta / sqrtl(pi) * expl(-(beta * d) * (beta * d));
    const long double damp = e * S, dprime = e * dS + de * S;
    return inv * (dprime * inv - damp * inv * inv);
}
// G(s) = erfc(beta d) S(d) / d, continued the same way
static long double es_energy_factor_reference(const long double beta, const long double s) {
    const long double pi = 3.14159265358979323846264338327950288L;
    const long double d = sqrtl(s);
    const long double x = d / 1.2L;
    const long double x2 = x * x, x4 = x2 * x2, q = x4 * x4;
    const long double c = cosl(0.5L * pi * q);
    return erfcl(beta * d) * c * c * c / d;
}

static void es_table_fit(long double (*fn)(long double, long double), const double beta, double *out) {
    const int n = ES_TAB_COEFFS;
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int iv = 0; iv < ES_TAB_INTERVALS; iv++) {
        const int e = ES_TAB_EXP_LO + iv / ES_TAB_PER_OCTAVE, j = iv % ES_TAB_PER_OCTAVE;
        const long double lo = ldexpl(1.0L + static_cast<long double>(j) / ES_TAB_PER_OCTAVE, e);
        const long double hi = ldexpl(1.0L + static_cast<long double>(j + 1) / ES_TAB_PER_OCTAVE, e);
        // interpolation at the Chebyshev nodes of [0, 1] (near-minimax), solved for monomial coefficients in t
        long double A[ES_TAB_COEFFS][ES_TAB_COEFFS + 1];
        for (int k = 0; k < n; k++) {
            const long double t = 0.5L + 0.5L * cosl((2 * k + 1) * pi / (2 * n));
            long double pw = 1.0L;
            for (int c = 0; c < n; c++) {
                A[k][c] = pw;
                pw *= t;
            }
            A[k][n] = fn(beta, lo + (hi - lo) * t);
        }
        for (int c = 0; c < n; c++) { // Gauss-Jordan with partial pivoting (6 x 6, long double)
            int piv = c;
            for (int r = c + 1; r < n; r++) {
                if (fabsl(A[r][c]) > fabsl(A[piv][c])) {
                    piv = r;
                }
            }
            for (int k = 0; k <= n; k++) {
                std::swap(A[c][k], A[piv][k]);
            }
            for (int r = 0; r < n; r++) {
                if (r != c) {
                    const long double f = A[r][c] / A[c][c];
                    for (int k = c; k <= n; k++) {
                        A[r][k] -= f * A[c][k];
                    }
                }
            }
        }
        for (int c = 0; c < n; c++) {
            out[iv * n + c] = static_cast<double>(A[c][n] / A[c][c]);
        }
    }
}

void es_force_table_host(const double beta, double *out) { es_table_fit(es_force_factor_reference, beta, out); }
void es_energy_table_host(const double beta, double *out) { es_table_fit(es_energy_factor_reference, beta, out); }

const double *es_force_table_device(const double beta) {
    // one table per (device, beta), kept for the life of the process: potentials of one state share it.  Constructors may run
    // on several host threads at once (the bindings release the GIL around calls): the cache is guarded.
    static std::map<std::pair<int, unsigned long long>, double *> cache;
    static std::mutex cache_mutex;
    if (!(beta == beta)) {
        throw std::runtime_error("beta must not be NaN");
    }
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    unsigned long long beta_bits = 0;
    std::memcpy(&beta_bits, &beta, sizeof(beta_bits)); // keyed on the bit pattern: -0.0 / 0.0 and the like stay apart
    const auto key = std::make_pair(dev, beta_bits);
    std::lock_guard<std::mutex> lock(cache_mutex);
    auto it = cache.find(key);
    if (it != cache.end()) {
        return it->second;
    }
    std::vector<double> host(2 * ES_TAB_DOUBLES);
    es_force_table_host(beta, host.data());
    es_energy_table_host(beta, host.data() + ES_TAB_DOUBLES);
    double *d = nullptr;
    HIP_CHECK(hipMalloc(&d, host.size() * sizeof(double)));
    HIP_CHECK(hipMemcpy(d, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
    cache[key] = d;
    return d;
}

#ifndef TM_STEPS_PER_SORT
#define TM_STEPS_PER_SORT 100
#endif
static const int STEPS_PER_SORT = TM_STEPS_PER_SORT; // reference: cpp/src/nonbonded_all_pairs.cu:16 (100).  Re-measured here (f64 ns/day at 50 / 100 / 200 / 400): 2907 / 2932 / 2927 / 2875
static const int STEPS_PER_SORT_GROUP = 200; // reference: cpp/src/nonbonded_interaction_group.cu:17

// =============================================================================================================
// Hilbert sort
// =============================================================================================================
static const int HILBERT_GRID_DIM = 128; // reference: cpp/src/kernels/k_hilbert.cuh:6
static const int HILBERT_N_BITS = 8;

// Hilbert index of a 3-D integer point, `nbits` bits per axis (Butz's algorithm; bit-compatible with the
// hilbert_c2i(3, nbits, ...) the reference uses to fill its LUT, cpp/src/hilbert_sort.cu:18-31 -- pinned by
// tests/test_oracle.py against that C code compiled into oracle/_ref).
static unsigned int hilbert_index_3d(unsigned int c0, unsigned int c1, unsigned int c2, int nbits) {
    unsigned long long index = 0;
    unsigned int rot = 0, flip = 0, prev = 0;
    for (int b = nbits - 1; b >= 0; b--) {
        const unsigned int g = ((c0 >> b) & 1u) | (((c1 >> b) & 1u) << 1) | (((c2 >> b) & 1u) << 2);
        const unsigned int t = g ^ prev ^ flip;
        prev = g;
        const unsigned int digit = ((t >> rot) | (t << (3 - rot))) & 7u;
        index = (index << 3) | digit;
        flip = 1u << rot;
        const unsigned int low = digit & (0u - digit) & 3u;
        rot = (rot + 1 + (low == 0 ? 0 : (low == 1 ? 1 : 2))) % 3;
    }
    unsigned long long pattern = 0;
    for (int k = 0; k < nbits; k++) {
        pattern |= 1ull << (3 * k);
    }
    index ^= pattern >> 1;
    for (int d = 1; d < 3 * nbits; d *= 2) {
        index ^= index >> d;
    }
    return static_cast<unsigned int>(index);
}

const std::vector<unsigned int> &HilbertSort::lut() {
    // (a function-local static with an initialiser: the first call is thread-safe -- constructors of potentials on different
    // devices may run on different host threads at once, each under its own device's API lock)
    static const std::vector<unsigned int> table = []() {
        std::vector<unsigned int> t(HILBERT_GRID_DIM * HILBERT_GRID_DIM * HILBERT_GRID_DIM);
        for (int i = 0; i < HILBERT_GRID_DIM; i++)
            for (int j = 0; j < HILBERT_GRID_DIM; j++)
                for (int k = 0; k < HILBERT_GRID_DIM; k++)
                    t[(i * HILBERT_GRID_DIM + j) * HILBERT_GRID_DIM + k] = hilbert_index_3d(i, j, k, HILBERT_N_BITS);
        return t;
    }();
    return table;
}

// reference: k_coords_to_kv_gather (cpp/src/kernels/k_hilbert.cu:9-54).  f64 on purpose: imaging with floor in f32
// can land outside the home box for large coordinates.
__global__ void k_hilbert_keys(
    const int N, const unsigned int *__restrict__ atom_idxs, const double *__restrict__ coords, const double *__restrict__ box,
    const unsigned int *__restrict__ bin_to_idx, unsigned int *__restrict__ keys, unsigned int *__restrict__ vals) {
    const double bx = box[0], by = box[4], bz = box[8];
    const double inv_bx = 1 / bx, inv_by = 1 / by, inv_bz = 1 / bz;
    const double inv_bin_width = min(min(inv_bx, inv_by), inv_bz) * (HILBERT_GRID_DIM - 1.0);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < N; idx += gridDim.x * blockDim.x) {
        const unsigned int a = atom_idxs[idx];
        double x = coords[a * 3 + 0], y = coords[a * 3 + 1], z = coords[a * 3 + 2];
        x -= bx * floor(x * inv_bx);
        y -= by * floor(y * inv_by);
        z -= bz * floor(z * inv_bz);
        // clamped: non-finite coordinates (a diverged run, an unfinished copy) must produce a bin, not a wild address
        const double top = HILBERT_GRID_DIM - 1.0;
        const unsigned int ix = static_cast<unsigned int>(fmin(fmax(x * inv_bin_width, 0.0), top));
        const unsigned int iy = static_cast<unsigned int>(fmin(fmax(y * inv_bin_width, 0.0), top));
        const unsigned int iz = static_cast<unsigned int>(fmin(fmax(z * inv_bin_width, 0.0), top));
        keys[idx] = bin_to_idx[(ix * HILBERT_GRID_DIM + iy) * HILBERT_GRID_DIM + iz];
        vals[idx] = a;
    }
}

HilbertSort::HilbertSort(const int N)
    : N_(N), d_bin_to_idx_(HILBERT_GRID_DIM * HILBERT_GRID_DIM * HILBERT_GRID_DIM), d_keys_in_(N), d_keys_out_(N), d_vals_in_(N),
      d_sort_storage_(nullptr), sort_storage_bytes_(0) {
    d_bin_to_idx_.copy_from(lut().data());
    // stable LSD radix sort == the cub::DeviceRadixSort::SortPairs the reference calls (hilbert_sort.cu:69-80)
    HIP_CHECK(rocprim::radix_sort_pairs(
        nullptr, sort_storage_bytes_, d_keys_in_.data, d_keys_out_.data, d_vals_in_.data, d_keys_in_.data, static_cast<size_t>(N_), 0, 32));
#ifdef TM_SORT_SLACK
    HIP_CHECK(hipMalloc(&d_sort_storage_, 2 * sort_storage_bytes_ + (4 << 20)));
#else
    HIP_CHECK(hipMalloc(&d_sort_storage_, sort_storage_bytes_ > 0 ? sort_storage_bytes_ : 1));
#endif
}

HilbertSort::~HilbertSort() {
    if (d_sort_storage_)
        (void)hipFree(d_sort_storage_);
}

void HilbertSort::sort_device(
    const int N, const unsigned int *d_atom_idxs, const double *d_coords, const double *d_box, unsigned int *d_output_perm,
    hipStream_t stream) {
    if (N > N_) {
        throw std::runtime_error("number of idxs to sort must be less than or equal to N");
    }
    const int tpb = DEFAULT_TPB;
    k_hilbert_keys<<<ceil_divide(N, tpb), tpb, 0, stream>>>(
        N, d_atom_idxs, d_coords, d_box, d_bin_to_idx_.data, d_keys_in_.data, d_vals_in_.data);
    HIP_CHECK(hipGetLastError());
    size_t bytes = sort_storage_bytes_;
    HIP_CHECK(rocprim::radix_sort_pairs(
        d_sort_storage_, bytes, d_keys_in_.data, d_keys_out_.data, d_vals_in_.data, d_output_perm, static_cast<size_t>(N), 0, 32, stream));
}

std::vector<unsigned int> HilbertSort::sort_host(const int N, const double *h_coords, const double *h_box) {
    std::vector<unsigned int> h_idxs(N);
    std::iota(h_idxs.begin(), h_idxs.end(), 0);
    DeviceBuffer<double> d_coords(N * 3), d_box(9);
    DeviceBuffer<unsigned int> d_idxs(N), d_perm(N);
    d_coords.copy_from(h_coords);
    d_box.copy_from(h_box);
    d_idxs.copy_from(h_idxs.data());
    hipStream_t stream = 0;
    sort_device(N, d_idxs.data, d_coords.data, d_box.data, d_perm.data, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    d_perm.copy_to(h_idxs.data());
    return h_idxs;
}

// =============================================================================================================
// Neighborlist
// =============================================================================================================
__global__ void k_arange(const int n, unsigned int *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = i;
}

// host-API helper: pack (x, y, z) into the Real[K][8] record layout the build kernels read
template <typename Real> __global__ void k_pack_coords(const int n, const double *__restrict__ x, Real *__restrict__ gathered) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        for (int d = 0; d < 3; d++)
            gathered[static_cast<size_t>(i) * 8 + d] = static_cast<Real>(x[i * 3 + d]);
        for (int d = 3; d < 8; d++)
            gathered[static_cast<size_t>(i) * 8 + d] = 0;
    }
}

template <typename Real>
Neighborlist<Real>::Neighborlist(const int N) : max_size_(N), N_(N), NC_(N), NR_(N) {
    if (N == 0) {
        throw std::runtime_error("Neighborlist N must be at least 1");
    }
    const int nb = ceil_divide(N, TILE);
    d_col_ctr_.realloc(nb * 3);
    d_col_ext_.realloc(nb * 3);
    d_row_ctr_.realloc(nb * 3);
    d_row_ext_.realloc(nb * 3);
    d_row_idxs_.realloc(N);
    d_col_idxs_.realloc(N);
    d_counters_.realloc(NB_COUNTERS_ALLOC); // (the words behind NB_NUM_COUNTERS stay zero for good: engine.hpp)
    HIP_CHECK(hipMemset(d_counters_.data, 0, NB_COUNTERS_ALLOC * sizeof(unsigned int)));
    // Pool sized for the worst case: every block pair interacting.  Row subsets need rows x cols <= (nb/2)^2 block
    // pairs, always below the upper-triangular count used by the reference (neighborlist.cu:368-376).
    const size_t max_block_pairs = static_cast<size_t>(nb) * (nb + 1) / 2;
    d_col_atoms_.realloc(max_block_pairs * TILE);
    // one bucket per (shard = row block % 8, cost class); a shard's row blocks each yield at most ceil(N / 64) items
    items_cap_ = std::min(static_cast<size_t>(nb / NB_SHARDS + 1) * (ceil_divide(N, NB_CHUNK) + 1),
                          max_block_pairs * TILE / NB_CHUNK + nb + 1);
    d_items_.realloc(static_cast<size_t>(NB_SHARDS) * NB_CLASSES * items_cap_);
    d_row_segments_.realloc(nb);
    HIP_CHECK(hipMemset(d_row_segments_.data, 0, nb * sizeof(int2)));
    this->reset_row_idxs();
}

template <typename Real> int Neighborlist<Real>::max_ixn_count() const {
    const int nb = ceil_divide(max_size_, TILE);
    return (nb * (nb + 1)) / 2 * TILE;
}

template <typename Real> void Neighborlist<Real>::reset_row_idxs() {
    const int tpb = DEFAULT_TPB;
    k_arange<<<ceil_divide(N_, tpb), tpb, 0, 0>>>(N_, d_col_idxs_.data);
    k_arange<<<ceil_divide(N_, tpb), tpb, 0, 0>>>(N_, d_row_idxs_.data);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(0));
    NR_ = N_;
    NC_ = N_;
}

template <typename Real> void Neighborlist<Real>::resize(const int size) {
    if (size <= 0) {
        throw std::runtime_error("size is must be at least 1");
    }
    if (size > max_size_) {
        throw std::runtime_error("size is greater than max size: " + std::to_string(size) + " > " + std::to_string(max_size_));
    }
    N_ = size;
    this->reset_row_idxs();
}

template <typename Real> void Neighborlist<Real>::set_row_idxs(std::vector<unsigned int> row_idxs) {
    if (row_idxs.size() == 0) {
        throw std::runtime_error("idxs can't be empty");
    }
    std::set<unsigned int> unique_idxs(row_idxs.begin(), row_idxs.end());
    if (unique_idxs.size() != row_idxs.size()) {
        throw std::runtime_error("atom indices must be unique");
    }
    if (row_idxs.size() >= static_cast<size_t>(N_)) {
        throw std::runtime_error("number of idxs must be less than N");
    }
    if (*std::max_element(row_idxs.begin(), row_idxs.end()) >= static_cast<unsigned int>(N_)) {
        throw std::runtime_error("indices values must be less than N");
    }
    std::vector<unsigned int> col_idxs;
    col_idxs.reserve(N_ - row_idxs.size());
    for (unsigned int i = 0; i < static_cast<unsigned int>(N_); i++) {
        if (!unique_idxs.count(i))
            col_idxs.push_back(i);
    }
    DeviceBuffer<unsigned int> d_rows(row_idxs.size()), d_cols(col_idxs.size());
    d_rows.copy_from(row_idxs.data());
    d_cols.copy_from(col_idxs.data());
    this->set_idxs_device(col_idxs.size(), row_idxs.size(), d_cols.data, d_rows.data, 0);
    HIP_CHECK(hipStreamSynchronize(0));
}

template <typename Real>
void Neighborlist<Real>::set_idxs_device(
    const int NC, const int NR, const unsigned int *d_in_col, const unsigned int *d_in_row, hipStream_t stream) {
    if (NC + NR != N_) {
        throw std::runtime_error("Total of indices must equal N");
    }
    if (NC == 0 || NR == 0) {
        throw std::runtime_error("Number of column and row indices must be non-zero");
    }
    HIP_CHECK(hipMemcpyAsync(d_col_idxs_.data, d_in_col, NC * sizeof(unsigned int), hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(d_row_idxs_.data, d_in_row, NR * sizeof(unsigned int), hipMemcpyDeviceToDevice, stream));
    NR_ = NR;
    NC_ = NC;
}

template <typename Real> unsigned int Neighborlist<Real>::num_tile_ixns() {
    unsigned int h[4];
    HIP_CHECK(hipMemcpy(h, d_counters_.data, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost));
    return h[2];
}

template <typename Real> unsigned int Neighborlist<Real>::num_builds() {
    unsigned int h[4];
    HIP_CHECK(hipMemcpy(h, d_counters_.data, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost));
    return h[3];
}

template <typename Real>
void Neighborlist<Real>::build_device(
    const Real *d_gathered, const double *d_box, const double cutoff, const double cost_cutoff, const int *d_flag,
    const int force, const int n_snap, const double *d_x, double *d_snap_x, double *d_snap_box, hipStream_t stream,
    const bool bounds_done, const bool rebase_box) {
    const bool ut = this->upper_triangular();
    const int ncb = this->num_column_blocks();
    const int nrb = this->num_row_blocks();
    const int total_blocks = ncb + (ut ? 0 : nrb);
    const int tpb = DEFAULT_TPB;
    // one wave per block (4 per workgroup); the same threads also copy the coordinate snapshot grid-stride
    int grid = std::min(std::max(ceil_divide(total_blocks, tpb / 64), 1), 2048);
    const int dummy_flag_force = d_flag ? force : 1;
    if (bounds_done && (!ut || force || d_snap_x == nullptr)) {
        throw std::runtime_error("Neighborlist::build_device: bounds_done needs the upper-triangular, flag-driven MD path");
    }
    if (!bounds_done) {
        k_block_bounds<Real, false><<<grid, tpb, 0, stream>>>(
            ncb, NC_, ut ? nullptr : d_col_idxs_.data, nrb, NR_, ut ? nullptr : d_row_idxs_.data, ut ? 1 : 0, d_gathered, d_box,
            d_col_ctr_.data, d_col_ext_.data, d_row_ctr_.data, d_row_ext_.data, d_counters_.data, n_snap, d_x, d_snap_x, d_snap_box,
            d_flag ? d_flag : reinterpret_cast<const int *>(d_counters_.data), dummy_flag_force, rebase_box ? d_snap_box : nullptr,
            ut ? guest_rows_ : 0, ut ? guest_blocks_ : 0);
        HIP_CHECK(hipGetLastError());
        TM_DEBUG_SYNC("k_block_bounds", stream);
    }
    // the list kernel's own snapshot duty (only when no bounds kernel ran)
    const int ls_n = bounds_done ? n_snap : 0;
    const double *ls_x = bounds_done ? d_x : nullptr;
    double *ls_snap_x = bounds_done ? d_snap_x : nullptr, *ls_snap_box = bounds_done ? d_snap_box : nullptr;
    const size_t lds = 0;
    if (ut) {
        k_find_ixns<Real, true><<<nrb, NBL_THREADS, lds, stream>>>(
            N_, NC_, NR_, nullptr, nullptr, d_col_ctr_.data, d_col_ext_.data, d_col_ctr_.data, d_col_ext_.data, d_gathered, d_box,
            cutoff, cost_cutoff, d_counters_.data, d_col_atoms_.data, d_items_.data, static_cast<unsigned int>(items_cap_), d_row_segments_.data,
            d_flag ? d_flag : reinterpret_cast<const int *>(d_counters_.data), dummy_flag_force, ls_n, ls_x, ls_snap_x, ls_snap_box,
            guest_rows_, guest_blocks_, guest_blocks_ > 0 ? d_guest_items_.data : nullptr);
    } else {
        k_find_ixns<Real, false><<<nrb, NBL_THREADS, lds, stream>>>(
            N_, NC_, NR_, d_col_idxs_.data, d_row_idxs_.data, d_col_ctr_.data, d_col_ext_.data, d_row_ctr_.data, d_row_ext_.data,
            d_gathered, d_box, cutoff, cost_cutoff, d_counters_.data, d_col_atoms_.data, d_items_.data, static_cast<unsigned int>(items_cap_), d_row_segments_.data,
            d_flag ? d_flag : reinterpret_cast<const int *>(d_counters_.data), dummy_flag_force, ls_n, ls_x, ls_snap_x, ls_snap_box);
    }
    HIP_CHECK(hipGetLastError());
}

template <typename Real>
void Neighborlist<Real>::gather_host_coords(const int N, const double *h_coords, const double *h_box, DeviceBuffer<double> &d_box) {
    DeviceBuffer<double> d_coords(N * 3);
    d_coords.copy_from(h_coords);
    d_box.copy_from(h_box);
    d_scratch_gathered_.reserve(static_cast<size_t>(N) * 8);
    k_pack_coords<Real><<<ceil_divide(N, DEFAULT_TPB), DEFAULT_TPB, 0, 0>>>(N, d_coords.data, d_scratch_gathered_.data);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(0));
}

template <typename Real>
std::vector<std::vector<int>>
Neighborlist<Real>::get_nblist_host(const int N, const double *h_coords, const double *h_box, const double cutoff) {
    if (N != N_) {
        throw std::runtime_error("N != N_");
    }
    DeviceBuffer<double> d_box(9);
    this->gather_host_coords(N, h_coords, h_box, d_box);
    this->build_device(d_scratch_gathered_.data, d_box.data, cutoff, cutoff, nullptr, 1, 0, nullptr, nullptr, nullptr, 0);
    HIP_CHECK(hipStreamSynchronize(0));
    const int nrb = this->num_row_blocks();
    std::vector<int2> segs(nrb);
    HIP_CHECK(hipMemcpy(segs.data(), d_row_segments_.data, nrb * sizeof(int2), hipMemcpyDeviceToHost));
    std::vector<std::vector<int>> out(nrb);
    std::vector<unsigned int> seg;
    for (int r = 0; r < nrb; r++) {
        seg.resize(segs[r].y);
        if (segs[r].y > 0) {
            HIP_CHECK(hipMemcpy(seg.data(), d_col_atoms_.data + segs[r].x, segs[r].y * sizeof(unsigned int), hipMemcpyDeviceToHost));
        }
        out[r].assign(seg.begin(), seg.end());
        std::sort(out[r].begin(), out[r].end());
    }
    return out;
}

template <typename Real>
void Neighborlist<Real>::compute_block_bounds_host(
    const int N, const double *h_coords, const double *h_box, double *h_bb_ctrs, double *h_bb_exts) {
    DeviceBuffer<double> d_box(9);
    this->gather_host_coords(N, h_coords, h_box, d_box);
    const bool ut = this->upper_triangular();
    const int ncb = this->num_column_blocks();
    const int nrb = this->num_row_blocks();
    const int total_blocks = ncb + (ut ? 0 : nrb);
    k_block_bounds<Real, true><<<std::max(ceil_divide(total_blocks, DEFAULT_TPB / 64), 1), DEFAULT_TPB, 0, 0>>>(
        ncb, NC_, ut ? nullptr : d_col_idxs_.data, nrb, NR_, ut ? nullptr : d_row_idxs_.data, ut ? 1 : 0, d_scratch_gathered_.data,
        d_box.data, d_col_ctr_.data, d_col_ext_.data, d_row_ctr_.data, d_row_ext_.data, d_counters_.data, 0, nullptr, nullptr, nullptr,
        reinterpret_cast<const int *>(d_counters_.data), 1, nullptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(0));
    std::vector<Real> ctr(ncb * 3), ext(ncb * 3);
    d_col_ctr_.copy_to(ctr.data(), ncb * 3);
    d_col_ext_.copy_to(ext.data(), ncb * 3);
    for (int i = 0; i < ncb * 3; i++) {
        h_bb_ctrs[i] = ctr[i];
        h_bb_exts[i] = ext[i];
    }
}

template class Neighborlist<float>;
template class Neighborlist<double>;

// =============================================================================================================
// NonbondedAllPairs
// =============================================================================================================
bool g_box_scaling_reuse = true;

void verify_atom_idxs(const int N, const std::vector<int> &atom_idxs, const bool allow_empty) {
    // reference: cpp/src/nonbonded_common.cpp:39-60 (messages are part of the contract)
    if (atom_idxs.size() == 0) {
        if (allow_empty)
            return;
        throw std::runtime_error("indices can't be empty");
    }
    std::set<int> unique_idxs(atom_idxs.begin(), atom_idxs.end());
    if (unique_idxs.size() != atom_idxs.size()) {
        throw std::runtime_error("atom indices must be unique");
    }
    if (*std::max_element(atom_idxs.begin(), atom_idxs.end()) >= N) {
        throw std::runtime_error("index values must be less than N(" + std::to_string(N) + ")");
    }
    if (*std::min_element(atom_idxs.begin(), atom_idxs.end()) < 0) {
        throw std::runtime_error("index values must be greater or equal to zero");
    }
}

void nb_du_dp_fixed_to_float(const int N, const u64 *du_dp, double *out) {
    // per-column exponents, reference: cpp/src/nonbonded_all_pairs.cu:292-308
    for (int i = 0; i < N; i++) {
        out[i * 4 + 0] = static_cast<double>(static_cast<long long>(du_dp[i * 4 + 0])) / TM_FIXED_EXPONENT_DU_DCHARGE;
        out[i * 4 + 1] = static_cast<double>(static_cast<long long>(du_dp[i * 4 + 1])) / TM_FIXED_EXPONENT_DU_DSIG;
        out[i * 4 + 2] = static_cast<double>(static_cast<long long>(du_dp[i * 4 + 2])) / TM_FIXED_EXPONENT_DU_DEPS;
        out[i * 4 + 3] = static_cast<double>(static_cast<long long>(du_dp[i * 4 + 3])) / TM_FIXED_EXPONENT_DU_DW;
    }
}

template <typename Real>
NonbondedAllPairs<Real>::NonbondedAllPairs(
    const int N, const double beta, const double cutoff, const std::optional<std::vector<int>> &atom_idxs,
    const bool disable_hilbert_sort, const double nblist_padding)
    : steps_per_sort_(STEPS_PER_SORT), N_(N), K_(N), beta_(beta), cutoff_(cutoff), nblist_padding_(nblist_padding),
      disable_hilbert_(disable_hilbert_sort), calls_since_sort_(0), parity_(0), force_rebuild_(true), nblist_(N) {
    std::vector<int> idxs;
    if (atom_idxs) {
        idxs = *atom_idxs;
        std::sort(idxs.begin(), idxs.end());
        idxs.erase(std::unique(idxs.begin(), idxs.end()), idxs.end());
    } else {
        idxs.resize(N_);
        std::iota(idxs.begin(), idxs.end(), 0);
    }
    verify_atom_idxs(N_, idxs);
    this->allocate();
    this->set_atom_idxs(idxs);
}

template <typename Real>
NonbondedAllPairs<Real>::NonbondedAllPairs(
    const int N, const double beta, const double cutoff, const bool disable_hilbert_sort, const double nblist_padding, GroupTag)
    : name_("NonbondedInteractionGroup"), steps_per_sort_(STEPS_PER_SORT_GROUP), N_(N), K_(N), beta_(beta), cutoff_(cutoff),
      nblist_padding_(nblist_padding), disable_hilbert_(disable_hilbert_sort), calls_since_sort_(0), parity_(0), force_rebuild_(true),
      nblist_(N) {
    this->allocate();
}

template <typename Real>
NonbondedAllPairs<Real>::NonbondedAllPairs(
    const int N, const double beta, const double cutoff, const bool disable_hilbert_sort, const double nblist_padding, MergedTag,
    const std::vector<unsigned int> &host_idxs, const std::vector<unsigned int> &guest_idxs)
    : merged_mode_(true), steps_per_sort_(STEPS_PER_SORT), N_(N), K_(N), beta_(beta), cutoff_(cutoff), nblist_padding_(nblist_padding),
      disable_hilbert_(disable_hilbert_sort), calls_since_sort_(0), parity_(0), force_rebuild_(true), nblist_(N + TILE) {
    this->allocate();
    const int L = static_cast<int>(guest_idxs.size()), K1 = static_cast<int>(host_idxs.size());
    guest_rows_ = L;
    guest_pad_ = ceil_divide(L, TILE) * TILE;
    std::vector<unsigned int> slots(guest_idxs);
    slots.resize(guest_pad_, NB_HOLE);
    slots.insert(slots.end(), host_idxs.begin(), host_idxs.end());
    K_ = guest_pad_ + K1;
    n_atoms_ = L + K1;
    h_atom_idxs_ = slots;
    d_atom_idxs_.copy_from(slots.data(), K_);
    nblist_.resize(K_);
    nblist_.set_guest(guest_rows_, guest_pad_ / TILE);
}

template <typename Real> void NonbondedAllPairs<Real>::allocate() {
    if (sizeof(Real) == 8) {
        d_es_table_ = es_force_table_device(beta_);
    }
    const int cap = slot_capacity(); // (a merged order pads the guest rows to a block boundary: up to TILE - 1 holes)
    d_atom_idxs_.realloc(cap);
    d_perm_.realloc(cap);
    d_gathered_.realloc(static_cast<size_t>(cap + 1) * 8 * (merged_mode_ ? 2 : 1)); // + one all-zero sentinel record (merged: two sets)
    acc_stride_ = (cap + 7) & ~7; // every component array starts on a 64-byte line
    d_g_du_dx_.realloc(static_cast<size_t>(acc_stride_) * 3);
    d_g_du_dp_.realloc(static_cast<size_t>(acc_stride_) * 4);
    d_snap_x_.realloc(static_cast<size_t>(N_) * 3);
    d_snap_box_.realloc(12); // [0..8] the box the snapshot is expressed in; [9..11] accumulated scale since the last build
    HIP_CHECK(hipMemset(d_snap_x_.data, 0, d_snap_x_.size()));
    HIP_CHECK(hipMemset(d_snap_box_.data, 0, d_snap_box_.size()));
    d_flags_.realloc(2);
    HIP_CHECK(hipMemset(d_flags_.data, 0, d_flags_.size()));
    d_slot_of_atom_.realloc(N_);
    HIP_CHECK(hipMemset(d_slot_of_atom_.data, 0xff, d_slot_of_atom_.size())); // -1: no atom is ours until K1 says so
    // persistent grid: TileWaves<Real> waves per SIMD on every CU, grouped into one or two workgroups per CU
    // (TileShape); grid_ = waves of the forces-only launch = the most any variant launches
    grid_ = device_cu_count() * 4 * TileWaves<Real>::value;
    d_timing_.realloc(static_cast<size_t>(grid_) * 8);
    HIP_CHECK(hipMemset(d_timing_.data, 0, d_timing_.size()));
    d_u_partials_.realloc(grid_);
    if (!disable_hilbert_) {
        hilbert_.reset(new HilbertSort(N_));
    }
}

// Measured on water boxes and solvated ligands (scripts/size_sweep.py, us per MD step, listed -> static; f64 / f32):
//   256 atoms 19.3 -> 16.4 / 18.3 -> 15.6;  900: 22.0 -> 17.9 / 20.2 -> 16.9;  2 243: 29.8 -> 24.2 / 27.5 -> 22.7;
//   3 600: 31.4 -> 26.1 / 33.8 -> 22.3;  6 318: 40.3 -> 38.7 / 34.6 -> 39.3;  9 000: 44.6 -> 58.9 / 40.9 -> 52.6
#ifndef TM_STATIC_LIST_MAX_K
#define TM_STATIC_LIST_MAX_K 4608
#endif
int g_static_list_max_k = std::getenv("TM_AMD_STATIC_LIST_MAX_K") ? std::atoi(std::getenv("TM_AMD_STATIC_LIST_MAX_K")) : TM_STATIC_LIST_MAX_K;
template <typename Real> int NonbondedAllPairs<Real>::static_list_max_k() { return g_static_list_max_k; }
// Forces-only launches over at least this many atoms run the row-block kernel (kernels_nonbonded_rowblock.hip.hpp: one workgroup
// per (row block, <= 1024 listed columns), lane-owned columns regrouped by hit count) -- IN LIBRARIES BUILT WITH -DTM_ROWBLOCK
// (csrc/build.py builds libtimemachine_amd_rowblock.so; the product library does not carry the kernel: round 4 built it as the
// re-decomposition of the tile kernel, it passes the whole GPU suite bit for bit and measures 74 us per launch against the item
// kernel's 55 on the DHFR-shaped box, EXPERIMENTS.md "row-block kernel").  Its value is that of a second, independent
// implementation the parity tests compare bit for bit (tests/test_gpu_second_binding.py runs them against the variant library).
#ifndef TM_ROWBLOCK_MIN_K
#define TM_ROWBLOCK_MIN_K 2147483647
#endif
#ifdef TM_ROWBLOCK
const bool g_rowblock_built = true;
int g_rowblock_min_k = std::getenv("TM_AMD_ROWBLOCK_MIN_K") ? std::atoi(std::getenv("TM_AMD_ROWBLOCK_MIN_K")) : TM_ROWBLOCK_MIN_K;
#else
const bool g_rowblock_built = false;
int g_rowblock_min_k = TM_ROWBLOCK_MIN_K;
#endif

template <typename Real> void NonbondedAllPairs<Real>::set_atom_idxs(const std::vector<int> &atom_idxs) {
    verify_atom_idxs(N_, atom_idxs);
    std::vector<unsigned int> u(atom_idxs.begin(), atom_idxs.end());
    const int K = static_cast<int>(u.size());
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemset(d_slot_of_atom_.data, 0xff, d_slot_of_atom_.size())); // atoms that leave the set must read -1
    d_atom_idxs_.copy_from(u.data(), K);
    nblist_.resize(K);
    K_ = K;
    n_atoms_ = K;
    h_atom_idxs_ = u;
    std::sort(h_atom_idxs_.begin(), h_atom_idxs_.end());
    idxs_version_++;
    calls_since_sort_ = 0; // next call sorts (and therefore rebuilds)
    force_rebuild_ = true;
    last_x_ = last_box_ = nullptr;
}

template <typename Real>
bool NonbondedAllPairs<Real>::piggyback_forces(
    const FusedTable *d_table, const int blocks, const int precision_bytes, u64 *acc, const int atom_stride, const int comp_stride) {
    if (precision_bytes != static_cast<int>(sizeof(Real)) || piggyback_table_ != nullptr || empty_) {
        return false;
    }
    piggyback_table_ = d_table;
    piggyback_blocks_ = blocks;
    piggyback_acc_ = acc;
    piggyback_atom_stride_ = atom_stride;
    piggyback_comp_stride_ = comp_stride;
    // A potential that covers every atom takes the table's forces into its own Hilbert-ordered accumulator: whoever consumes
    // that accumulator (the un-permute pass, or the integrator through a deferred hand-over) then finds bonded and
    // nonbonded forces of an atom in one place, and the caller's accumulator is not touched.
    piggyback_redirect_ = covers_all();
    return true;
}

template <typename Real> bool NonbondedAllPairs<Real>::piggyback_lands_in_own_accumulator() const { return piggyback_table_ != nullptr && piggyback_redirect_; }

template <typename Real> bool NonbondedAllPairs<Real>::piggyback_energy(const FusedTable *d_table, const int blocks, const int precision_bytes) {
    if (precision_bytes != static_cast<int>(sizeof(Real)) || piggyback_energy_table_ != nullptr || empty_) {
        return false;
    }
    piggyback_energy_table_ = d_table;
    piggyback_energy_blocks_ = blocks;
    return true;
}

// process-wide A/B switch (tm_debug_set_same_frame_hint): Potential::hint_same_frame is honoured
bool g_same_frame_hint = true;
thread_local long long g_eval_serial = 0;
long long next_eval_serial() {
    static std::atomic<long long> counter{0};
    return ++counter;
}

// process-wide A/B switch (tm_debug_set_energy_memo, TM_AMD_NO_ENERGY_MEMO): energy-only evaluations are remembered on the device
bool g_energy_memo = std::getenv("TM_AMD_NO_ENERGY_MEMO") == nullptr;

template <typename Real> void NonbondedAllPairs<Real>::memo_stats(long long *evaluations, long long *skipped) {
    *evaluations = 0;
    *skipped = 0;
    NonbondedAllPairs<Real> *who[2] = {this, merged_.get()};
    for (NonbondedAllPairs<Real> *p : who) {
        if (p != nullptr && p->d_memo_.data != nullptr) {
            EnergyMemo m;
            HIP_CHECK(hipDeviceSynchronize());
            p->d_memo_.copy_to(&m, 1);
            *evaluations += m.evaluations;
            *skipped += m.skipped;
        }
    }
}

// process-wide A/B switch (tm_debug_set_merge_producers, TM_AMD_NO_MERGE): planned all-pairs + interaction-group pairs run as one pipeline
bool g_merge_producers = std::getenv("TM_AMD_NO_MERGE") == nullptr;

template <typename Real>
Potential *NonbondedAllPairs<Real>::merged_carrier(NonbondedAllPairsBase *group, const int P_group, const double *d_p_group) {
    if (!g_merge_producers || merged_mode_ || group_rows_ != 0 || empty_ || group == nullptr || group == this || d_p_group == nullptr) {
        return nullptr;
    }
    if (merged_group_ != group || merged_versions_[0] != idxs_version_ || merged_versions_[1] != group->idxs_version()) {
        // a new pairing (or an atom set changed: set_atom_idxs, local MD narrowing this potential): decide again, once
        merged_.reset();
        merged_group_ = group;
        merged_versions_[0] = idxs_version_;
        merged_versions_[1] = group->idxs_version();
        merged_refused_ = true;
        const std::vector<unsigned int> &g = group->host_atom_idxs();
        const int L = group->num_group_rows();
        const bool fits = group->is_interaction_group() && !group->is_empty_group() && group->precision_bytes() == static_cast<int>(sizeof(Real)) &&
                          group->get_beta() == beta_ && group->get_cutoff() == cutoff_ && group->get_nblist_padding() == nblist_padding_ &&
                          group->hilbert_disabled() == disable_hilbert_ && L > 0 && static_cast<int>(g.size()) - L == K_;
        if (fits) {
            // the group's columns must be exactly this potential's atoms (its rows are disjoint from its columns by construction)
            std::vector<unsigned int> cols(g.begin() + L, g.end());
            std::sort(cols.begin(), cols.end());
            if (cols == h_atom_idxs_) {
                merged_.reset(new NonbondedAllPairs<Real>(
                    N_, beta_, cutoff_, disable_hilbert_, nblist_padding_, MergedTag{}, h_atom_idxs_, std::vector<unsigned int>(g.begin(), g.begin() + L)));
                merged_refused_ = false;
                merged_group_epoch_ = group->inputs_epoch();
            }
        }
    }
    if (merged_refused_ || P_group != N_ * PARAMS_PER_ATOM) {
        return nullptr;
    }
    if (merged_group_epoch_ != group->inputs_epoch()) { // the group's inputs changed behind an unchanged pointer (set_params)
        merged_->invalidate_cached_inputs();
        merged_group_epoch_ = group->inputs_epoch();
    }
    merged_->guest_p_ = d_p_group;
    if (same_frame_hint_) { // (the plan evaluates the carrier in this potential's place)
        merged_->hint_same_frame(true, hint_call_);
        same_frame_hint_ = false;
    }
    merged_->box_scales_ = box_scales_ || group->expects_box_scaling();
    this->carrier_took_over();
    group->carrier_took_over();
    return merged_.get();
}

template <typename Real> std::vector<long long> NonbondedAllPairs<Real>::debug_timing() {
    std::vector<long long> raw(static_cast<size_t>(grid_) * 8);
    HIP_CHECK(hipDeviceSynchronize());
    d_timing_.copy_to(raw.data());
    return raw;
}

template <typename Real> std::vector<int> NonbondedAllPairs<Real>::get_atom_idxs() {
    std::vector<unsigned int> u(K_);
    d_atom_idxs_.copy_to(u.data(), K_);
    return std::vector<int>(u.begin(), u.end());
}

// the static-list limit (a process-wide debug knob) or the atom set changed: switch modes with a fresh list, before any caller
// decides whether pre-gathered state may be used
template <typename Real> void NonbondedAllPairs<Real>::sync_list_mode() {
    if (wants_static_list() != static_mode_) {
        static_mode_ = wants_static_list();
        static_list_built_ = false;
        force_rebuild_ = true;
    }
}

template <typename Real>
bool NonbondedAllPairs<Real>::execute_forces_deferred(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, hipStream_t stream,
    DeferredForces &out) {
    same_frame_now_ = false; // (MD steps move the atoms)
    same_frame_hint_ = false;
    this->check_sizes(N, P);
    if (empty_) {
        return false; // nothing to hand over; the caller falls back to execute_device (a no-op)
    }
    sync_list_mode();
    // positions pre-gathered by the consumer of the previous deferred call are usable iff they were made from exactly
    // these inputs and the order they were written in still stands (no re-sort, no forced rebuild on this call)
    const bool pregathered = pre_valid_ && d_x == pre_x_ && d_p == pre_p_ && guest_p_ == pre_guest_p_ && d_box == pre_box_ && !force_rebuild_ &&
                             calls_since_sort_ % steps_per_sort_ != 0;
    out.consumed_sorted_pregather = pregathered && pre_sorted_;
    this->run_pipeline(d_x, d_p, d_box, d_du_dx, nullptr, nullptr, false, stream, pregathered);
    out.g_du_dx = d_g_du_dx_.data;
    out.stride = acc_stride_;
    out.slot_of_atom = d_slot_of_atom_.data;
    out.owner = this;
    out.next.gathered = d_gathered_.data;
    out.next.real_bytes = static_cast<int>(sizeof(Real));
    out.next.snap_x = d_snap_x_.data;
    out.next.pad2_quarter = rebuild_threshold2();
    out.next.snap_box = scale_aware() && !static_list() ? d_snap_box_.data : nullptr; // (an accepted barostat move may leave the box ahead of the snapshot's)
    out.next.cur_box = d_box;
    out.next.flag_set = d_flags_.data + parity_; // run_pipeline has already advanced parity_: the NEXT call's pair
    out.next.flag_clear = d_flags_.data + (parity_ ^ 1);
    out.next.g_du_dx = d_g_du_dx_.data;
    out.next.stride = acc_stride_;
    out.next.second_records = merged_mode_ ? K_ + 1 : 0;
    if (covers_all() && nblist_.upper_triangular()) { // plain (or merged) all-pairs over every atom: see PregatherTarget
        out.next.perm = d_perm_.data;
        out.next.sorted_n = K_;
        out.next.covers_atoms = n_atoms_;
        out.next.blk_ctr = nblist_.d_col_ctr();
        out.next.blk_ext = nblist_.d_col_ext();
        out.next.nbl_counters = nblist_.d_counters_rw();
    }
    offer_p_ = d_p;
    offer_guest_p_ = guest_p_;
    return true;
}

template <typename Real> void NonbondedAllPairs<Real>::pregather_committed(const double *d_x, const double *d_box, const bool sorted_bounds_done) {
    pre_valid_ = true;
    memo_chain_ = false; // (the consumer rewrote records: the device's energy memo no longer describes them)
    last_x_ = last_box_ = nullptr;
    pre_sorted_ = sorted_bounds_done;
    pre_x_ = d_x;
    pre_box_ = d_box;
    pre_p_ = offer_p_;
    pre_guest_p_ = offer_guest_p_;
}

template <typename Real>
void NonbondedAllPairs<Real>::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    same_frame_now_ = same_frame_hint_; // (taken before anything can throw: a hint never outlives the call it was given for)
    hint_call_now_ = hint_call_;
    same_frame_hint_ = false;
    this->check_sizes(N, P);
    if (empty_) {
        return; // reference: nonbonded_interaction_group.cu:171-174 (outputs untouched, d_u left as the caller set it)
    }
    sync_list_mode();
    // An energy-only call on exactly the inputs the last MD step left pre-gathered (the barostat's "before" energy,
    // a frame's energy right after multiple_steps) reads the sorted records as they are: no gather, no bounds kernel, and
    // above all no forced list rebuild -- the update kernel has already made the displacement test for these coordinates.
    // (Accumulators are not touched by an energy-only launch; the hand-over is consumed: the next call gathers itself.)
    const bool pregathered = d_du_dx == nullptr && d_du_dp == nullptr && d_u != nullptr && pre_valid_ && d_x == pre_x_ &&
                             d_p == pre_p_ && guest_p_ == pre_guest_p_ && d_box == pre_box_ && !force_rebuild_ && calls_since_sort_ % steps_per_sort_ != 0;
    this->run_pipeline(d_x, d_p, d_box, d_du_dx, d_du_dp, d_u, true, stream, pregathered);
}

template <typename Real>
bool NonbondedAllPairs<Real>::execute_energy_partials(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, hipStream_t stream, const i128 *&partials,
    int &count, i128 *d_final) {
    same_frame_now_ = same_frame_hint_;
    hint_call_now_ = hint_call_;
    same_frame_hint_ = false;
    this->check_sizes(N, P);
    if (empty_) {
        partials = d_u_partials_.data; // an interaction group without interactions: nothing to add
        count = 0;
        return true;
    }
    sync_list_mode();
    const bool pregathered = pre_valid_ && d_x == pre_x_ && d_p == pre_p_ && guest_p_ == pre_guest_p_ && d_box == pre_box_ && !force_rebuild_ &&
                             calls_since_sort_ % steps_per_sort_ != 0;
    defer_u_reduce_ = true;
    memo_final_ = d_final;
    try {
        this->run_pipeline(d_x, d_p, d_box, nullptr, nullptr, d_u_partials_.data, true, stream, pregathered);
    } catch (...) {
        defer_u_reduce_ = false;
        memo_final_ = nullptr;
        throw;
    }
    defer_u_reduce_ = false;
    memo_final_ = nullptr;
    if (u_partials_count_ < 0 && d_final != nullptr) { // a memo evaluation that left its total where the caller wants it
        partials = nullptr;
        count = -1;
    } else if (u_partials_count_ < 0) { // a memo evaluation: its total, one value (run_pipeline)
        partials = d_u_partials_.data + grid_ - 1;
        count = 1;
    } else {
        partials = d_u_partials_.data;
        count = u_partials_count_;
    }
    return true;
}

// item-splitting thresholds of the tile launches (run_pipeline has the measurements)
#ifndef TM_SPLIT4_MAX_K
#define TM_SPLIT4_MAX_K 1536
#define TM_SPLIT2_MAX_K_F32 7168
#define TM_SPLIT2_MAX_K_F64 4608
#endif
// ---- probing a second geometry on the current list (engine.hpp: ProbeTarget; the barostat's fast path) --------------------
template <typename Real>
bool NonbondedAllPairs<Real>::probe_ready(const int N, const int P, const double *d_x, const double *d_p, const double *d_box) {
    if (N != N_ || P != N_ * PARAMS_PER_ATOM || empty_) {
        return false;
    }
    sync_list_mode();
    return pre_valid_ && pre_sorted_ && d_x == pre_x_ && d_p == pre_p_ && guest_p_ == pre_guest_p_ && d_box == pre_box_ && !force_rebuild_ &&
           calls_since_sort_ % steps_per_sort_ != 0 && covers_all() && nblist_.upper_triangular() &&
           (static_list() || scale_aware()) && piggyback_table_ == nullptr && piggyback_energy_table_ == nullptr;
}

template <typename Real> ProbeTarget NonbondedAllPairs<Real>::probe_begin() {
    const Real *had = d_gathered2_.data;
    d_gathered2_.reserve(static_cast<size_t>(slot_capacity() + 1) * 8 * (merged_mode_ ? 2 : 1));
    if (merged_mode_ && d_gathered2_.data != had && guest_pad_ > guest_rows_) {
        // the holes of a merged order have no atom whose thread would write them: their records (w = 1e18, write_hole_record) once
        std::vector<Real> holes(static_cast<size_t>(guest_pad_ - guest_rows_) * 8, static_cast<Real>(0));
        for (int k = 0; k < guest_pad_ - guest_rows_; k++) {
            holes[static_cast<size_t>(k) * 8 + 3] = static_cast<Real>(1e18);
        }
        HIP_CHECK(hipMemcpy(d_gathered2_.data + static_cast<size_t>(guest_rows_) * 8, holes.data(), holes.size() * sizeof(Real), hipMemcpyHostToDevice));
    }
    d_u_partials2_.reserve(grid_);
    ProbeTarget t;
    t.gathered = d_gathered_.data;
    t.gathered2 = d_gathered2_.data;
    t.real_bytes = static_cast<int>(sizeof(Real));
    t.n = K_;
    t.slot_of_atom = d_slot_of_atom_.data;
    t.perm = d_perm_.data;
    t.snap_x = d_snap_x_.data;
    t.snap_box = d_snap_box_.data;
    t.threshold2 = rebuild_threshold2();
    t.scale_aware = (scale_aware() && !static_list()) ? 1 : 0; // (a static complete list has no snapshot to test against and no counters to reset)
    // the probe is one evaluation as far as the flag pair and the sort cadence go: its list launch consumes the flag the last
    // update kernel may have raised, the force call after it reads the other one (which the commit may raise)
    t.flag_probe = d_flags_.data + parity_;
    parity_ ^= 1;
    t.flag_next = d_flags_.data + parity_;
    calls_since_sort_++;
    t.nbl_counters = nblist_.d_counters_rw();
    t.blk_ctr = nblist_.d_col_ctr();
    t.blk_ext = nblist_.d_col_ext();
    t.second_records = merged_mode_ ? K_ + 1 : 0;
    t.cutoff = cutoff_;
    t.list_reach = static_list() ? 0.0 : cutoff_ + list_padding();
    memo_chain_ = false; // (a commit rewrites records)
    last_x_ = last_box_ = nullptr;
    probe_d_box_ = pre_box_;
    return t; // pre_valid_ / pre_sorted_ stay: the sorted records still describe (x, box), or -- after the commit -- (x', box')
}

template <typename Real>
void NonbondedAllPairs<Real>::probe_energy(
    const int which, const double *d_box_which, const FusedTable *table, const int table_blocks, const double *coords, hipStream_t stream,
    const i128 *&partials, int &count) {
    if (which == 0) {
        this->probe_list_launch(stream);
    }
    const Real *gathered = which == 0 ? d_gathered_.data : d_gathered2_.data;
    i128 *out = which == 0 ? d_u_partials_.data : d_u_partials2_.data;
    const int n_cus = grid_ / (4 * TileWaves<Real>::value);
    const int split = K_ <= TM_SPLIT4_MAX_K ? 4 : (K_ <= (sizeof(Real) == 8 ? TM_SPLIT2_MAX_K_F64 : TM_SPLIT2_MAX_K_F32) ? 2 : 1);
    using ProbeShape = TileShape<Real, tile_wide<Real, true, false, false, false>()>;
#define TM_LAUNCH_PROBE(...)                                                                                            \
    k_nonbonded_tiles<Real, true, false, false, ##__VA_ARGS__><<<n_cus * ProbeShape::wgs_per_cu, 64 * ProbeShape::waves, 0, stream>>>( \
        K_, nblist_.get_num_row_idxs(), 1, nullptr, nblist_.d_counters() + NB_COUNTER_CLASS0, nblist_.items_cap(), nblist_.d_items(), \
        nblist_.d_col_atoms(), gathered, d_box_which, beta_, cutoff_, d_es_table_, d_g_du_dx_.data, d_g_du_dp_.data, acc_stride_, out, table, \
        table_blocks, coords, nullptr, 3, 1, nullptr, d_timing_.data)
    const int prof = Profiler::get().begin("nonbonded_tiles", stream);
    if (split == 4) {
        TM_LAUNCH_PROBE(false, 4);
    } else if (split == 2) {
        TM_LAUNCH_PROBE(false, 2);
    } else {
        TM_LAUNCH_PROBE();
    }
#undef TM_LAUNCH_PROBE
    Profiler::get().end("nonbonded_tiles", prof, stream);
    HIP_CHECK(hipGetLastError());
    partials = out;
    count = n_cus * ProbeShape::wgs_per_cu;
}

template <typename Real> void NonbondedAllPairs<Real>::probe_list_launch(hipStream_t stream) {
    if (static_list() && static_list_built_) {
        return;
    }
    // the list launch an ordinary evaluation would make: rebuilds iff the update kernel -- or the proposal's own test, made by the
    // mover -- raised the flag; block bounds and counters are the sorted hand-over's (bounds_done)
    int *flag = d_flags_.data + (parity_ ^ 1); // (probe_begin has already advanced parity_)
    const int prof_list = Profiler::get().begin("nblist_build", stream);
    nblist_.build_device(d_gathered_.data, probe_d_box_, cutoff_ + list_padding(), cutoff_, flag, 0, N_ * 3, pre_x_, d_snap_x_.data, d_snap_box_.data, stream, true, false);
    Profiler::get().end("nblist_build", prof_list, stream);
}

template <typename Real>
void NonbondedAllPairs<Real>::probe_energy_dual(
    const double *d_box2, const FusedTable *table, const int table_blocks, const double *coords, const double *coords2, const float *r2_blocks,
    const int n_r2, hipStream_t stream, const i128 *&partials, const i128 *&partials2, int &count) {
    this->probe_list_launch(stream);
    const int n_cus = grid_ / (4 * TileWaves<Real>::value);
    const int split = K_ <= TM_SPLIT4_MAX_K ? 4 : (K_ <= (sizeof(Real) == 8 ? TM_SPLIT2_MAX_K_F64 : TM_SPLIT2_MAX_K_F32) ? 2 : 1);
    using DualShape = TileShape<Real, tile_wide<Real, true, false, false, true>()>;
#define TM_LAUNCH_DUAL(SPLIT)                                                                                           \
    k_nonbonded_tiles<Real, true, false, false, false, SPLIT, true><<<n_cus * DualShape::wgs_per_cu, 64 * DualShape::waves, 0, stream>>>( \
        K_, nblist_.get_num_row_idxs(), 1, nullptr, nblist_.d_counters() + NB_COUNTER_CLASS0, nblist_.items_cap(), nblist_.d_items(), \
        nblist_.d_col_atoms(), d_gathered_.data, probe_d_box_, beta_, cutoff_, d_es_table_, d_g_du_dx_.data, d_g_du_dp_.data, acc_stride_, \
        d_u_partials_.data, table, table_blocks, coords, nullptr, 3, 1, nullptr, d_timing_.data, d_gathered2_.data, d_box2, coords2,   \
        d_u_partials2_.data, r2_blocks, n_r2)
    const int prof = Profiler::get().begin("nonbonded_tiles", stream);
    if (split == 4) {
        TM_LAUNCH_DUAL(4);
    } else if (split == 2) {
        TM_LAUNCH_DUAL(2);
    } else {
        TM_LAUNCH_DUAL(1);
    }
#undef TM_LAUNCH_DUAL
    Profiler::get().end("nonbonded_tiles", prof, stream);
    HIP_CHECK(hipGetLastError());
    partials = d_u_partials_.data;
    partials2 = d_u_partials2_.data;
    count = n_cus * DualShape::wgs_per_cu;
}

template <typename Real> void NonbondedAllPairs<Real>::check_sizes(const int N, const int P) const {
    if (N != N_) {
        throw std::runtime_error(
            std::string(name_) + "::execute_device(): expected N == N_, got N=" + std::to_string(N) + ", N_=" + std::to_string(N_));
    }
    if (P != N_ * PARAMS_PER_ATOM) {
        throw std::runtime_error(
            std::string(name_) + "::execute_device(): expected P == N_*" + std::to_string(PARAMS_PER_ATOM) + ", got P=" +
            std::to_string(P) + ", N_*" + std::to_string(PARAMS_PER_ATOM) + "=" + std::to_string(N_ * PARAMS_PER_ATOM));
    }
}

template <typename Real>
void NonbondedAllPairs<Real>::run_pipeline(
    const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, const bool scatter_du_dx,
    hipStream_t stream, const bool pregathered) {
    const int tpb = DEFAULT_TPB;
    sync_list_mode();
    if (merged_mode_ && (guest_p_ == nullptr || d_du_dp != nullptr)) {
        throw std::runtime_error("NonbondedAllPairs (merged carrier): needs the group's parameters bound, and evaluates forces or energies only");
    }
    pipeline_calls_++;
    // Energy-only evaluations that gather for themselves (batches over stored frames and parameter sets: execute_batch[_sparse],
    // compute_potential_matrix, u_kln re-evaluation; fe/free_energy.py:1148-1200) are REMEMBERED: see EnergyMemo.  `trust`: the
    // device's memo describes the records as they are -- true iff the previous call into this pipeline was a memo evaluation too
    // (any other call form rewrites records without comparing, and so do the integrator's and the barostat's hand-overs).
    const bool memo_mode = g_energy_memo && d_u != nullptr && d_du_dx == nullptr && d_du_dp == nullptr && !pregathered && group_rows_ == 0;
    const bool memo_trust = memo_mode && memo_chain_;
    memo_chain_ = false; // (set again at the end of a memo evaluation)
    if (memo_mode && d_memo_.data == nullptr) {
        d_memo_.realloc(1);
        HIP_CHECK(hipMemsetAsync(d_memo_.data, 0, sizeof(EnergyMemo), stream));
        d_u_partials_b_.realloc(grid_);
    }
    EnergyMemo *memo = memo_mode ? d_memo_.data : nullptr;
    // the caller's word (hint_same_frame) that coordinates and box are the last call's, and this pipeline's own that nothing has touched
    // its state since: whatever the list kernels could do they did a call ago
    const bool same_frame = g_same_frame_hint && same_frame_now_ && hint_call_now_ == last_call_ && last_x_ != nullptr && last_x_ == d_x && last_box_ == d_box && !pregathered;
    same_frame_now_ = false;
    last_call_ = g_eval_serial;
    last_x_ = last_box_ = nullptr; // (set again where this call ends)
    pre_valid_ = false; // consumed by this call or stale after it
    // A sorted hand-over that this call does not consume may already have reset the list counters on the device (its maker
    // does that whenever it raises the rebuild flag, and the host cannot know): the list has to be rebuilt whatever the
    // check below finds.
    const bool sorted_pending = pre_sorted_;
    pre_sorted_ = false;

    // (a) every STEPS_PER_SORT calls: re-sort along the Hilbert curve; a new order invalidates the list
    // (a static list does not care that bounds / counters of an unconsumed hand-over are stale: only a new order invalidates it)
    int force = (force_rebuild_ || (sorted_pending && !pregathered && !static_list())) ? 1 : 0;
    if (calls_since_sort_ % steps_per_sort_ == 0) {
        if (!disable_hilbert_ && group_rows_ > 0) { // interaction group: each side keeps its own contiguous, sorted range
            hilbert_->sort_device(group_rows_, d_atom_idxs_.data, d_x, d_box, d_perm_.data, stream);
            hilbert_->sort_device(K_ - group_rows_, d_atom_idxs_.data + group_rows_, d_x, d_box, d_perm_.data + group_rows_, stream);
        } else if (!disable_hilbert_ && merged_mode_) { // merged carrier: [guest rows | holes | all-pairs atoms], each side sorted for itself
            hilbert_->sort_device(guest_rows_, d_atom_idxs_.data, d_x, d_box, d_perm_.data, stream);
            if (guest_pad_ > guest_rows_) {
                HIP_CHECK(hipMemsetAsync(d_perm_.data + guest_rows_, 0xff, (guest_pad_ - guest_rows_) * sizeof(unsigned int), stream)); // NB_HOLE
            }
            hilbert_->sort_device(K_ - guest_pad_, d_atom_idxs_.data + guest_pad_, d_x, d_box, d_perm_.data + guest_pad_, stream);
        } else if (!disable_hilbert_) {
            hilbert_->sort_device(K_, d_atom_idxs_.data, d_x, d_box, d_perm_.data, stream);
        } else {
            HIP_CHECK(hipMemcpyAsync(d_perm_.data, d_atom_idxs_.data, K_ * sizeof(unsigned int), hipMemcpyDeviceToDevice, stream));
        }
        force = 1;
        TM_DEBUG_SYNC("hilbert sort", stream);
    }

    // (b) K1: displacement check against the last build's snapshot + gather into Hilbert order.  The rebuild flag
    // lives on the device and is consumed on the device: the host never waits for it (the reference blocks on a
    // pinned-memory flag every call, nonbonded_all_pairs.cu:217-236).
    // On MD steps K1 is not launched at all: the integrator's update kernel has already written the new positions into
    // `gathered`, made the displacement test and zeroed the accumulator it consumed (PregatherTarget).
    int *flag_now = d_flags_.data + parity_;
    int *flag_next = d_flags_.data + (parity_ ^ 1);
    if (!pregathered && scale_aware()) {
        // a mover (barostat) changes the box by fractions of a percent between list builds: see k_check_gather_scaled
        k_check_gather_scaled<Real><<<ceil_divide(std::max(K_, 16), tpb), tpb, 0, stream>>>(
            K_, d_perm_.data, d_x, d_p, d_box, d_snap_x_.data, d_snap_box_.data, rebuild_threshold2(), flag_now, flag_next,
            d_gathered_.data, d_du_dx ? d_g_du_dx_.data : nullptr, d_du_dp ? d_g_du_dp_.data : nullptr, acc_stride_, d_slot_of_atom_.data,
            merged_mode_ ? guest_p_ : nullptr, guest_pad_, memo);
        HIP_CHECK(hipGetLastError());
    } else if (!pregathered) {
        k_check_gather<Real><<<ceil_divide(std::max(K_, 16), tpb), tpb, 0, stream>>>(
            K_, d_perm_.data, d_x, d_p, d_box, d_snap_x_.data, d_snap_box_.data, rebuild_threshold2(), flag_now,
            flag_next, d_gathered_.data, d_du_dx ? d_g_du_dx_.data : nullptr, d_du_dp ? d_g_du_dp_.data : nullptr, acc_stride_, d_slot_of_atom_.data,
            merged_mode_ ? guest_p_ : nullptr, guest_pad_, memo);
        HIP_CHECK(hipGetLastError());
        TM_DEBUG_SYNC("k_check_gather", stream);
    }

    // (c) K2 + K3: rebuild iff forced or flagged (kernels exit immediately otherwise)
    if (static_list() && static_list_built_ && !force) {
        // the complete list of this order exists and nothing can invalidate it: no list kernel on this call
    } else if (same_frame && !force) {
        // the last call's coordinates in the last call's box: the check kernel cannot have raised the flag (either that call rebuilt --
        // the snapshot is these coordinates -- or its test passed and passes again), bounds and counters stand: no list kernel either
        same_frame_skips_++;
    } else {
        const bool bounds_done = pregathered && sorted_pending && !force; // (a forced build computes its own bounds and counters)
        const int prof_list = Profiler::get().begin("nblist_build", stream); // rebuilds AND the launches that only read the flag
        nblist_.build_device(
            d_gathered_.data, d_box, cutoff_ + list_padding(), cutoff_, flag_now, force, N_ * 3, d_x, d_snap_x_.data, d_snap_box_.data, stream,
            bounds_done, !pregathered && scale_aware());
        Profiler::get().end("nblist_build", prof_list, stream);
        static_list_built_ = static_list() && force != 0;
    }

    TM_DEBUG_SYNC("list build", stream);
    // (d) K4: tile kernel
    const unsigned int *d_counters = nblist_.d_counters();
#define TM_LAUNCH_TILES(U, X, PP, ...)                                                                                 \
    launched_waves = n_cus * TileShape<Real, tile_wide<Real, U, X, PP, false>()>::wgs_per_cu; /* energy launches leave one partial sum per WORKGROUP */   \
    k_nonbonded_tiles<Real, U, X, PP, ##__VA_ARGS__><<<n_cus * TileShape<Real, tile_wide<Real, U, X, PP, false>()>::wgs_per_cu, 64 * TileShape<Real, tile_wide<Real, U, X, PP, false>()>::waves, 0, stream>>>( \
        K_, nblist_.get_num_row_idxs(), nblist_.upper_triangular() ? 1 : 0, nblist_.row_idxs_or_null(),               \
        d_counters + NB_COUNTER_CLASS0, nblist_.items_cap(), nblist_.d_items(), nblist_.d_col_atoms(), d_gathered_.data,   \
        d_box, beta_, cutoff_, d_es_table_, d_g_du_dx_.data, d_g_du_dp_.data, acc_stride_, d_u_partials_.data, pig_table, pig_blocks, d_x, pig_acc, pig_atom_stride, pig_comp_stride, pig_remap, \
        d_timing_.data)
    const int n_cus = grid_ / (4 * TileWaves<Real>::value);
    int launched_waves = 0; // workgroups of this launch = energy partials it writes (the name dates from one partial per wave)
    const int sel = (d_u ? 4 : 0) | (d_du_dx ? 2 : 0) | (d_du_dp ? 1 : 0);
    // a ForcePlan table offered through piggyback_forces() rides on the forces-only launch; any other call drops it
    // back to its owner's stand-alone path by never having accepted it (the plan only offers it for forces-only calls)
    // (an energy table offered through piggyback_energy() rides on the energy-only launch whose partial sums the caller adds up)
    const bool pig_energy = sel == 4 && defer_u_reduce_ && piggyback_energy_table_ != nullptr;
    if (piggyback_energy_table_ != nullptr && !pig_energy) {
        throw std::runtime_error("NonbondedAllPairs: a piggy-backed energy table is pending but this call is not an energy-only partial-sum evaluation");
    }
    const FusedTable *pig_table = sel == 2 ? piggyback_table_ : (pig_energy ? piggyback_energy_table_ : nullptr);
    const int pig_blocks = sel == 2 ? piggyback_blocks_ : (pig_energy ? piggyback_energy_blocks_ : 0);
    piggyback_energy_table_ = nullptr;
    piggyback_energy_blocks_ = 0;
    const bool pig_redirect = pig_table != nullptr && piggyback_redirect_;
    u64 *pig_acc = pig_redirect ? d_g_du_dx_.data : piggyback_acc_;
    const int pig_atom_stride = pig_redirect ? 1 : piggyback_atom_stride_, pig_comp_stride = pig_redirect ? acc_stride_ : piggyback_comp_stride_;
    const int *pig_remap = pig_redirect ? d_slot_of_atom_.data : nullptr;
    if (piggyback_table_ != nullptr && sel != 2) {
        throw std::runtime_error("NonbondedAllPairs: a piggy-backed force table is pending but this call is not forces-only");
    }
    piggyback_table_ = nullptr;
    piggyback_blocks_ = 0;
    if (memo_mode) {
        // (a) the all-pairs items (guest rows' items emptied: bit 1 of the order flag) -- every wave of that launch decides for itself
        // whether there is work (memo_skips_main) and leaves at once if not; (b) the guest rows' items from their own list (its "bucket
        // counts": the guest counter and the zeros behind it), with the plan's table riding -- skipped when there is neither; (c) the total
        const int split_m = K_ <= TM_SPLIT4_MAX_K ? 4 : (K_ <= (sizeof(Real) == 8 ? TM_SPLIT2_MAX_K_F64 : TM_SPLIT2_MAX_K_F32) ? 2 : 1);
        using MemoShape = TileShape<Real, tile_wide<Real, true, false, false, false>()>;
        const int n_wg = n_cus * MemoShape::wgs_per_cu;
#define TM_LAUNCH_MEMO(COUNTS, CAP, ITEMS, ORDER, OUT, TABLE, TBLOCKS, GATE, ...)                                      \
    k_nonbonded_tiles<Real, true, false, false, ##__VA_ARGS__><<<n_wg, 64 * MemoShape::waves, 0, stream>>>(               \
        K_, nblist_.get_num_row_idxs(), ORDER, nullptr, COUNTS, CAP, ITEMS, nblist_.d_col_atoms(), d_gathered_.data, d_box, beta_, cutoff_, \
        d_es_table_, d_g_du_dx_.data, d_g_du_dp_.data, acc_stride_, OUT, TABLE, TBLOCKS, d_x, nullptr, 3, 1, nullptr, d_timing_.data, \
        nullptr, nullptr, nullptr, nullptr, nullptr, 0, GATE, memo_trust ? 1 : 0)
        const int prof_m = Profiler::get().begin("nonbonded_tiles", stream);
        const int order_main = (nblist_.upper_triangular() ? 1 : 0) | (merged_mode_ ? 2 : 0);
        const unsigned int *main_counts = nblist_.d_counters() + NB_COUNTER_CLASS0;
        if (split_m == 4) {
            TM_LAUNCH_MEMO(main_counts, nblist_.items_cap(), nblist_.d_items(), order_main, d_u_partials_.data, nullptr, 0, memo, false, 4);
        } else if (split_m == 2) {
            TM_LAUNCH_MEMO(main_counts, nblist_.items_cap(), nblist_.d_items(), order_main, d_u_partials_.data, nullptr, 0, memo, false, 2);
        } else {
            TM_LAUNCH_MEMO(main_counts, nblist_.items_cap(), nblist_.d_items(), order_main, d_u_partials_.data, nullptr, 0, memo);
        }
        const bool second = merged_mode_ || pig_table != nullptr;
        if (second) {
            TM_LAUNCH_MEMO(nblist_.d_counters() + NB_COUNTER_GUEST, nblist_.guest_items_cap(), nblist_.d_guest_items(), 1, d_u_partials_b_.data, pig_table, pig_blocks, nullptr);
        }
#undef TM_LAUNCH_MEMO
        Profiler::get().end("nonbonded_tiles", prof_m, stream);
        HIP_CHECK(hipGetLastError());
        // (deferred: the caller reads ONE value -- from the end of the partials, or where it asked for the total)
        i128 *total = defer_u_reduce_ ? (memo_final_ != nullptr ? memo_final_ : d_u_partials_.data + grid_ - 1) : d_u;
        k_memo_finish<<<1, 256, 0, stream>>>(memo, memo_trust ? 1 : 0, d_box, d_u_partials_.data, n_wg, d_u_partials_b_.data, second ? n_wg : 0, total);
        HIP_CHECK(hipGetLastError());
        if (defer_u_reduce_) {
            u_partials_count_ = -1; // marks "one value at d_u_partials_ + grid_ - 1" for execute_energy_partials
        }
        calls_since_sort_++;
        parity_ ^= 1;
        force_rebuild_ = false;
        memo_chain_ = true;
        memo_skips_++;
        last_x_ = d_x;
        last_box_ = d_box;
        return;
    }
    const int prof = Profiler::get().begin("nonbonded_tiles", stream);
    // Small systems have fewer work items (about K / 2) than the launch has waves: their items are dealt in halves or
    // quarters (SPLIT) so that a launch does not last as long as one lone wave needs for a whole tile.
    // Measured on water boxes at liquid density, us per MD step, SPLIT 1 / 2 / 4 (scripts/size_sweep.py):
    //   K = 256: 29.8 / 22.2 / 18.1 (f32), 28.9 / 22.5 / 19.0 (f64);  900: 29.9 / 22.9 / 19.9, 29.8 / 23.8 / 21.7;
    //   2243: 33.9 / 27.1 / 29.7, 33.7 / 29.1 / 31.7;  3600: 35.7 / 32.0 / 35.1, 36.1 / 31.2 / 37.0;
    //   6318: 38.7 / 35.8 / 45.7, 40.1 / 42.0 / 48.4;  9000: 40.8 / 42.5 / 56.1, 44.4 / 49.5 / 61.5
    int split = K_ <= TM_SPLIT4_MAX_K ? 4 : (K_ <= (sizeof(Real) == 8 ? TM_SPLIT2_MAX_K_F64 : TM_SPLIT2_MAX_K_F32) ? 2 : 1);
#ifdef TM_SPLIT_ENV
    if (const char *e = getenv("TM_AMD_SPLIT")) {
        split = atoi(e);
    }
#endif
    switch (sel) {
    case 0: TM_LAUNCH_TILES(false, false, false); break;
    case 1: TM_LAUNCH_TILES(false, false, true); break;
    case 2: {
        // MD: the f64 kernel has a form for cutoffs that do not reach beyond the end of the electrostatic switch (all of the
        // reference's callers: cutoff == 1.2 nm); see INSIDE_SWITCH.
        const bool inside = sizeof(Real) == 8 && cutoff_ <= TM_ES_SWITCH_D;
#ifdef TM_ROWBLOCK
        if (K_ >= g_rowblock_min_k && nblist_.num_row_blocks() <= RB_MAX_ROW_BLOCKS && !merged_mode_) {
            // large systems: one workgroup per (row block, column range) unit, see kernels_nonbonded_rowblock.hip.hpp
#define TM_LAUNCH_ROWBLOCKS(INSIDE)                                                                                    \
    k_nonbonded_rowblocks<Real, INSIDE><<<n_cus * RB_WGS_PER_CU, RB_THREADS, 0, stream>>>(                             \
        K_, nblist_.get_num_row_idxs(), nblist_.upper_triangular() ? 1 : 0, nblist_.row_idxs_or_null(), nblist_.num_row_blocks(), \
        nblist_.d_row_segments(), nblist_.d_col_atoms(), d_gathered_.data, d_box, beta_, cutoff_, d_es_table_, d_g_du_dx_.data, acc_stride_, \
        pig_table, pig_blocks, d_x, pig_acc, pig_atom_stride, pig_comp_stride, pig_remap, d_timing_.data)
            if (inside) {
                if constexpr (sizeof(Real) == 8) {
                    TM_LAUNCH_ROWBLOCKS(true);
                }
            } else {
                TM_LAUNCH_ROWBLOCKS(false);
            }
#undef TM_LAUNCH_ROWBLOCKS
        } else
#endif
        if (inside) {
            if constexpr (sizeof(Real) == 8) {
                if (split == 4) {
                    TM_LAUNCH_TILES(false, true, false, true, 4);
                } else if (split == 2) {
                    TM_LAUNCH_TILES(false, true, false, true, 2);
                } else {
                    TM_LAUNCH_TILES(false, true, false, true);
                }
            }
        } else if (split == 4) {
            TM_LAUNCH_TILES(false, true, false, false, 4);
        } else if (split == 2) {
            TM_LAUNCH_TILES(false, true, false, false, 2);
        } else {
            TM_LAUNCH_TILES(false, true, false);
        }
        break;
    }
    case 3: TM_LAUNCH_TILES(false, true, true); break;
    case 4: // energy only (barostat attempts, HREX energy rows): the same deal for small systems
        if (split == 4) {
            TM_LAUNCH_TILES(true, false, false, false, 4);
        } else if (split == 2) {
            TM_LAUNCH_TILES(true, false, false, false, 2);
        } else {
            TM_LAUNCH_TILES(true, false, false);
        }
        break;
    case 5: TM_LAUNCH_TILES(true, false, true); break;
    case 6: TM_LAUNCH_TILES(true, true, false); break;
    case 7: TM_LAUNCH_TILES(true, true, true); break;
    }
#undef TM_LAUNCH_TILES
    Profiler::get().end("nonbonded_tiles", prof, stream);
    HIP_CHECK(hipGetLastError());

    TM_DEBUG_SYNC("tile kernel", stream);
    // (e) K5: back to the caller's atom order
    if (d_du_dx && scatter_du_dx) {
        k_scatter_accum<3><<<ceil_divide(K_ * 3, tpb), tpb, 0, stream>>>(K_, d_perm_.data, d_g_du_dx_.data, acc_stride_, d_du_dx);
        HIP_CHECK(hipGetLastError());
    }
    if (d_du_dp) {
        k_scatter_accum<4><<<ceil_divide(K_ * 4, tpb), tpb, 0, stream>>>(K_, d_perm_.data, d_g_du_dp_.data, acc_stride_, d_du_dp);
        HIP_CHECK(hipGetLastError());
    }
    if (d_u && defer_u_reduce_) {
        u_partials_count_ = launched_waves; // the caller adds them up together with everything else it evaluates
    } else if (d_u) {
        reduce_i128_device(d_u_partials_.data, launched_waves, d_u, stream);
    }
    TM_DEBUG_SYNC("scatter / reduce", stream);
    calls_since_sort_++;
    parity_ ^= 1;
    force_rebuild_ = false;
    last_x_ = d_x;
    last_box_ = d_box;
}

template <typename Real>
void NonbondedAllPairs<Real>::du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) {
    nb_du_dp_fixed_to_float(N, du_dp, du_dp_float);
}

template class NonbondedAllPairs<float>;
template class NonbondedAllPairs<double>;

// =============================================================================================================
// NonbondedInteractionGroup
// =============================================================================================================
template <typename Real>
void NonbondedInteractionGroup<Real>::validate_idxs(
    const int N, const std::vector<int> &row_atom_idxs, const std::vector<int> &col_atom_idxs, const bool allow_empty) {
    if (!allow_empty) {
        if (row_atom_idxs.size() == 0) {
            throw std::runtime_error("row_atom_idxs must be nonempty");
        }
        if (col_atom_idxs.size() == 0) {
            throw std::runtime_error("col_atom_idxs must be nonempty");
        }
        if (row_atom_idxs.size() == static_cast<size_t>(N)) {
            throw std::runtime_error("must be less then N(" + std::to_string(N) + ") row indices");
        }
        if (col_atom_idxs.size() == static_cast<size_t>(N)) {
            throw std::runtime_error("must be less then N(" + std::to_string(N) + ") col indices");
        }
    }
    verify_atom_idxs(N, row_atom_idxs, allow_empty);
    verify_atom_idxs(N, col_atom_idxs, allow_empty);
    const std::set<int> rows(row_atom_idxs.begin(), row_atom_idxs.end());
    for (int c : col_atom_idxs) {
        if (rows.count(c)) {
            throw std::runtime_error("row and col indices must be disjoint");
        }
    }
}

template <typename Real>
NonbondedInteractionGroup<Real>::NonbondedInteractionGroup(
    const int N, const std::vector<int> &row_atom_idxs, const std::vector<int> &col_atom_idxs, const double beta,
    const double cutoff, const bool disable_hilbert_sort, const double nblist_padding)
    : NonbondedAllPairs<Real>(N, beta, cutoff, disable_hilbert_sort, nblist_padding, typename NonbondedAllPairs<Real>::GroupTag{}) {
    validate_idxs(N, row_atom_idxs, col_atom_idxs, false);
    this->set_atom_idxs(row_atom_idxs, col_atom_idxs);
}

template <typename Real>
void NonbondedInteractionGroup<Real>::set_atom_idxs(const std::vector<int> &row_atom_idxs, const std::vector<int> &col_atom_idxs) {
    validate_idxs(this->N_, row_atom_idxs, col_atom_idxs, true);
    const std::set<int> row_set(row_atom_idxs.begin(), row_atom_idxs.end());
    std::vector<unsigned int> all(row_set.begin(), row_set.end());
    const int NR = static_cast<int>(all.size());
    const int NC = static_cast<int>(col_atom_idxs.size());
    if (NR + NC > this->N_) {
        throw std::runtime_error("number of idxs must be less than or equal to N");
    }
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemset(this->d_slot_of_atom_.data, 0xff, this->d_slot_of_atom_.size()));
    n_cols_ = NC;
    this->group_rows_ = NR;
    // reference quirk kept: an empty side, or every atom a row atom, turns the potential into a no-op
    this->empty_ = NR == 0 || NC == 0 || NR == this->N_;
    if (!this->empty_) {
        all.insert(all.end(), col_atom_idxs.begin(), col_atom_idxs.end());
        const int K = NR + NC;
        this->d_atom_idxs_.copy_from(all.data(), K);
        this->nblist_.resize(K);
        // list positions: rows are [0, NR), columns [NR, K) of the sorted order
        std::vector<unsigned int> rows(NR), cols(NC);
        std::iota(rows.begin(), rows.end(), 0u);
        std::iota(cols.begin(), cols.end(), static_cast<unsigned int>(NR));
        DeviceBuffer<unsigned int> d_rows(NR), d_cols(NC);
        d_rows.copy_from(rows.data());
        d_cols.copy_from(cols.data());
        this->nblist_.set_idxs_device(NC, NR, d_cols.data, d_rows.data, 0);
        HIP_CHECK(hipStreamSynchronize(0));
        this->K_ = K;
    }
    this->n_atoms_ = this->empty_ ? 0 : NR + NC;
    this->h_atom_idxs_ = all; // rows (ascending), then -- unless empty -- the columns as given
    this->idxs_version_++;
    this->calls_since_sort_ = 0; // next call sorts (and therefore rebuilds)
    this->force_rebuild_ = true;
}

template class NonbondedInteractionGroup<float>;
template class NonbondedInteractionGroup<double>;

// =============================================================================================================
// NonbondedPairList
// =============================================================================================================
template <typename Real, bool Negated>
NonbondedPairList<Real, Negated>::NonbondedPairList(
    const std::vector<int> &pair_idxs, const std::vector<double> &scales, const double beta, const double cutoff)
    : M_(pair_idxs.size() / 2), beta_(beta), cutoff_(cutoff) {
    if (pair_idxs.size() % 2 != 0) {
        throw std::runtime_error("pair_idxs.size() must be even, but got " + std::to_string(pair_idxs.size()));
    }
    for (int i = 0; i < M_; i++) {
        const int src = pair_idxs[i * 2 + 0], dst = pair_idxs[i * 2 + 1];
        if (src == dst) {
            throw std::runtime_error("illegal pair with src == dst: " + std::to_string(src) + ", " + std::to_string(dst));
        }
    }
    if (static_cast<int>(scales.size() / 2) != M_) {
        throw std::runtime_error(
            "expected same number of pairs and scale tuples, but got " + std::to_string(M_) + " != " + std::to_string(scales.size() / 2));
    }
    if (sizeof(Real) == 8) {
        d_es_table_ = es_force_table_device(beta_);
    }
    d_pair_idxs_.realloc(M_ * 2);
    d_scales_.realloc(M_ * 2);
    if (M_ > 0) {
        d_pair_idxs_.copy_from(pair_idxs.data());
        this->note_term_atoms(pair_idxs);
        d_scales_.copy_from(scales.data());
    }
    d_u_partials_.realloc(ceil_divide(M_, 256) * 4 + 1);
}

template <typename Real, bool Negated>
void NonbondedPairList<Real, Negated>::plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) {
    if (M_ > 0) {
        plan.add_segment(
            sizeof(Real),
            FusedSegment{Negated ? FUSED_PAIR_LIST_NEGATED : FUSED_PAIR_LIST, M_, d_pair_idxs_.data, d_p, d_scales_.data, beta_, cutoff_, nullptr, d_es_table_},
            this, P, d_p);
    }
}

template <typename Real, bool Negated>
void NonbondedPairList<Real, Negated>::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    if (M_ > 0) {
        const int tpb = 256;
        const int blocks = ceil_divide(M_, tpb);
        k_nonbonded_pair_list<Real, Negated><<<blocks, tpb, 0, stream>>>(
            M_, d_x, d_p, d_box, d_pair_idxs_.data, d_scales_.data, beta_, cutoff_, d_es_table_, d_du_dx, d_du_dp, d_u ? d_u_partials_.data : nullptr);
        HIP_CHECK(hipGetLastError());
        if (d_u) {
            reduce_i128_device(d_u_partials_.data, blocks * 4, d_u, stream);
        }
    }
}

template <typename Real, bool Negated>
void NonbondedPairList<Real, Negated>::du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) {
    nb_du_dp_fixed_to_float(N, du_dp, du_dp_float);
}

template class NonbondedPairList<float, true>;
template class NonbondedPairList<float, false>;
template class NonbondedPairList<double, true>;
template class NonbondedPairList<double, false>;

// =============================================================================================================
// NonbondedPairListPrecomputed
// =============================================================================================================
template <typename Real>
NonbondedPairListPrecomputed<Real>::NonbondedPairListPrecomputed(const std::vector<int> &pair_idxs, const double beta, const double cutoff)
    : B_(pair_idxs.size() / 2), beta_(beta), cutoff_(cutoff) {
    if (pair_idxs.size() % 2 != 0) {
        throw std::runtime_error("idxs.size() must be exactly 2*B!");
    }
    for (int b = 0; b < B_; b++) {
        const int src = pair_idxs[b * 2 + 0], dst = pair_idxs[b * 2 + 1];
        if (src == dst) {
            throw std::runtime_error("illegal pair with src == dst: " + std::to_string(src) + ", " + std::to_string(dst));
        }
    }
    d_pair_idxs_.realloc(B_ * 2);
    if (B_ > 0) {
        d_pair_idxs_.copy_from(pair_idxs.data());
        this->note_term_atoms(pair_idxs);
    }
    d_u_partials_.realloc(ceil_divide(B_, 256) * 4 + 1);
}

template <typename Real> void NonbondedPairListPrecomputed<Real>::check_size(const int P) const {
    if (P != PARAMS_PER_ATOM * B_) {
        throw std::runtime_error(
            "NonbondedPairListPrecomputed::execute_device(): expected P == " + std::to_string(PARAMS_PER_ATOM) + "*B, got P=" +
            std::to_string(P) + ", " + std::to_string(PARAMS_PER_ATOM) + "*B=" + std::to_string(PARAMS_PER_ATOM * B_));
    }
}

template <typename Real>
void NonbondedPairListPrecomputed<Real>::plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) {
    this->check_size(P);
    if (B_ > 0) {
        plan.add_segment(
            sizeof(Real), FusedSegment{FUSED_PAIR_LIST_PRECOMPUTED, B_, d_pair_idxs_.data, d_p, nullptr, beta_, cutoff_, nullptr}, this, P, d_p);
    }
}

template <typename Real>
void NonbondedPairListPrecomputed<Real>::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    this->check_size(P);
    if (B_ > 0) {
        const int blocks = ceil_divide(B_, 256);
        k_nonbonded_precomputed<Real><<<blocks, 256, 0, stream>>>(
            B_, d_x, d_p, d_box, d_pair_idxs_.data, beta_, cutoff_, d_du_dx, d_du_dp, d_u ? d_u_partials_.data : nullptr);
        HIP_CHECK(hipGetLastError());
        if (d_u) {
            reduce_i128_device(d_u_partials_.data, blocks * 4, d_u, stream);
        }
    }
}

template <typename Real>
void NonbondedPairListPrecomputed<Real>::du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) {
    nb_du_dp_fixed_to_float(B_, du_dp, du_dp_float); // rows are pairs here, same per-column exponents
}

template class NonbondedPairListPrecomputed<float>;
template class NonbondedPairListPrecomputed<double>;

// =============================================================================================================
// Debug entry: the DEVICE fixed-point conversions on caller-supplied values (tests/test_gpu_fixed_point.py compares them
// bit for bit with oracle/fixed_point.py).  Compiled in this translation unit so that they are the very functions, under
// the very flags, that the kernels above inline.
// =============================================================================================================
template <typename Real, int KIND> __global__ void k_debug_float_to_fixed(const int n, const double *__restrict__ in, u64 *__restrict__ out) {
    // no early return: the conversions contain wave-uniform branches (ballots) that every lane has to reach
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    u64 r = 0;
    if constexpr (KIND == 3) { // the nonbonded force form FIX(prefactor * delta): in = (prefactor, delta) pairs
        const Real p = live ? static_cast<Real>(in[2 * i]) : 0, d = live ? static_cast<Real>(in[2 * i + 1]) : 0;
        u64 fy, fz;
        pair_force_fixed(p, d, static_cast<Real>(0), static_cast<Real>(0), r, fy, fz);
    } else {
        const Real v = live ? static_cast<Real>(in[i]) : 0;
        if constexpr (KIND == 0) {
            r = float_to_fixed<Real>(v);
        } else if constexpr (KIND == 1) {
            r = float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DSIG>(v);
        } else {
            r = float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DEPS>(v);
        }
    }
    if (live) {
        out[i] = r;
    }
}

template <typename Real> __global__ void k_debug_float_to_fixed_energy(const int n, const double *__restrict__ in, i128 *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const i128 r = float_to_fixed_energy<Real>(i < n ? static_cast<Real>(in[i]) : 0);
    if (i < n) {
        out[i] = r;
    }
}

void debug_float_to_fixed(const int precision_bytes, const int kind, const int n, const double *h_in, u64 *h_out) {
    if (kind < 0 || kind > 3) {
        throw std::runtime_error("debug_float_to_fixed: kind must be 0 (2^36), 1 (2^37), 2 (2^38) or 3 (force product)");
    }
    const int n_in = kind == 3 ? 2 * n : n;
    DeviceBuffer<double> d_in(std::max(n_in, 1));
    DeviceBuffer<u64> d_out(std::max(n, 1));
    if (n == 0) {
        return;
    }
    d_in.copy_from(h_in, n_in);
    const int tpb = 256, blocks = ceil_divide(n, tpb);
#define TM_DBG(REAL, KIND) k_debug_float_to_fixed<REAL, KIND><<<blocks, tpb, 0, 0>>>(n, d_in.data, d_out.data)
    if (precision_bytes == 8) {
        switch (kind) {
        case 0: TM_DBG(double, 0); break;
        case 1: TM_DBG(double, 1); break;
        case 2: TM_DBG(double, 2); break;
        default: TM_DBG(double, 3); break;
        }
    } else {
        switch (kind) {
        case 0: TM_DBG(float, 0); break;
        case 1: TM_DBG(float, 1); break;
        case 2: TM_DBG(float, 2); break;
        default: TM_DBG(float, 3); break;
        }
    }
#undef TM_DBG
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(0));
    d_out.copy_to(h_out, n);
}

void debug_float_to_fixed_energy(const int precision_bytes, const int n, const double *h_in, i128 *h_out) {
    DeviceBuffer<double> d_in(std::max(n, 1));
    DeviceBuffer<i128> d_out(std::max(n, 1));
    if (n == 0) {
        return;
    }
    d_in.copy_from(h_in, n);
    const int tpb = 256, blocks = ceil_divide(n, tpb);
    if (precision_bytes == 8) {
        k_debug_float_to_fixed_energy<double><<<blocks, tpb, 0, 0>>>(n, d_in.data, d_out.data);
    } else {
        k_debug_float_to_fixed_energy<float><<<blocks, tpb, 0, 0>>>(n, d_in.data, d_out.data);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(0));
    d_out.copy_to(h_out, n);
}

} // namespace tmamd
