// Potential base-class host entry points, BoundPotential, Summed / Fanout composition, the signed-128 reduction.
// reference: cpp/src/potential.cu, bound_potential.cu, summed_potential.cu, fanout_summed_potential.cu, stream_manager.cu
#include "engine.hpp"
#include "fixed_point.hip.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <numeric>
#include <vector>

namespace tmamd {

int device_cu_count() {
    // (a function-local static with an initialiser: C++11 makes the first call thread-safe -- the enqueueing threads of
    // Context::multiple_steps_group may construct nothing, but nothing here should depend on that)
    static const int cus = []() {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    return cus;
}

// One workgroup: the inputs are a few hundred to a few thousand partials, so a single pass is launch-bound anyway.
__global__ __launch_bounds__(256) void k_reduce_i128(const i128 *__restrict__ in, const int n, i128 *__restrict__ out) {
    __shared__ i128 s_part[4];
    i128 acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        acc += in[i];
    }
    acc = wave_sum_i128(acc);
    if ((threadIdx.x & 63) == 0) {
        s_part[threadIdx.x >> 6] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    }
}

void reduce_i128_device(const i128 *d_in, int n, i128 *d_out, hipStream_t stream) {
    k_reduce_i128<<<1, 256, 0, stream>>>(d_in, n, d_out);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------
void Potential::du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) {
    for (int i = 0; i < P; i++) {
        du_dp_float[i] = fixed_to_float<double>(du_dp[i]);
    }
}

double g_last_host_call_device_ms = 0.0; // device time of the evaluations of the last execute_host_f64 call of this process (diagnostic)

Potential::~Potential() {
    if (hf_pinned_ != nullptr) {
        (void)hipHostFree(hf_pinned_);
    }
}

// fixed-point accumulators -> the doubles the binding returns (wrap_kernels.cpp:1066-1101: FIXED_TO_FLOAT, du_dp_fixed_to_float,
// convert_energy_to_fp): (double)(int64) v / 2^k is exact arithmetic on both sides, so the values equal the host conversions' bit for bit
static const int DU_DP_MAX_SPANS = 32;
struct DuDpSpanTable {
    int n;
    int offset[DU_DP_MAX_SPANS], count[DU_DP_MAX_SPANS];
};
__global__ __launch_bounds__(256) void k_outputs_to_double(
    const size_t n_dx, const u64 *__restrict__ du_dx, double *__restrict__ o_dx, const size_t n_dp, const int P, const u64 *__restrict__ du_dp,
    double *__restrict__ o_dp, const DuDpSpanTable spans, const int n_u, const i128 *__restrict__ u, double *__restrict__ o_u) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_dx; i += stride) {
        o_dx[i] = static_cast<double>(static_cast<long long>(du_dx[i])) / static_cast<double>(TM_FIXED_EXPONENT);
    }
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_dp; i += stride) {
        const int k = static_cast<int>(i % static_cast<size_t>(P)); // position in the parameter vector
        double scale = static_cast<double>(TM_FIXED_EXPONENT);
        for (int sp = 0; sp < spans.n; sp++) {
            const int r = k - spans.offset[sp];
            if (r >= 0 && r < spans.count[sp]) {
                const int col = r & 3; // (q, sig, eps, w): nonbonded_all_pairs.cu:292-308
                scale = col == 1 ? static_cast<double>(TM_FIXED_EXPONENT_DU_DSIG) : (col == 2 ? static_cast<double>(TM_FIXED_EXPONENT_DU_DEPS) : (col == 0 ? static_cast<double>(TM_FIXED_EXPONENT_DU_DCHARGE) : static_cast<double>(TM_FIXED_EXPONENT_DU_DW)));
            }
        }
        o_dp[i] = static_cast<double>(static_cast<long long>(du_dp[i])) / scale;
    }
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < static_cast<size_t>(n_u); i += stride) {
        const i128 v = u[i];
        o_u[i] = fixed_point_overflow(v) ? __builtin_nan("") : static_cast<double>(static_cast<long long>(v)) / static_cast<double>(TM_FIXED_EXPONENT);
    }
}

void Potential::execute_host_f64(
    const int coords_size, const int N, const int params_size, const int P, const int batch_size, const unsigned int *coords_batch_idxs,
    const unsigned int *params_batch_idxs, const double *h_x, const double *h_p, const double *h_box, double *h_du_dx, double *h_du_dp,
    double *h_u, const double *d_bound_p) {
    hipStream_t stream = 0;
    static const bool debug_io = std::getenv("TM_AMD_DEBUG_HOSTIO") != nullptr; // stage timings of this call on stderr
    static const bool debug_io_sync = debug_io && std::string(std::getenv("TM_AMD_DEBUG_HOSTIO")) == "sync"; // ... each stage waited for
    const auto t_begin = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {
        if (debug_io) {
            if (debug_io_sync) {
                HIP_CHECK(hipStreamSynchronize(stream));
            }
            fprintf(stderr, "[hostio] %-10s %9.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
        }
    };
    const bool dense = batch_size < 0;
    const size_t total = dense ? static_cast<size_t>(coords_size) * params_size : static_cast<size_t>(batch_size);
    const size_t n_x = static_cast<size_t>(coords_size) * N * D, n_p = d_bound_p ? 0 : static_cast<size_t>(params_size) * P, n_box = static_cast<size_t>(coords_size) * D * D;
    const size_t n_dx = h_du_dx ? total * N * D : 0, n_dp = (h_du_dp && P > 0) ? total * P : 0, n_u = h_u ? total : 0;
    // device block: [x | p | box] doubles, [u] i128 (16-byte aligned: first of the outputs), [du_dx | du_dp] u64, then the doubles [du_dx | du_dp | u]
    const size_t in_bytes = ((n_x + n_p + n_box) * sizeof(double) + 15) & ~static_cast<size_t>(15);
    const size_t fixed_bytes = n_u * sizeof(i128) + (n_dx + n_dp) * sizeof(u64);
    const size_t out_bytes = (n_dx + n_dp + n_u) * sizeof(double);
    hf_block_.reserve(in_bytes + fixed_bytes + out_bytes + 64);
    char *base = hf_block_.data;
    double *d_x = reinterpret_cast<double *>(base), *d_p = d_x + n_x, *d_box = d_p + n_p;
    i128 *d_u = reinterpret_cast<i128 *>(base + in_bytes);
    u64 *d_du_dx = reinterpret_cast<u64 *>(d_u + n_u), *d_du_dp = d_du_dx + n_dx;
    double *o_dx = reinterpret_cast<double *>(base + in_bytes + fixed_bytes), *o_dp = o_dx + n_dx, *o_u = o_dp + n_dp;
    // pinned staging: the inputs, and the outputs while they are small (large outputs go straight to the caller's arrays)
    const size_t CHUNK = static_cast<size_t>(8) << 20;
    const size_t small_out = out_bytes <= CHUNK ? out_bytes : 0;
    const size_t need = in_bytes + (small_out ? small_out : 2 * CHUNK);
    if (need > hf_pinned_bytes_) {
        if (hf_pinned_ != nullptr) {
            HIP_CHECK(hipHostFree(hf_pinned_));
            hf_pinned_ = nullptr;
        }
        hf_pinned_bytes_ = need + (need >> 2);
        HIP_CHECK(hipHostMalloc(&hf_pinned_, hf_pinned_bytes_, hipHostMallocDefault));
    }
    double *s_in = static_cast<double *>(hf_pinned_);
    std::memcpy(s_in, h_x, n_x * sizeof(double));
    if (n_p > 0) {
        std::memcpy(s_in + n_x, h_p, n_p * sizeof(double));
    }
    std::memcpy(s_in + n_x + n_p, h_box, n_box * sizeof(double));
    stamp("staged");
    HIP_CHECK(hipMemcpyAsync(d_x, s_in, (n_x + n_p + n_box) * sizeof(double), hipMemcpyHostToDevice, stream));
    if (fixed_bytes > 0) {
        HIP_CHECK(hipMemsetAsync(d_u, 0, fixed_bytes, stream)); // the kernels accumulate
    }
    // device time of the evaluations alone (diagnostic: tm_debug_last_host_call_device_ms): events behind the staging copies and
    // in front of the conversion
    static thread_local hipEvent_t ev_eval[2] = {nullptr, nullptr};
    static thread_local int ev_eval_device = -1;
    {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        if (ev_eval[0] == nullptr || ev_eval_device != dev) {
            for (hipEvent_t &e : ev_eval) {
                if (e != nullptr) {
                    (void)hipEventDestroy(e);
                }
                HIP_CHECK(hipEventCreate(&e));
            }
            ev_eval_device = dev;
        }
    }
    HIP_CHECK(hipEventRecord(ev_eval[0], stream));
    const double *p_dev = d_bound_p ? d_bound_p : (P > 0 ? d_p : nullptr);
    u64 *a_dx = n_dx ? d_du_dx : nullptr, *a_dp = n_dp ? d_du_dp : nullptr;
    i128 *a_u = n_u ? d_u : nullptr;
    if (dense) {
        this->execute_batch_device(coords_size, N, params_size, P, d_x, p_dev, d_box, a_dx, a_dp, a_u, stream);
    } else {
        this->execute_batch_sparse_device(N, P, batch_size, coords_batch_idxs, params_batch_idxs, d_x, p_dev, d_box, a_dx, a_dp, a_u, stream);
    }
    HIP_CHECK(hipEventRecord(ev_eval[1], stream));
    stamp("evaluated");
    DuDpSpanTable spans;
    spans.n = 0;
    if (n_dp) {
        std::vector<DuDpSpan> v;
        this->du_dp_nonbonded_spans(N, P, 0, v);
        if (v.size() > static_cast<size_t>(DU_DP_MAX_SPANS)) {
            throw std::runtime_error("execute_host_f64: more nonbonded parameter blocks than the conversion table holds");
        }
        for (const DuDpSpan &sp : v) {
            spans.offset[spans.n] = sp.offset;
            spans.count[spans.n] = sp.count;
            spans.n++;
        }
    }
    const size_t work = std::max(n_dx, std::max(n_dp, n_u));
    if (work > 0) {
        const int blocks = static_cast<int>(std::min<size_t>(ceil_divide(static_cast<int>(std::min<size_t>(work, 1u << 30)), 256), 2048));
        k_outputs_to_double<<<std::max(blocks, 1), 256, 0, stream>>>(n_dx, d_du_dx, o_dx, n_dp, std::max(P, 1), d_du_dp, o_dp, spans, static_cast<int>(n_u), d_u, o_u);
        HIP_CHECK(hipGetLastError());
    }
    stamp("converted");
    if (small_out > 0) {
        char *s_out = static_cast<char *>(hf_pinned_) + in_bytes;
        HIP_CHECK(hipMemcpyAsync(s_out, o_dx, out_bytes, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        const double *r = reinterpret_cast<const double *>(s_out);
        if (n_dx) {
            std::memcpy(h_du_dx, r, n_dx * sizeof(double));
        }
        if (n_dp) {
            std::memcpy(h_du_dp, r + n_dx, n_dp * sizeof(double));
        }
        if (n_u) {
            std::memcpy(h_u, r + n_dx + n_dp, n_u * sizeof(double));
        }
    } else {
        // large outputs: device -> two pinned chunks in turn (a copy into the caller's pageable array is staged by the runtime at a
        // fraction of the link's rate), each chunk moved on to the caller's array while the next one is in flight
        struct Piece {
            const char *src;
            char *dst;
            size_t bytes;
        };
        std::vector<Piece> pieces;
        auto cut = [&](const void *src, void *dst, size_t bytes) {
            for (size_t off = 0; off < bytes; off += CHUNK) {
                pieces.push_back({static_cast<const char *>(src) + off, static_cast<char *>(dst) + off, std::min(CHUNK, bytes - off)});
            }
        };
        cut(o_dx, h_du_dx, n_dx * sizeof(double));
        cut(o_dp, h_du_dp, n_dp * sizeof(double));
        cut(o_u, h_u, n_u * sizeof(double));
        static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
        static thread_local int ev_device = -1;
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        if (ev[0] == nullptr || ev_device != dev) {
            for (hipEvent_t &e : ev) {
                if (e != nullptr) {
                    (void)hipEventDestroy(e);
                }
                HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
            ev_device = dev;
        }
        char *pin[2] = {static_cast<char *>(hf_pinned_) + in_bytes, static_cast<char *>(hf_pinned_) + in_bytes + CHUNK};
        for (size_t k = 0; k <= pieces.size(); k++) {
            if (k < pieces.size()) {
                HIP_CHECK(hipMemcpyAsync(pin[k & 1], pieces[k].src, pieces[k].bytes, hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipEventRecord(ev[k & 1], stream));
            }
            if (k > 0) {
                HIP_CHECK(hipEventSynchronize(ev[(k - 1) & 1]));
                std::memcpy(pieces[k - 1].dst, pin[(k - 1) & 1], pieces[k - 1].bytes);
            }
        }
    }
    stamp("returned");
    float ms = 0.0f; // (both events have completed: the stream was waited for above)
    if (hipEventElapsedTime(&ms, ev_eval[0], ev_eval[1]) == hipSuccess) {
        g_last_host_call_device_ms = static_cast<double>(ms);
    }
}

namespace {
struct HintWithdrawn { // Potential::hint_same_frame is for ONE call
    Potential *p;
    explicit HintWithdrawn(Potential *q) : p(q) {}
    ~HintWithdrawn() { p->hint_same_frame(false); }
};
} // namespace

void Potential::execute_batch_device(
    const int coord_batch_size, const int N, const int param_batch_size, const int P, const double *d_x, const double *d_p,
    const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) {
    // outer loop over coordinates, inner over parameters: stateful children (neighbor list) see each frame once
    for (int i = 0; i < coord_batch_size; i++) {
        for (int j = 0; j < param_batch_size; j++) {
            const size_t k = static_cast<size_t>(i) * param_batch_size + j;
            HintWithdrawn withdrawn(this); // (after the call, thrown out of or not: whatever child did not run in it must not keep the hint)
            const long long prev_call = g_eval_serial; // (the entry before this one, if this loop made one: hinted entries only)
            g_eval_serial = next_eval_serial();
            if (j > 0) {
                this->hint_same_frame(true, prev_call); // (the frame's coordinates and box sit where they sat a call ago, untouched)
            }
            this->execute_device(
                N, P, d_x + static_cast<size_t>(i) * N * D, P > 0 ? d_p + static_cast<size_t>(j) * P : nullptr, d_box + i * D * D,
                d_du_dx ? d_du_dx + k * N * D : nullptr, d_du_dp ? d_du_dp + k * P : nullptr, d_u ? d_u + k : nullptr, stream);
        }
    }
}

void Potential::execute_batch_sparse_device(
    const int N, const int P, const int batch_size, const unsigned int *coords_batch_idxs, const unsigned int *params_batch_idxs,
    const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) {
    // Entries are evaluated grouped by coordinate set (stable within a group) while every result still lands in its
    // own batch slot: a stateful child (neighbor list) then rebuilds once per distinct frame however the caller ordered
    // the pairs -- the HREX energy matrix asks for each replica's frame under ~9 neighbouring parameter sets.
    std::vector<int> order(batch_size);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return coords_batch_idxs[a] < coords_batch_idxs[b]; });
    long long ic_last = -1;
    for (const int i : order) {
        const size_t ic = coords_batch_idxs[i], ip = params_batch_idxs[i];
        HintWithdrawn withdrawn(this);
        const long long prev_call = g_eval_serial;
        g_eval_serial = next_eval_serial();
        if (static_cast<long long>(ic) == ic_last) {
            this->hint_same_frame(true, prev_call);
        }
        ic_last = static_cast<long long>(ic);
        this->execute_device(
            N, P, d_x + ic * N * D, P > 0 ? d_p + ip * P : nullptr, d_box + ic * D * D,
            d_du_dx ? d_du_dx + static_cast<size_t>(i) * N * D : nullptr, d_du_dp ? d_du_dp + static_cast<size_t>(i) * P : nullptr,
            d_u ? d_u + i : nullptr, stream);
    }
}

// Shared body of the three host entry points: stage inputs, zero the accumulators, run, copy back.
struct HostStage {
    DeviceBuffer<double> &x, &p, &box;
    DeviceBuffer<u64> &du_dx, &du_dp;
    DeviceBuffer<i128> &u;
    void stage(size_t nx, const double *h_x, size_t np, const double *h_p, size_t nbox, const double *h_box, size_t n_du_dx,
               size_t n_du_dp, size_t n_u, hipStream_t stream) {
        x.reserve(nx);
        box.reserve(nbox);
        p.reserve(np);
        HIP_CHECK(hipMemcpyAsync(x.data, h_x, nx * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(box.data, h_box, nbox * sizeof(double), hipMemcpyHostToDevice, stream));
        if (np > 0) {
            HIP_CHECK(hipMemcpyAsync(p.data, h_p, np * sizeof(double), hipMemcpyHostToDevice, stream));
        }
        // the kernels accumulate: outputs must start from zero
        if (n_du_dx) {
            du_dx.reserve(n_du_dx);
            du_dx.zero_async(stream, n_du_dx);
        }
        if (n_du_dp) {
            du_dp.reserve(n_du_dp);
            du_dp.zero_async(stream, n_du_dp);
        }
        if (n_u) {
            u.reserve(n_u);
            u.zero_async(stream, n_u);
        }
        // inputs and zeroed outputs are in place before any kernel is enqueued (host entry points are not a hot path)
        HIP_CHECK(hipStreamSynchronize(stream));
    }
};

void Potential::execute_host(
    const int N, const int P, const double *h_x, const double *h_p, const double *h_box, u64 *h_du_dx, u64 *h_du_dp, i128 *h_u) {
    hipStream_t stream = 0;
    HostStage st{hs_x_, hs_p_, hs_box_, hs_du_dx_, hs_du_dp_, hs_u_};
    st.stage(static_cast<size_t>(N) * D, h_x, P, h_p, D * D, h_box, h_du_dx ? static_cast<size_t>(N) * D : 0, h_du_dp ? P : 0, h_u ? 1 : 0, stream);
    this->execute_device(
        N, P, hs_x_.data, P > 0 ? hs_p_.data : nullptr, hs_box_.data, h_du_dx ? hs_du_dx_.data : nullptr,
        (h_du_dp && P > 0) ? hs_du_dp_.data : nullptr, h_u ? hs_u_.data : nullptr, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    if (h_du_dx)
        hs_du_dx_.copy_to(h_du_dx, static_cast<size_t>(N) * D);
    if (h_du_dp && P > 0)
        hs_du_dp_.copy_to(h_du_dp, P);
    if (h_u)
        hs_u_.copy_to(h_u, 1);
}

void Potential::execute_batch_host(
    const int coord_batch_size, const int N, const int param_batch_size, const int P, const double *h_x, const double *h_p,
    const double *h_box, u64 *h_du_dx, u64 *h_du_dp, i128 *h_u) {
    hipStream_t stream = 0;
    const size_t total = static_cast<size_t>(coord_batch_size) * param_batch_size;
    HostStage st{hs_x_, hs_p_, hs_box_, hs_du_dx_, hs_du_dp_, hs_u_};
    st.stage(
        static_cast<size_t>(coord_batch_size) * N * D, h_x, static_cast<size_t>(param_batch_size) * P, h_p,
        static_cast<size_t>(coord_batch_size) * D * D, h_box, h_du_dx ? total * N * D : 0, h_du_dp ? total * P : 0, h_u ? total : 0, stream);
    this->execute_batch_device(
        coord_batch_size, N, param_batch_size, P, hs_x_.data, P > 0 ? hs_p_.data : nullptr, hs_box_.data,
        h_du_dx ? hs_du_dx_.data : nullptr, (h_du_dp && P > 0) ? hs_du_dp_.data : nullptr, h_u ? hs_u_.data : nullptr, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    if (h_du_dx)
        hs_du_dx_.copy_to(h_du_dx, total * N * D);
    if (h_du_dp && P > 0)
        hs_du_dp_.copy_to(h_du_dp, total * P);
    if (h_u)
        hs_u_.copy_to(h_u, total);
}

void Potential::execute_batch_sparse_host(
    const int coords_size, const int N, const int params_size, const int P, const int batch_size,
    const unsigned int *coords_batch_idxs, const unsigned int *params_batch_idxs, const double *h_x, const double *h_p,
    const double *h_box, u64 *h_du_dx, u64 *h_du_dp, i128 *h_u) {
    hipStream_t stream = 0;
    const size_t total = batch_size;
    HostStage st{hs_x_, hs_p_, hs_box_, hs_du_dx_, hs_du_dp_, hs_u_};
    st.stage(
        static_cast<size_t>(coords_size) * N * D, h_x, static_cast<size_t>(params_size) * P, h_p, static_cast<size_t>(coords_size) * D * D,
        h_box, h_du_dx ? total * N * D : 0, h_du_dp ? total * P : 0, h_u ? total : 0, stream);
    this->execute_batch_sparse_device(
        N, P, batch_size, coords_batch_idxs, params_batch_idxs, hs_x_.data, P > 0 ? hs_p_.data : nullptr, hs_box_.data,
        h_du_dx ? hs_du_dx_.data : nullptr, (h_du_dp && P > 0) ? hs_du_dp_.data : nullptr, h_u ? hs_u_.data : nullptr, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    if (h_du_dx)
        hs_du_dx_.copy_to(h_du_dx, total * N * D);
    if (h_du_dp && P > 0)
        hs_du_dp_.copy_to(h_du_dp, total * P);
    if (h_u)
        hs_u_.copy_to(h_u, total);
}

// ------------------------------------------------------------------------------------------------------------
BoundPotential::BoundPotential(std::shared_ptr<Potential> potential, const std::vector<double> &params)
    : size(params.size()), d_p(params.size()), potential(potential) {
    set_params(params);
}

void BoundPotential::set_params(const std::vector<double> &params) {
    if (params.size() != d_p.length) {
        throw std::runtime_error(
            "parameter size is not equal to device buffer size: " + std::to_string(params.size()) + " != " + std::to_string(d_p.length));
    }
    if (params.size() > 0) {
        d_p.copy_from(params.data());
    }
    this->size = params.size();
    this->potential->invalidate_cached_inputs(); // new values behind the same device pointer
}

void BoundPotential::set_params_prefix(const std::vector<double> &params) {
    if (params.size() > d_p.length) {
        throw std::runtime_error(
            "parameter size is greater than device buffer size: " + std::to_string(params.size()) + " > " + std::to_string(d_p.length));
    }
    if (params.size() > 0) {
        d_p.copy_from(params.data(), params.size());
    }
    this->size = params.size();
    this->potential->invalidate_cached_inputs();
}

void BoundPotential::set_params_device(const int new_size, const double *d_new_params, hipStream_t stream) {
    if (static_cast<size_t>(new_size) > d_p.length) {
        throw std::runtime_error(
            "parameter size is greater than device buffer size: " + std::to_string(new_size) + " > " + std::to_string(d_p.length));
    }
    HIP_CHECK(hipMemcpyAsync(d_p.data, d_new_params, new_size * sizeof(double), hipMemcpyDeviceToDevice, stream));
    this->size = new_size;
    this->potential->invalidate_cached_inputs();
}

void BoundPotential::execute_device(
    const int N, const double *d_x, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) {
    this->potential->execute_device(N, this->size, d_x, this->size > 0 ? this->d_p.data : nullptr, d_box, d_du_dx, d_du_dp, d_u, stream);
}

void BoundPotential::execute_host(const int N, const double *h_x, const double *h_box, u64 *h_du_dx, i128 *h_u) {
    this->execute_batch_host(1, N, h_x, h_box, h_du_dx, h_u);
}

void BoundPotential::execute_batch_host(
    const int coord_batch_size, const int N, const double *h_x, const double *h_box, u64 *h_du_dx, i128 *h_u) {
    const int D = 3;
    hipStream_t stream = 0;
    const size_t total = coord_batch_size;
    hs_x_.reserve(total * N * D);
    hs_box_.reserve(total * D * D);
    HIP_CHECK(hipMemcpyAsync(hs_x_.data, h_x, total * N * D * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(hs_box_.data, h_box, total * D * D * sizeof(double), hipMemcpyHostToDevice, stream));
    if (h_du_dx) {
        hs_du_dx_.reserve(total * N * D);
        hs_du_dx_.zero_async(stream, total * N * D);
    }
    if (h_u) {
        hs_u_.reserve(total);
        hs_u_.zero_async(stream, total);
    }
    this->potential->execute_batch_device(
        coord_batch_size, N, 1, this->size, hs_x_.data, this->size > 0 ? this->d_p.data : nullptr, hs_box_.data,
        h_du_dx ? hs_du_dx_.data : nullptr, nullptr, h_u ? hs_u_.data : nullptr, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    if (h_du_dx)
        hs_du_dx_.copy_to(h_du_dx, total * N * D);
    if (h_u)
        hs_u_.copy_to(h_u, total);
}

// ------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------
SummedPotential::SummedPotential(
    const std::vector<std::shared_ptr<Potential>> potentials, const std::vector<int> params_sizes, const bool parallel)
    : potentials_(potentials), params_sizes_(params_sizes), P_(std::accumulate(params_sizes.begin(), params_sizes.end(), 0)),
      parallel_(parallel) {
    if (potentials_.size() != params_sizes_.size()) {
        throw std::runtime_error("number of potentials != number of parameter sizes");
    }
    d_u_buffer_.realloc(potentials_.size());
}

void SummedPotential::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    if (P != P_) {
        throw std::runtime_error(
            "SummedPotential::execute_device(): expected " + std::to_string(P_) + " parameters, got " + std::to_string(P));
    }
    const int n = potentials_.size();
    if (d_du_dx && !d_du_dp && !d_u) {
        // forces only (the MD path): fuse the short per-term kernels of all children into one launch
        plan_.clear();
        this->plan_forces(N, P, d_p, plan_);
        plan_.run(N, d_x, d_box, d_du_dx, stream);
        return;
    }
    if (d_u && !d_du_dx && !d_du_dp) {
        // energy only (barostat attempts, HREX energy matrices, frame energies): the same plan, evaluated for energies --
        // one launch for all short terms, one reduction for everything (ForcePlan::run_energy)
        plan_.clear();
        this->plan_forces(N, P, d_p, plan_);
        plan_.run_energy(N, d_x, d_box, d_u, stream);
        return;
    }
    if (d_u) {
        d_u_buffer_.zero_async(stream, n);
    }
    // Children run one after the other on the caller's stream whatever `parallel` says (see engine.hpp).
    int offset = 0;
    static const bool debug_children = std::getenv("TM_AMD_DEBUG_CHILDREN") != nullptr;
    for (int i = 0; i < n; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        potentials_[i]->execute_device(
            N, params_sizes_[i], d_x, d_p + offset, d_box, d_du_dx, d_du_dp == nullptr ? nullptr : d_du_dp + offset,
            d_u == nullptr ? nullptr : d_u_buffer_.data + i, stream);
        if (debug_children) {
            fprintf(stderr, "[child %d] enqueue %.1f us\n", i, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        offset += params_sizes_[i];
    }
    if (d_u) {
        reduce_i128_device(d_u_buffer_.data, n, d_u, stream);
    }
}

void SummedPotential::plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) {
    if (P != P_) {
        throw std::runtime_error(
            "SummedPotential::execute_device(): expected " + std::to_string(P_) + " parameters, got " + std::to_string(P));
    }
    // `parallel` asks for children on concurrent streams; it is an execution hint with no effect on results, and one
    // fused launch beats several concurrent tiny ones (measured), so the forces-only plan supersedes it
    int offset = 0;
    for (size_t i = 0; i < potentials_.size(); i++) {
        potentials_[i]->plan_forces(N, params_sizes_[i], d_p + offset, plan);
        offset += params_sizes_[i];
    }
}

void SummedPotential::du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) {
    int offset = 0;
    for (size_t i = 0; i < potentials_.size(); i++) {
        potentials_[i]->du_dp_fixed_to_float(N, params_sizes_[i], du_dp + offset, du_dp_float + offset);
        offset += params_sizes_[i];
    }
}

FanoutSummedPotential::FanoutSummedPotential(const std::vector<std::shared_ptr<Potential>> potentials, const bool parallel)
    : potentials_(potentials), parallel_(parallel), d_u_buffer_(potentials.size()) {}

void FanoutSummedPotential::execute_device(
    const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u,
    hipStream_t stream) {
    const int n = potentials_.size();
    if (d_du_dx && !d_du_dp && !d_u) {
        plan_.clear();
        this->plan_forces(N, P, d_p, plan_);
        plan_.run(N, d_x, d_box, d_du_dx, stream);
        return;
    }
    if (d_u && !d_du_dx && !d_du_dp) {
        plan_.clear();
        this->plan_forces(N, P, d_p, plan_);
        plan_.run_energy(N, d_x, d_box, d_u, stream);
        return;
    }
    if (d_u) {
        d_u_buffer_.zero_async(stream, n);
    }
    for (int i = 0; i < n; i++) {
        potentials_[i]->execute_device(N, P, d_x, d_p, d_box, d_du_dx, d_du_dp, d_u == nullptr ? nullptr : d_u_buffer_.data + i, stream);
    }
    if (d_u) {
        reduce_i128_device(d_u_buffer_.data, n, d_u, stream);
    }
}

void FanoutSummedPotential::plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) {
    for (auto &pot : potentials_) {
        pot->plan_forces(N, P, d_p, plan);
    }
}

void FanoutSummedPotential::du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) {
    if (!potentials_.empty()) {
        potentials_[0]->du_dp_fixed_to_float(N, P, du_dp, du_dp_float);
    }
}

void collect_nonbonded_cutoffs(const std::shared_ptr<Potential> &pot, std::vector<double> &out) {
    if (auto nb = std::dynamic_pointer_cast<NonbondedAllPairsBase>(pot)) {
        out.push_back(nb->get_cutoff() + nb->get_nblist_padding());
    } else if (auto f = std::dynamic_pointer_cast<FanoutSummedPotential>(pot)) {
        for (auto &c : f->get_potentials())
            collect_nonbonded_cutoffs(c, out);
    } else if (auto s = std::dynamic_pointer_cast<SummedPotential>(pot)) {
        for (auto &c : s->get_potentials())
            collect_nonbonded_cutoffs(c, out);
    }
}

} // namespace tmamd

// ------------------------------------------------------------------------------------------------------------
#include "profiler.hpp"
namespace tmamd {

Profiler &Profiler::get() {
    static Profiler p;
    return p;
}

int Profiler::begin(const char *name, hipStream_t stream) {
    if (!enabled_)
        return -1;
    auto &v = events_[name];
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, stream));
    v.emplace_back(a, b);
    return static_cast<int>(v.size()) - 1;
}

void Profiler::end(const char *name, int idx, hipStream_t stream) {
    if (idx < 0)
        return;
    HIP_CHECK(hipEventRecord(events_[name][idx].second, stream));
}

void Profiler::read(const char *name, double *total_ms, long long *launches) {
    HIP_CHECK(hipDeviceSynchronize());
    double total = 0;
    long long n = 0;
    auto it = events_.find(name);
    if (it != events_.end()) {
        for (auto &pr : it->second) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
            total += ms;
            n++;
        }
    }
    *total_ms = total;
    *launches = n;
}

void Profiler::reset() {
    (void)hipDeviceSynchronize();
    for (auto &kv : events_) {
        for (auto &pr : kv.second) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    }
    events_.clear();
}

} // namespace tmamd
