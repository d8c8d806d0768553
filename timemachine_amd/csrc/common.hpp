// Host-side plumbing shared by every translation unit: error handling, RAII device buffers, small math.
// gfx950 / ROCm only -- no CUDA dual paths.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace tmamd {

typedef unsigned long long u64;
typedef __int128 i128;

// Raised for "no usable GPU" conditions; surfaces in Python as custom_ops.InvalidHardware
// (reference: cpp/src/exceptions.hpp, cpp/src/gpu_utils.cuh:27-53).
struct InvalidHardware : public std::runtime_error {
    explicit InvalidHardware(const std::string &m) : std::runtime_error(m) {}
};

inline void hip_check(hipError_t code, const char *file, int line) {
    if (code == hipSuccess)
        return;
    std::string msg = std::string("HIP error: ") + hipGetErrorString(code) + " (" + file + ":" + std::to_string(line) + ")";
    switch (code) {
    case hipErrorNoDevice:
    case hipErrorInvalidDevice:
    case hipErrorInsufficientDriver:
    case hipErrorNotInitialized:
        throw InvalidHardware(msg);
    default:
        // The reference exit()s the process here (gpu_utils.cuh:46-52); a replacement may throw instead.
        throw std::runtime_error(msg);
    }
}
#define HIP_CHECK(x) ::tmamd::hip_check((x), __FILE__, __LINE__)

template <typename T> inline T ceil_divide(T x, T y) { return (x + y - 1) / y; }

// RAII hipMalloc. Grow-only `reserve` lets host entry points reuse scratch instead of malloc/free per call.
#ifdef TM_GUARD
// Debug builds (-DTM_GUARD): every device buffer sits between two 64 KiB guard zones filled with 0xA5; guard_check()
// reports the zones that no longer are (out-of-bounds writes by any kernel).  tm_debug_check_guards() in the C ABI.
struct GuardRegistry {
    struct Zone {
        char *base;
        size_t bytes;
    };
    std::vector<Zone> zones;
    static GuardRegistry &get() {
        static GuardRegistry r;
        return r;
    }
    static constexpr size_t GUARD = 65536;
    void *alloc(size_t bytes) {
        char *base = nullptr;
        HIP_CHECK(hipMalloc(&base, bytes + 2 * GUARD));
        HIP_CHECK(hipMemset(base, 0xA5, GUARD));
        HIP_CHECK(hipMemset(base + GUARD + bytes, 0xA5, GUARD));
        zones.push_back({base, bytes});
        if (std::getenv("TM_GUARD_TRACE")) {
            fprintf(stderr, "[alloc] %p .. %p  (%zu bytes)\n", static_cast<void *>(base + GUARD), static_cast<void *>(base + GUARD + bytes), bytes);
        }
        return base + GUARD;
    }
    void free(void *p) {
        char *base = static_cast<char *>(p) - GUARD;
        for (size_t i = 0; i < zones.size(); i++) {
            if (zones[i].base == base) {
                zones.erase(zones.begin() + i);
                break;
            }
        }
        (void)hipFree(base);
    }
    int check() {
        int bad = 0;
        std::vector<unsigned char> h(GUARD);
        HIP_CHECK(hipDeviceSynchronize());
        for (const Zone &z : zones) {
            for (int side = 0; side < 2; side++) {
                HIP_CHECK(hipMemcpy(h.data(), side ? z.base + GUARD + z.bytes : z.base, GUARD, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < GUARD; i++) {
                    if (h[i] != 0xA5) {
                        fprintf(stderr, "guard violated: buffer of %zu bytes, %s zone, offset %zu\n", z.bytes, side ? "upper" : "lower", i);
                        bad++;
                        break;
                    }
                }
            }
        }
        return bad;
    }
};
#endif

template <typename T> struct DeviceBuffer {
    T *data = nullptr;
    size_t length = 0;

    DeviceBuffer() {}
    explicit DeviceBuffer(size_t n) { realloc(n); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() {
        if (data)
            release(data);
    }
    static void release(T *p) {
#ifdef TM_GUARD
        GuardRegistry::get().free(p);
#else
        (void)hipFree(p);
#endif
    }
    size_t size() const { return length * sizeof(T); }
    void realloc(size_t n) {
        if (data) {
            release(data);
            data = nullptr;
        }
        length = n;
        // never hand back nullptr for an empty buffer: kernels take the pointer even when nothing is read
#ifdef TM_GUARD
        data = static_cast<T *>(GuardRegistry::get().alloc((n > 0 ? n : 1) * sizeof(T)));
#else
        HIP_CHECK(hipMalloc(&data, (n > 0 ? n : 1) * sizeof(T)));
#endif
    }
    void reserve(size_t n) {
        if (n > length || data == nullptr)
            realloc(n);
    }
    void copy_from(const T *host, size_t n) { HIP_CHECK(hipMemcpy(data, host, n * sizeof(T), hipMemcpyHostToDevice)); }
    void copy_from(const T *host) { copy_from(host, length); }
    void copy_to(T *host, size_t n) const { HIP_CHECK(hipMemcpy(host, data, n * sizeof(T), hipMemcpyDeviceToHost)); }
    void copy_to(T *host) const { copy_to(host, length); }
    void zero_async(hipStream_t s, size_t n) { HIP_CHECK(hipMemsetAsync(data, 0, n * sizeof(T), s)); }
};

static const int PARAMS_PER_ATOM = 4; // (q*sqrt(ONE_4PI_EPS0), sigma/2, sqrt(eps), w) -- cpp/src/nonbonded_common.hpp
static const int TILE = 32;           // row-block size exposed by the Neighborlist API (wrap_kernels.cpp:125-128)
static const int DEFAULT_TPB = 256;

} // namespace tmamd
