// Microbenchmark: what a PERSISTENT multi-step MD kernel would pay per step for its two device-wide barriers and for handing data
// from one phase's writers to the next phase's readers across the eight (mutually non-coherent) XCD L2s of an MI355X -- the cost
// floor of "several steps of a small system in ONE launch" (VERDICT round 4, item 7), measured before building it.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/persistent_step.hip -o gpurun_out/persistent_step && gpurun_out/persistent_step
// Per step: phase A every thread publishes one double, barrier, phase B every thread reads a double published by ANOTHER workgroup
// (and checks it), barrier.  Variants: how the data moves (agent-scope atomic load / store, i.e. past the L2s; or plain accesses
// bracketed by agent-scope release / acquire fences = L2 write-back + invalidate) and how many workgroups take part.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned int *counter, const unsigned int target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// the barrier of MODE 3: no cache maintenance at all -- a workgroup-scope release (the wave's own memory operations have completed) and a
// RELAXED spin (an acquire load at agent scope invalidates the whole L2 of its XCD on every iteration)
__device__ __forceinline__ void grid_barrier_light(unsigned int *counter, const unsigned int target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

template <int MODE> // 0: atomic load / store of the data; 1: plain accesses + fences; 2: barriers only (no data); 3: plain accesses to UNCACHED memory, light barrier
__global__ __launch_bounds__(256) void k_persistent(const int steps, unsigned int *counter, double *data, unsigned int *errors) {
    const unsigned int G = gridDim.x, tid = blockIdx.x * 256 + threadIdx.x, n = G * 256;
    unsigned int bad = 0;
    for (int s = 0; s < steps; s++) {
        const double v = static_cast<double>(s) * 1024.0 + 1.0;
        if (MODE == 0) {
            __hip_atomic_store(data + tid, v + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 1) {
            data[tid] = v + tid;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        } else if (MODE == 3) {
            data[tid] = v + tid;
        }
        if (MODE == 3) {
            grid_barrier_light(counter, (2u * s + 1u) * G);
        } else {
            grid_barrier(counter, (2u * s + 1u) * G);
        }
        const unsigned int src = (tid + 256u * (1u + (G > 8 ? 8u : 0u) / 2u) + 77u) % n; // another workgroup's slot (another XCD when G > 8)
        double got = 0;
        if (MODE == 0) {
            got = __hip_atomic_load(data + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            got = data[src];
        } else if (MODE == 3) {
            got = *const_cast<volatile double *>(data + src);
        }
        bad += (MODE != 2 && got != v + src) ? 1u : 0u;
        if (MODE == 3) {
            grid_barrier_light(counter, (2u * s + 2u) * G);
        } else {
            grid_barrier(counter, (2u * s + 2u) * G);
        }
    }
    if (bad) {
        atomicAdd(errors, bad);
    }
}

template <int MODE> int run(const char *name, const int G, const int steps, unsigned int *d_counter, double *d_data, unsigned int *d_err) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    unsigned int h_err = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipMemset(d_counter, 0, 4));
        CHECK(hipMemset(d_err, 0, 4));
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_persistent<MODE>, dim3(G), dim3(256), 0, 0, steps, d_counter, d_data, d_err);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
        CHECK(hipMemcpy(&h_err, d_err, 4, hipMemcpyDeviceToHost));
    }
    printf("%-44s G = %3d workgroups: %6.2f us per step (2 barriers + publish + read), %u wrong reads\n", name, G, 1e3 * best / steps, h_err);
    return 0;
}

int main() {
    unsigned int *d_counter, *d_err;
    double *d_data;
    CHECK(hipMalloc(&d_counter, 4));
    CHECK(hipMalloc(&d_err, 4));
    CHECK(hipMalloc(&d_data, 256 * 256 * sizeof(double)));
    double *d_uncached;
    CHECK(hipExtMallocWithFlags(reinterpret_cast<void **>(&d_uncached), 256 * 256 * sizeof(double), hipDeviceMallocUncached));
    const int steps = 2000;
    for (int G : {4, 8, 16, 32, 64, 128, 256}) {
        if (run<3>("plain accesses to UNCACHED memory, light barrier", G, steps, d_counter, d_uncached, d_err)) return 1;
        if (run<2>("barriers only", G, steps, d_counter, d_data, d_err)) return 1;
        if (run<0>("agent-scope atomic load / store of the data", G, steps, d_counter, d_data, d_err)) return 1;
        if (run<1>("plain accesses + agent release / acquire fences", G, steps, d_counter, d_data, d_err)) return 1;
    }
    return 0;
}
