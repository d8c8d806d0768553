// Microbenchmark: can a small kernel be taken off a step's critical path by running it on a SECOND stream underneath the long kernel?
// The MD step is  update(t-1) -> [list launch that only reads a flag: ~4 us] -> tiles(t): 55 us -> update(t).  Forked form:
//   stream A: update -> (event E1) -> small -> (wait E2) -> update ...        stream B: (wait E1) -> long -> (event E2)
// i.e. the small kernel and the long one both start behind the update; the next update waits for both.  What two cross-stream
// dependencies per step cost decides whether this can pay.   hipcc --offload-arch=gfx950 -O3 scripts/microbench/fork_join.hip -o ... 
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_spin(long long cycles, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
    }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) {
        atomicAdd(sink, 1);
    }
}

int main() {
    int *d_sink;
    CHECK(hipMalloc(&d_sink, 4));
    hipStream_t a, b;
    CHECK(hipStreamCreate(&a));
    CHECK(hipStreamCreate(&b));
    hipEvent_t e1, e2, t0, t1;
    CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    CHECK(hipEventCreate(&t0));
    CHECK(hipEventCreate(&t1));
    const long long GHZ_TICKS_PER_US = 100; // wall_clock64: the 100 MHz constant clock
    const int steps = 2000;
    for (int variant = 0; variant < 3; variant++) {
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(t0, a));
            for (int s = 0; s < steps; s++) {
                // update: 369 small workgroups, ~5 us
                hipLaunchKernelGGL(k_spin, dim3(369), dim3(64), 0, a, 5 * GHZ_TICKS_PER_US, d_sink);
                if (variant == 0) { // serial: small (737 workgroups that exit at once) then long (256 x 1024 threads, 50 us)
                    hipLaunchKernelGGL(k_spin, dim3(737), dim3(1024), 0, a, 0, (int *)nullptr);
                    hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, a, 50 * GHZ_TICKS_PER_US, d_sink);
                } else if (variant == 1) { // no small kernel at all (the bound)
                    hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, a, 50 * GHZ_TICKS_PER_US, d_sink);
                } else { // forked
                    CHECK(hipEventRecord(e1, a));
                    CHECK(hipStreamWaitEvent(b, e1, 0));
                    hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, b, 50 * GHZ_TICKS_PER_US, d_sink);
                    CHECK(hipEventRecord(e2, b));
                    hipLaunchKernelGGL(k_spin, dim3(737), dim3(1024), 0, a, 0, (int *)nullptr);
                    CHECK(hipStreamWaitEvent(a, e2, 0));
                }
            }
            CHECK(hipEventRecord(t1, a));
            CHECK(hipEventSynchronize(t1));
            CHECK(hipStreamSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, t0, t1));
            if (rep == 1) {
                printf("%-52s %7.2f us per step\n", variant == 0 ? "serial: update, flag-only launch, long kernel" : variant == 1 ? "without the flag-only launch (bound)" : "forked: long kernel on a second stream, joined", 1e3 * ms / steps);
            }
        }
    }
    return 0;
}
