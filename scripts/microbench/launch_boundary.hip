// What does a kernel boundary cost on this machine, and does a captured HIP graph make it cheaper?
// Three dependent kernels per "step" shaped like the MD step's (a 737 x 1024-thread launch that reads one flag and leaves, a
// 256 x 1024-thread launch that does ~50 us of dependent FMAs, a 369 x 64-thread launch that touches 23.5k atoms), enqueued
// (a) as plain stream launches and (b) as a graph of STEPS steps captured once and replayed.  Prints us per step for both, and
// the same with the middle kernel removed (pure boundary cost).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_boundary scripts/microbench/launch_boundary.hip && /tmp/launch_boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_flag(const int *flag, int *out) {
    if (*flag) {
        out[blockIdx.x * blockDim.x + threadIdx.x] = 1;
    }
}
__global__ void k_work(double *buf, const int iters) {
    double a = buf[threadIdx.x], b = 1.0000001;
    for (int i = 0; i < iters; i++) {
        a = __builtin_fma(a, b, 1e-9);
    }
    if (a == 123.456) {
        buf[threadIdx.x] = a;
    }
}
__global__ void k_update(const int n, const double *x, double *y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        y[i] = x[i] * 1.0000001 + 1e-9;
    }
}
int main() {
    const int STEPS = 100, N = 23559 * 3;
    int *flag, *out;
    double *buf, *x, *y;
    CHECK(hipMalloc(&flag, 4));
    CHECK(hipMemset(flag, 0, 4));
    CHECK(hipMalloc(&out, 737 * 1024 * 4));
    CHECK(hipMalloc(&buf, 1024 * 8));
    CHECK(hipMemset(buf, 0, 1024 * 8));
    CHECK(hipMalloc(&x, N * 8));
    CHECK(hipMalloc(&y, N * 8));
    CHECK(hipMemset(x, 0, N * 8));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int with_work = 1; with_work >= 0; with_work--) {
        auto enqueue = [&](int steps) {
            for (int k = 0; k < steps; k++) {
                k_flag<<<737, 1024, 0, s>>>(flag, out);
                if (with_work) {
                    k_work<<<256, 1024, 0, s>>>(buf, 6000);
                }
                k_update<<<(N + 63) / 64, 64, 0, s>>>(N, (k & 1) ? y : x, (k & 1) ? x : y);
            }
        };
        enqueue(STEPS); // warm up
        CHECK(hipStreamSynchronize(s));
        float ms_plain = 0, ms_graph = 0;
        CHECK(hipEventRecord(e0, s));
        for (int r = 0; r < 10; r++) {
            enqueue(STEPS);
        }
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_plain, e0, e1));
        hipGraph_t g;
        hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        enqueue(STEPS);
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CHECK(hipGraphLaunch(ge, s));
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventRecord(e0, s));
        for (int r = 0; r < 10; r++) {
            CHECK(hipGraphLaunch(ge, s));
        }
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_graph, e0, e1));
        printf("%s: plain stream launches %.2f us per step, captured graph (%d steps per launch) %.2f us per step\n",
               with_work ? "flag + 50 us work + update" : "flag + update only          ", 1e3 * ms_plain / (10 * STEPS), STEPS, 1e3 * ms_graph / (10 * STEPS));
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
    }
    return 0;
}
