// Microbenchmark: LDS atomic throughput on gfx950 (cycles per wave-level instruction, all CUs busy, 16 waves/CU).
// hipcc --offload-arch=gfx950 -O3 scripts/microbench/lds_atomics.hip -o gpurun_out/lds_atomics && gpurun_out/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE> __global__ __launch_bounds__(64) void k(long long *out, int iters, int stride_mode) {
    __shared__ unsigned long long s64[1024];
    unsigned int *s32 = reinterpret_cast<unsigned int *>(s64);
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) s64[i] = 0;
    __syncthreads();
    // address pattern: 0 = all lanes distinct & conflict-free, 1 = lanes l and l+32 share an address, 2 = random-ish
    int idx = lane;
    if (stride_mode == 1) idx = lane & 31;
    if (stride_mode == 2) idx = (lane * 37 + 11) & 63;
    if (stride_mode == 3) idx = lane & 15;               // every address hit by 4 lanes
    if (stride_mode == 4) idx = lane & 7;                // 8 lanes per address
    if (stride_mode == 5) idx = (lane * 2654435761u >> 27) & 31; // hashed onto 32 addresses (a realistic batch: ~2 per row, max ~5)
    if (stride_mode == 6) idx = lane * 2;                // distinct addresses, stride 16 B (u64 view): 2-way bank conflict
    unsigned long long v = lane + 1;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { // 6 x ds_add_u64
#pragma unroll
            for (int c = 0; c < 6; c++) __hip_atomic_fetch_add(&s64[c * 64 + idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 1) { // 18 x ds_add_u32
#pragma unroll
            for (int c = 0; c < 18; c++) __hip_atomic_fetch_add(&s32[c * 64 + idx], (unsigned int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 2) { // 6 x (ds_read_b64 + add + ds_write_b64), non-atomic
#pragma unroll
            for (int c = 0; c < 6; c++) s64[c * 64 + idx] += v;
        } else if (MODE == 3) { // 6 x ds_add_u32
#pragma unroll
            for (int c = 0; c < 6; c++) __hip_atomic_fetch_add(&s32[c * 64 + idx], (unsigned int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        v += 3;
    }
    __syncthreads();
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    if (s64[lane] == 0x1234567) out[0] = 0; // keep results alive
}

template <int MODE> void run(const char *name, int ninstr) {
    const int grid = 256 * 16, iters = 2000;
    long long *d;
    hipMalloc(&d, grid * sizeof(long long));
    for (int sm = 0; sm < 7; sm++) {
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, sm);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, sm);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<long long> h(grid);
        hipMemcpy(h.data(), d, grid * sizeof(long long), hipMemcpyDeviceToHost);
        double mean = 0; for (auto x : h) mean += x; mean /= grid;
        // CU-level cost: 16 waves per CU share one LDS: cycles per wave-instruction = kernel cycles / (16 waves * iters * ninstr)
        double cyc_per_instr_cu = (ms * 1e-3 * 2.1e9) / (16.0 * iters * ninstr);
        printf("%-28s pattern %d: %.3f ms, per-wave ticks/iter %.1f, LDS-pipe cycles per wave-instruction (at 2.1 GHz) %.2f\n", name, sm, ms, mean / iters, cyc_per_instr_cu);
    }
    hipFree(d);
}

int main() {
    run<0>("6 x ds_add_u64", 6);
    run<1>("18 x ds_add_u32", 18);
    run<3>("6 x ds_add_u32", 6);
    run<2>("6 x (read+write b64)", 12);
    return 0;
}
