// Issue cost of packed f32 VALU instructions on gfx950: cycles per wave-instruction of v_fma_f32, v_pk_fma_f32, v_pk_mul_f32,
// v_pk_add_f32 and v_fma_f64 with 1, 2 and 4 waves per SIMD (8 independent accumulators per lane, s_memtime around 4096
// instructions; every wave reports its own start and end).   hipcc --offload-arch=gfx950 -O3 pk_f32_rate.hip -o pk_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void k(long long *out, float seed) {
    v2f a[8];
    double d[8];
    for (int i = 0; i < 8; i++) {
        a[i].x = seed + i + threadIdx.x;
        a[i].y = seed - i;
        d[i] = seed * i + threadIdx.x;
    }
    const v2f m = {1.0001f, 0.9999f}, c = {1e-7f, -1e-7f};
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 512; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(1.0001), "v"(1e-9));
            if (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y + static_cast<float>(d[i]);
    if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t0; out[2 * (threadIdx.x >> 6) + 1] = t1; } // per wave: start, end
    if (s == 12345.678f) out[0] = 0;
}
template <int KIND> void run(const char *name, long long *d) {
    for (int threads : {64, 256, 512, 768, 1024}) { // one workgroup on one CU: waves per SIMD = threads / 256 (min 1)
        k<KIND><<<1, threads>>>(d, 1.0f);
        long long h[32];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const int waves = threads / 64;
        long long first = h[0], last = h[1];
        double mean = 0;
        for (int w = 0; w < waves; w++) {
            first = h[2 * w] < first ? h[2 * w] : first;
            last = h[2 * w + 1] > last ? h[2 * w + 1] : last;
            mean += static_cast<double>(h[2 * w + 1] - h[2 * w]) / waves;
        }
        // 4096 instructions per wave; a workgroup's waves are dealt round-robin to the CU's four SIMDs
        printf("%-14s %2d waves on one CU: %6.2f ticks per instruction per wave (mean), %6.2f per instruction per SIMD (all waves, first start to last end)\n",
               name, waves, mean / 4096.0, static_cast<double>(last - first) / (4096.0 * ((waves + 3) / 4)));
    }
}
int main() {
    long long *d;
    hipMalloc(&d, 32 * sizeof(long long));
    run<0>("v_fma_f32", d);
    run<5>("v_mul_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<2>("v_pk_mul_f32", d);
    run<3>("v_pk_add_f32", d);
    run<4>("v_fma_f64", d);
    return 0;
}
