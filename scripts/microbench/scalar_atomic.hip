// Scalar-memory atomics on gfx950 (s_atomic_add ... glc): do they work as per-XCD ticket counters, and does the returned value
// come back ahead of the wave's own outstanding VECTOR atomics (a returning vector atomic does not: it is queued behind them)?
//   hipcc --offload-arch=gfx950 -O3 scalar_atomic.hip -o scalar_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ unsigned int xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; } // HW_REG_XCC_ID
__global__ void k(unsigned int *counters /* 8 x 64 B apart */, unsigned int *tickets, unsigned int *xcc, unsigned long long *sink, long long *lat, int flood) {
    const int wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const unsigned int x = xcc_id();
    if (flood) { // this wave's own vector atomics in flight first (component-major flush look-alike)
        for (int i = 0; i < 4; i++) {
            atomicAdd(sink + ((wave * 4 + i) * 64 + (threadIdx.x & 63)) % (1 << 20), 1ull);
        }
    }
    unsigned int *p = counters + x * 16;
    unsigned int v = 1;
    const long long t0 = clock64();
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(p) : "memory");
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) {
        tickets[wave] = v;
        xcc[wave] = x;
        lat[wave] = t1 - t0;
    }
}
__global__ void kv(unsigned int *counters, unsigned int *tickets, unsigned long long *sink, long long *lat, int flood) {
    const int wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (flood) {
        for (int i = 0; i < 4; i++) {
            atomicAdd(sink + ((wave * 4 + i) * 64 + (threadIdx.x & 63)) % (1 << 20), 1ull);
        }
    }
    unsigned int v = 0;
    const long long t0 = clock64();
    if ((threadIdx.x & 63) == 0) {
        v = atomicAdd(counters, 1u); // returning vector atomic, device scope
    }
    v = __builtin_amdgcn_readfirstlane(v);
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) {
        tickets[wave] = v;
        lat[wave] = t1 - t0;
    }
}
// the realistic pattern: every wave requests a ticket now and then (one per ~20k cycles of other work, as a tile-kernel wave
// does once per item), with its own vector atomics in flight
__global__ void kstag(unsigned int *counters, unsigned long long *sink, long long *lat, float *dummy) {
    const int wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const unsigned int x = xcc_id();
    unsigned int *p = counters + x * 16;
    float acc = threadIdx.x;
    long long worst = 0, sum = 0;
    for (int it = 0; it < 8; it++) {
        const int spin = 3000 + ((wave * 37 + it * 101) % 2000); // ~15-25k cycles of dependent FMAs
        for (int i = 0; i < spin; i++) {
            acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
        }
        for (int i = 0; i < 4; i++) {
            atomicAdd(sink + ((wave * 4 + i) * 64 + (threadIdx.x & 63)) % (1 << 20), 1ull);
        }
        unsigned int v = 1;
        const long long t0 = clock64();
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(p) : "memory");
        const long long t1 = clock64();
        worst = t1 - t0 > worst ? t1 - t0 : worst;
        sum += t1 - t0;
    }
    if ((threadIdx.x & 63) == 0) {
        lat[wave] = sum / 8;
        lat[8192 + wave] = worst;
    }
    if (acc == 1.2345f) dummy[0] = acc;
}
int main() {
    const int G = 256, T = 1024, W = G * T / 64;
    unsigned int *c, *tk, *xc;
    unsigned long long *sink;
    long long *lat;
    hipMalloc(&c, 8 * 64);
    hipMalloc(&tk, W * 4);
    hipMalloc(&xc, W * 4);
    hipMalloc(&sink, (1 << 20) * 8);
    hipMalloc(&lat, 2 * 8192 * 8);
    float *dummy;
    hipMalloc(&dummy, 4);
    std::vector<unsigned int> htk(W), hxc(W), hc(128);
    std::vector<long long> hl(W);
    for (int flood = 0; flood < 2; flood++) {
        hipMemset(c, 0, 8 * 64);
        hipMemset(sink, 0, (1 << 20) * 8);
        k<<<G, T>>>(c, tk, xc, sink, lat, flood);
        hipDeviceSynchronize();
        hipMemcpy(htk.data(), tk, W * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hxc.data(), xc, W * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), c, 8 * 64, hipMemcpyDeviceToHost);
        hipMemcpy(hl.data(), lat, W * 8, hipMemcpyDeviceToHost);
        int ok = 1;
        printf("scalar atomics, own vector atomics in flight: %d\n", flood);
        for (int x = 0; x < 8; x++) {
            std::vector<unsigned int> t;
            for (int w = 0; w < W; w++) if (hxc[w] == (unsigned)x) t.push_back(htk[w]);
            std::sort(t.begin(), t.end());
            int perm = 1;
            for (size_t i = 0; i < t.size(); i++) perm = perm && t[i] == i;
            printf("  xcc %d: %zu waves, counter %u, tickets are 0..n-1: %s\n", x, t.size(), hc[x * 16], perm ? "yes" : "NO");
            ok = ok && perm && hc[x * 16] == t.size();
        }
        std::sort(hl.begin(), hl.end());
        printf("  %s; latency (cycles) median %lld, p90 %lld, max %lld\n", ok ? "OK" : "BROKEN", hl[W / 2], hl[W * 9 / 10], hl[W - 1]);
        // blockIdx -> xcc pattern
        printf("  xcc of blocks 0..15:");
        for (int b = 0; b < 16; b++) printf(" %u", hxc[b * (T / 64)]);
        printf("\n");
        hipMemset(c, 0, 8 * 64);
        kv<<<G, T>>>(c, tk, sink, lat, flood);
        hipDeviceSynchronize();
        hipMemcpy(hl.data(), lat, W * 8, hipMemcpyDeviceToHost);
        std::sort(hl.begin(), hl.end());
        printf("  returning VECTOR atomic on one counter: latency median %lld, p90 %lld, max %lld\n", hl[W / 2], hl[W * 9 / 10], hl[W - 1]);
    }
    {
        hipMemset(c, 0, 8 * 64);
        kstag<<<G, T>>>(c, sink, lat, dummy);
        hipDeviceSynchronize();
        std::vector<long long> m(W), wv(W);
        hipMemcpy(m.data(), lat, W * 8, hipMemcpyDeviceToHost);
        hipMemcpy(wv.data(), lat + 8192, W * 8, hipMemcpyDeviceToHost);
        std::sort(m.begin(), m.end());
        std::sort(wv.begin(), wv.end());
        printf("staggered requests (8 per wave, ~20k cycles apart, own vector atomics in flight): mean latency per wave median %lld, p90 %lld, max %lld; worst single median %lld max %lld\n",
               m[W / 2], m[W * 9 / 10], m[W - 1], wv[W / 2], wv[W - 1]);
    }
    return 0;
}
