// Where does a global atomic of WORKGROUP scope execute on gfx950 -- in the issuing XCD's L2 (then adds from different XCDs to
// one address lose updates: eight non-coherent L2s) or at the memory side like agent scope (then the sum is exact)?  And what
// does each cost per wave instruction?    hipcc --offload-arch=gfx950 -O3 atomic_l2_scope.hip -o atomic_l2_scope
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SCOPE> __global__ void k(unsigned long long *p, int reps, long long *cycles) {
    const long long t0 = clock64();
    for (int i = 0; i < reps; i++) {
        // 64 lanes -> 64 consecutive u64 (8 lines), every wave of the grid the same 64 addresses shifted by its block's slot
        __hip_atomic_fetch_add(p + ((blockIdx.x & 63) * 64 + (threadIdx.x & 63)), 1ull, __ATOMIC_RELAXED, SCOPE);
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned long long *p, h[4096];
    long long *c, hc[256];
    hipMalloc(&p, sizeof(h));
    hipMalloc(&c, sizeof(hc));
    const int reps = 200;
    for (int scope = 0; scope < 2; scope++) {
        hipMemset(p, 0, sizeof(h));
        if (scope == 0) k<__HIP_MEMORY_SCOPE_AGENT><<<256, 64>>>(p, reps, c);
        else k<__HIP_MEMORY_SCOPE_WORKGROUP><<<256, 64>>>(p, reps, c);
        hipDeviceSynchronize();
        hipMemcpy(h, p, sizeof(h), hipMemcpyDeviceToHost);
        hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
        unsigned long long sum = 0;
        for (int i = 0; i < 4096; i++) sum += h[i];
        double mean = 0;
        for (int b = 0; b < 256; b++) mean += hc[b] / 256.0;
        printf("%s scope: sum %llu of %llu expected (%s); %.0f cycles per wave instruction\n", scope ? "workgroup" : "agent", sum,
               256ull * 64 * reps, sum == 256ull * 64 * reps ? "exact: executed coherently" : "LOST UPDATES: executed per L2", mean / reps);
    }
    return 0;
}
