// wave_rol:1 / wave_ror:1 DPP controls on gfx950: which lane does lane l read?  (the rotation tile kernel passes the
// row-force accumulators from lane to lane with them)   hipcc --offload-arch=gfx950 -O3 dpp_rotate.hip -o dpp_rotate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned int *out) {
    const unsigned int v = threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(777u, v, 0x134, 0xf, 0xf, false);        // wave_rol:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(777u, v, 0x13C, 0xf, 0xf, false);   // wave_ror:1
    // under a partial exec mask: what do lanes read whose source lane is disabled?
    unsigned int w = 555u;
    if (threadIdx.x & 1) {
        w = __builtin_amdgcn_update_dpp(777u, v, 0x134, 0xf, 0xf, false);
    }
    out[128 + threadIdx.x] = w;
}
int main() {
    unsigned int *d, h[192];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_rol:1  lane l reads lane:");
    for (int i = 0; i < 64; i++) printf(" %u", h[i]);
    printf("\nwave_ror:1  lane l reads lane:");
    for (int i = 0; i < 64; i++) printf(" %u", h[64 + i]);
    printf("\nwave_rol:1 with only odd lanes enabled:");
    for (int i = 0; i < 64; i++) printf(" %u", h[128 + i]);
    printf("\n");
    return 0;
}
