// row_ror:n DPP control on gfx950: which lane does lane l read?  (the tile kernel's filter meets its rows through it)
//   hipcc --offload-arch=gfx950 -O3 dpp_row_ror.hip -o dpp_row_ror
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __device__ __forceinline__ float ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, true));
}
__global__ void k(float *out, const float *in) {
    const float v = static_cast<float>(threadIdx.x);
    out[threadIdx.x] = ror<1>(v);
    out[64 + threadIdx.x] = ror<4>(v);
    out[128 + threadIdx.x] = ror<5>(ror<4>(v)); // composition: 9
    // the folded form the filter uses: acc += ror(v) * c
    float acc = in[threadIdx.x]; // 1000
    const float c = in[64 + threadIdx.x], c2 = in[128 + threadIdx.x]; // 2, 0
    acc = __builtin_fmaf(ror<3>(v), c, acc);
    acc = __builtin_fmaf(ror<3>(acc), c2, acc);
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(c2)); // + 0
    out[192 + threadIdx.x] = acc;
}
int main() {
    float *d, *din, h[256], hin[192];
    for (int i = 0; i < 64; i++) { hin[i] = 1000.0f; hin[64 + i] = 2.0f; hin[128 + i] = 0.0f; }
    hipMalloc(&d, sizeof(h));
    hipMalloc(&din, sizeof(hin));
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, din);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"row_ror:1", "row_ror:4", "row_ror:5 of row_ror:4", "1000 + 2 * row_ror:3"};
    for (int t = 0; t < 4; t++) {
        printf("%s  lane l reads:", names[t]);
        for (int i = 0; i < 64; i++) printf(" %g", h[t * 64 + i]);
        printf("\n");
    }
    return 0;
}
