// v_mfma_f32_4x4x1_16b_f32 on gfx950: sixteen independent 4 x 4 outer products per instruction.  Which lane supplies what,
// and where do the results land?   hipcc --offload-arch=gfx950 -O3 mfma_4x4_layout.hip -o mfma_4x4_layout
//   claim checked:  lane l = 4 b + i supplies A[b][i] and B[b][i];  accumulator v of lane l = 4 b + j is D[b][v][j] = A[b][v] * B[b][j]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(float *D) {
    const int l = threadIdx.x;
    const float a = 1.0f + l;          // distinct per lane
    const float b = 1000.0f + 3.0f * l;
    floatx4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) D[l * 4 + v] = acc[v];
}
int main() {
    float h[256], *d;
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        for (int v = 0; v < 4; v++) {
            const int b = l / 4, j = l % 4;
            const float expect = (1.0f + (4 * b + v)) * (1000.0f + 3.0f * (4 * b + j));
            if (h[l * 4 + v] != expect) bad++;
        }
    }
    printf("mismatches against the claim: %d of 256\n", bad);
    if (bad) {
        for (int l = 0; l < 8; l++) printf("lane %d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
