// v_mfma_f32_16x16x4_f32 on gfx950: operand and result layout, and the arithmetic error of a distance matrix computed as
// |r|^2 - 2 r.c in f32 (the tile kernel's MFMA filter).   hipcc --offload-arch=gfx950 -O3 mfma_layout.hip -o mfma_layout
//   claim checked here:  A: lane l supplies A[row = l % 16][k = l / 16];  B: lane l supplies B[k = l / 16][col = l % 16];
//                        D: accumulator v of lane l is D[row = 4 * (l / 16) + v][col = l % 16]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, float *D) { // A[16][4], B[4][16] row-major, D[16][16]
    const int l = threadIdx.x;
    const float a = A[(l % 16) * 4 + l / 16];
    const float b = B[(l / 16) * 16 + l % 16];
    floatx4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) {
        D[(4 * (l / 16) + v) * 16 + l % 16] = acc[v];
    }
}
int main() {
    float hA[64], hB[64], hD[256], *dA, *dB, *dD;
    // distance-matrix inputs: rows r (x, y, z, |r|^2), columns (-2x, -2y, -2z, 1)
    double r[16][3], c[16][3];
    srand(1);
    for (int i = 0; i < 16; i++) {
        for (int d = 0; d < 3; d++) {
            r[i][d] = 0.9 * rand() / RAND_MAX;        // row block extent
            c[i][d] = 4.0 * rand() / RAND_MAX - 1.0;  // columns up to a few nm from the origin
        }
        const float x = (float)r[i][0], y = (float)r[i][1], z = (float)r[i][2];
        hA[i * 4 + 0] = x; hA[i * 4 + 1] = y; hA[i * 4 + 2] = z; hA[i * 4 + 3] = x * x + y * y + z * z;
        hB[0 * 16 + i] = -2.0f * (float)c[i][0]; hB[1 * 16 + i] = -2.0f * (float)c[i][1]; hB[2 * 16 + i] = -2.0f * (float)c[i][2]; hB[3 * 16 + i] = 1.0f;
    }
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    double worst = 0, worst_layout = 0;
    for (int i = 0; i < 16; i++) {
        for (int j = 0; j < 16; j++) {
            double ref = 0; // what the layout claim predicts, in double from the f32 operands
            for (int kk = 0; kk < 4; kk++) ref += (double)hA[i * 4 + kk] * (double)hB[kk * 16 + j];
            worst_layout = fmax(worst_layout, fabs(ref - hD[i * 16 + j]));
            double d2 = 0, nc = 0;
            for (int d = 0; d < 3; d++) { d2 += (r[i][d] - c[j][d]) * (r[i][d] - c[j][d]); nc += (double)(float)c[j][d] * (double)(float)c[j][d]; }
            worst = fmax(worst, fabs(hD[i * 16 + j] + nc - d2));
        }
    }
    printf("max |D - A.B (layout claim, double)| = %.3e   (layout wrong if this is O(1))\n", worst_layout);
    printf("max | (D + |c|^2) - true d^2 | = %.3e for |c| up to %.1f nm\n", worst, sqrt(27.0));
    return 0;
}
