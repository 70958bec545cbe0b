// Probe: v_mfma_scale_f32_16x16x128_f8f6f4 with BOTH operands laid out "lane (row, g4) holds bytes [32 g4, 32 g4 + 32) of its row's 128-byte
// K-tile" (shadow.hip) -- is D[4 (l >> 4) + r][l & 15] = sum_k a[row 4(l>>4)+r][k] b[row l&15][k]?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t* A, const uint8_t* B, float* D) {
    const int lane = threadIdx.x, j = lane & 15, g4 = lane >> 4;
    const uint4* pa = reinterpret_cast<const uint4*>(A + j * 128 + g4 * 32);
    const uint4* pb = reinterpret_cast<const uint4*>(B + j * 128 + g4 * 32);
    const uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
    const v8i av = v8i{(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    const v8i bv = v8i{(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    f4 c = f4{0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 4; ++r) D[(4 * g4 + r) * 16 + j] = c[r];
}
static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}
int main() {
    uint8_t hA[16 * 128], hB[16 * 128];
    srand(3);
    for (int i = 0; i < 16 * 128; ++i) { hA[i] = (rand() % 120) | ((rand() & 1) << 7); hB[i] = (rand() % 120) | ((rand() & 1) << 7); }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    float hD[256]; hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double worst_ij = 0, worst_ji = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int kk = 0; kk < 128; ++kk) s += (double)e4m3(hA[i * 128 + kk]) * e4m3(hB[j * 128 + kk]);
            worst_ij = fmax(worst_ij, fabs(hD[i * 16 + j] - s) / (fabs(s) + 1));
            worst_ji = fmax(worst_ji, fabs(hD[j * 16 + i] - s) / (fabs(s) + 1));
        }
    printf("D[a-row][b-row] vs reference: worst relative deviation %.3e ; transposed reading %.3e\n", worst_ij, worst_ji);
    return 0;
}
