// Probe: v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit scales.
// D[i][j] = sum_k A[i][k] * B[j][k]  (both operands K-contiguous rows, lane (row = l&15, g = l>>4) loads the 32
// bytes at k = 32 g .. 32 g + 31 of its row for BOTH operands).  Checks against an integer reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void probe(const uint8_t* a, const uint8_t* b, float* d, int swapped) {
    const int l = threadIdx.x, row = l & 15, g = l >> 4;
    v8i fa = *reinterpret_cast<const v8i*>(a + row * 128 + g * 32);
    v8i fb = *reinterpret_cast<const v8i*>(b + row * 128 + g * 32);
    v4f c = {0.f, 0.f, 0.f, 0.f};
    // cbsz (A format) = 0: fp8 e4m3, blgp (B format) = 0; scales: E8M0 127 = 1.0 in every byte
    if (swapped) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    // C/D layout (16x16): col = lane & 15, row = 4 * (lane >> 4) + reg
    for (int r = 0; r < 4; ++r) d[(4 * g + r) * 16 + row] = c[r];
}

// e4m3fn encodings of small integers
static uint8_t enc(int v) {
    static const uint8_t t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};  // 0..8
    uint8_t s = v < 0 ? 0x80 : 0;
    return s | t[abs(v)];
}
int main() {
    int A[16][128], B[16][128];
    uint8_t ha[16 * 128], hb[16 * 128];
    srand(1);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) {
        A[i][k] = rand() % 9 - 4; B[i][k] = rand() % 9 - 4;
        ha[i * 128 + k] = enc(A[i][k]); hb[i * 128 + k] = enc(B[i][k]);
    }
    uint8_t *da, *db; float* dd; float hd[256];
    CK(hipMalloc(&da, sizeof ha)); CK(hipMalloc(&db, sizeof hb)); CK(hipMalloc(&dd, sizeof hd));
    CK(hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice));
    for (int sw = 0; sw < 2; ++sw) {
        probe<<<1, 64>>>(da, db, dd, sw);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost));
        int bad = 0, badT = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            int ref = 0; for (int k = 0; k < 128; ++k) ref += A[i][k] * B[j][k];
            if (hd[i * 16 + j] != (float)ref) ++bad;       // D[i][j] = A_i . B_j
            if (hd[j * 16 + i] != (float)ref) ++badT;      // transposed
        }
        printf("swapped=%d: mismatches as D[i][j]=A_i.B_j: %d, as transposed: %d (of 256)\n", sw, bad, badT);
    }
    return 0;
}
