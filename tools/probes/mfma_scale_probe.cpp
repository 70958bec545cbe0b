// Probe: block scales of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands), as the f16c GEMM stream uses them.
//   D[i][j] = sum_g 2^(sa[i][g] - 127) 2^(sb[j][g] - 127) sum_{k in block g} A[i][k] B[j][k]     block g = k in [32 g, 32 g + 32)
// Lane (row = l & 15, g = l >> 4) supplies bytes [16 g, 16 g + 16) and [64 + 16 g, 64 + 16 g + 16) of its row for both operands
// (the instruction's K order: registers 0-3 / 4-7) and, in ONE scale VGPR per operand, four candidate E8M0 bytes; the opsel
// immediate picks the byte.  Checks, for opsel_a, opsel_b in 0..3, that byte `opsel` of lane (row, g)'s scale register is the
// scale of block g = k in [32 g, 32 g + 32) of that row -- data of OTHER lanes (tools/probes/mfma_scale_map2.cpp found the
// pairing: scale lane group = (data lane group >> 1) + 2 (register >> 2)).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OA, int OB>
__global__ void probe(const uint8_t* a, const uint8_t* b, const uint32_t* sa, const uint32_t* sb, float* d) {
    const int l = threadIdx.x, row = l & 15, g = l >> 4;
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i a0 = *reinterpret_cast<const v4i*>(a + row * 128 + g * 16), a1 = *reinterpret_cast<const v4i*>(a + row * 128 + 64 + g * 16);
    const v4i b0 = *reinterpret_cast<const v4i*>(b + row * 128 + g * 16), b1 = *reinterpret_cast<const v4i*>(b + row * 128 + 64 + g * 16);
    const v8i fa = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const v8i fb = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, c, 0, 0, OA, (int)sa[l], OB, (int)sb[l]);
    for (int r = 0; r < 4; ++r) d[(4 * g + r) * 16 + row] = c[r];      // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
}

static uint8_t enc(int v) {
    static const uint8_t t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};  // 0..8
    uint8_t s = v < 0 ? 0x80 : 0;
    return s | t[abs(v)];
}

int main() {
    int A[16][128], B[16][128];
    uint8_t ha[16 * 128], hb[16 * 128];
    uint32_t hsa[64], hsb[64];
    srand(3);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) {
        A[i][k] = rand() % 9 - 4; B[i][k] = rand() % 9 - 4;
        ha[i * 128 + k] = enc(A[i][k]); hb[i * 128 + k] = enc(B[i][k]);
    }
    for (int l = 0; l < 64; ++l) {          // four different exponents per lane and operand
        hsa[l] = 0; hsb[l] = 0;
        for (int by = 0; by < 4; ++by) {
            hsa[l] |= (uint32_t)(120 + (l * 7 + by * 3) % 13) << (8 * by);
            hsb[l] |= (uint32_t)(122 + (l * 5 + by * 11) % 9) << (8 * by);
        }
    }
    uint8_t *da, *db; uint32_t *dsa, *dsb; float* dd; float hd[256];
    CK(hipMalloc(&da, sizeof ha)); CK(hipMalloc(&db, sizeof hb)); CK(hipMalloc(&dd, sizeof hd));
    CK(hipMalloc(&dsa, sizeof hsa)); CK(hipMalloc(&dsb, sizeof hsb));
    CK(hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice));
    int total_bad = 0;
    for (int oa = 0; oa < 4; ++oa) for (int ob = 0; ob < 4; ++ob) {
#define LAUNCH(OA, OB) if (oa == OA && ob == OB) probe<OA, OB><<<1, 64>>>(da, db, dsa, dsb, dd);
        LAUNCH(0, 0) LAUNCH(0, 1) LAUNCH(0, 2) LAUNCH(0, 3) LAUNCH(1, 0) LAUNCH(1, 1) LAUNCH(1, 2) LAUNCH(1, 3)
        LAUNCH(2, 0) LAUNCH(2, 1) LAUNCH(2, 2) LAUNCH(2, 3) LAUNCH(3, 0) LAUNCH(3, 1) LAUNCH(3, 2) LAUNCH(3, 3)
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int g = 0; g < 4; ++g) {
                int s = 0; for (int k = 32 * g; k < 32 * g + 32; ++k) s += A[i][k] * B[j][k];
                const int ea = (int)((hsa[g * 16 + i] >> (8 * oa)) & 255) - 127, eb = (int)((hsb[g * 16 + j] >> (8 * ob)) & 255) - 127;
                ref += ldexp((double)s, ea + eb);
            }
            if (fabs(hd[i * 16 + j] - ref) > 1e-6 * fabs(ref) + 1e-12) ++bad;
        }
        printf("opsel_a=%d opsel_b=%d: %d mismatches of 256 (D[0][0] = %g)\n", oa, ob, bad, hd[0]);
        total_bad += bad;
    }
    printf(total_bad ? "scale probe: MISMATCH -- the byte-select / lane mapping assumed by the f16c stream is wrong\n"
                     : "scale probe: ok -- byte `opsel` of lane (row, g)'s scale register scales block g (k in [32 g, 32 g + 32)) of that row, for both operands\n");
    return total_bad ? 1 : 0;
}
