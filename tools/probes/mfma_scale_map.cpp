// Discovery probe: WHICH lane / byte of the scale registers of v_mfma_scale_f32_16x16x128_f8f6f4 scales WHICH part of the
// product?  All data = 1.0 (every 32-element block of every row pair contributes 32), all scales 1.0 (E8M0 127) except ONE
// byte of ONE lane of one operand's scale register = 2.0; the rows / columns of D that move, and by how much (+32 = one
// block doubled), identify (row, block) of that (lane, byte) for each opsel value.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OA, int OB>
__global__ void probe(const uint32_t* sa, const uint32_t* sb, float* d) {
    const int l = threadIdx.x, row = l & 15, g = l >> 4;
    v8i ones;
    for (int i = 0; i < 8; ++i) ones[i] = 0x38383838;        // e4m3 1.0
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, ones, c, 0, 0, OA, (int)sa[l], OB, (int)sb[l]);
    for (int r = 0; r < 4; ++r) d[(4 * g + r) * 16 + row] = c[r];      // D[i = 4 g + r][j = lane & 15]
}

static void run(int oa, int ob, const uint32_t* dsa, const uint32_t* dsb, float* dd) {
#define L(OA, OB) if (oa == OA && ob == OB) probe<OA, OB><<<1, 64>>>(dsa, dsb, dd);
    L(0, 0) L(1, 0) L(2, 0) L(3, 0) L(0, 1) L(0, 2) L(0, 3)
#undef L
    CK(hipDeviceSynchronize());
}

int main() {
    uint32_t hs[64], base[64];
    uint32_t *dsa, *dsb; float* dd; float hd[256];
    CK(hipMalloc(&dsa, sizeof hs)); CK(hipMalloc(&dsb, sizeof hs)); CK(hipMalloc(&dd, sizeof hd));
    for (int l = 0; l < 64; ++l) base[l] = 0x7F7F7F7Fu;
    for (int operand = 0; operand < 2; ++operand)
        for (int o = 0; o < 4; ++o) {
            printf("== operand %c, opsel %d: (lane, byte) -> what moved\n", operand ? 'B' : 'A', o);
            for (int b = 0; b < 4; ++b)
                for (int lane = 0; lane < 64; ++lane) {
                    memcpy(hs, base, sizeof hs);
                    hs[lane] = (hs[lane] & ~(0xFFu << (8 * b))) | (0x80u << (8 * b));
                    CK(hipMemcpy(operand ? dsb : dsa, hs, sizeof hs, hipMemcpyHostToDevice));
                    CK(hipMemcpy(operand ? dsa : dsb, base, sizeof base, hipMemcpyHostToDevice));
                    run(operand ? 0 : o, operand ? o : 0, dsa, dsb, dd);
                    CK(hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost));
                    // which rows i (operand A) or columns j (operand B) moved, and by how much
                    int moved = -1, n_moved = 0; float delta = 0.f; bool uniform = true;
                    for (int x = 0; x < 16; ++x) {
                        const float v = operand ? hd[0 * 16 + x] : hd[x * 16 + 0];
                        if (v != 128.f) { ++n_moved; moved = x; delta = v - 128.f; }
                        for (int y = 0; y < 16; ++y) { const float w = operand ? hd[y * 16 + x] : hd[x * 16 + y]; if (w != v) uniform = false; }
                    }
                    if (n_moved) printf("  lane %2d byte %d -> %s %2d%s +%g%s\n", lane, b, operand ? "col" : "row", moved, n_moved > 1 ? " (and others)" : "",
                                        delta, uniform ? "" : " (non-uniform along the other axis)");
                }
        }
    return 0;
}
