// Discovery probe 2: which SCALE lane group multiplies which DATA of v_mfma_scale_f32_16x16x128_f8f6f4?
// A = 1.0 only in ONE (lane group g, register pair p) region of every row (8 bytes: registers 2p, 2p+1 of the lanes of group g),
// 0 elsewhere; B = 1.0 everywhere; scales 1.0 except A's lane group gs = 2.0.  D[i][j] = 8, or 16 when the scale of lane group
// gs is the one applied to that region.  Prints, for every (g, p), the gs that doubles it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void probe(int g_data, int p_data, int g_scale, int on_b, float* d) {
    const int l = threadIdx.x, row = l & 15, g = l >> 4;
    v8i ones, reg;
    for (int i = 0; i < 8; ++i) { ones[i] = 0x38383838; reg[i] = (g == g_data && (i >> 1) == p_data) ? 0x38383838 : 0; }
    const int s2 = (g == g_scale) ? 0x80808080 : 0x7F7F7F7F, s1 = 0x7F7F7F7F;
    v4f c = {0.f, 0.f, 0.f, 0.f};
    if (on_b) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, reg, c, 0, 0, 0, s1, 0, s2);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(reg, ones, c, 0, 0, 0, s2, 0, s1);
    for (int r = 0; r < 4; ++r) d[(4 * g + r) * 16 + row] = c[r];
}

int main() {
    float* dd; float hd[256];
    CK(hipMalloc(&dd, sizeof hd));
    for (int on_b = 0; on_b < 2; ++on_b) {
        printf("== region on operand %c: rows = data lane group g, columns = register pair p; entry = scale lane group(s) that double it [D value unscaled]\n", on_b ? 'B' : 'A');
        for (int g = 0; g < 4; ++g) {
            for (int p = 0; p < 4; ++p) {
                char buf[64]; int n = 0; float base = -1;
                for (int gs = 0; gs < 5; ++gs) {
                    probe<<<1, 64>>>(g, p, gs < 4 ? gs : -1, on_b, dd);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost));
                    if (gs == 4) base = hd[0];
                    else if (hd[0] != 8.f) n += snprintf(buf + n, sizeof buf - n, "%d(x%g) ", gs, hd[0] / 8.f);
                }
                buf[n] = 0;
                printf("  g=%d p=%d: %s[%g]   ", g, p, n ? buf : "none ", base);
            }
            printf("\n");
        }
    }
    return 0;
}
