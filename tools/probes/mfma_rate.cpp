// Probe: what the chip sustains on a stream of nothing but MFMAs, per instruction form (no operand traffic): 8 waves per CU,
// 16 independent accumulators per wave, as in the GEMM's C slots.  Prints TF/s-equivalent and cycles per instruction.
//   f16  : v_mfma_f32_16x16x32_f16                      (2 * 16*16*32 flops)
//   fp8s : v_mfma_scale_f32_16x16x128_f8f6f4, e4m3      (2 * 16*16*128 flops)
//   mix  : 32 f16 + 16 fp8s alternating in blocks, as the f16c K stream does per pair of tiles
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int seed) {
    v4f acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    h8 ha[4], hb[2];
    v8i fa[4], fb[2];
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 8; ++j) { ha[i][j] = (_Float16)(0.001f * ((threadIdx.x + i + j + seed) % 7)); fa[i][j] = 0x38303438 + seed + i * 131 + j * 17 + threadIdx.x; } }
    for (int i = 0; i < 2; ++i) { for (int j = 0; j < 8; ++j) { hb[i][j] = (_Float16)(0.002f * ((threadIdx.x + 3 * i + j + seed) % 5)); fb[i][j] = 0x30383430 + seed + i * 37 + j * 7 + threadIdx.x; } }
    const int sa = 0x7F7E7D7C + seed, sb = 0x7C7D7E7F + seed;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        acc[nt * 4 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb[nt], ha[mt], acc[nt * 4 + mt], 0, 0, 0);
                        acc[8 + nt * 4 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb[nt], ha[mt], acc[8 + nt * 4 + mt], 0, 0, 0);
                    }
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    acc[nt * 4 + mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[nt], fa[mt], acc[nt * 4 + mt], 0, 0, 0, sa, 0, sb);
                    acc[8 + nt * 4 + mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[nt], fa[mt], acc[8 + nt * 4 + mt], 0, 0, 1, sa, 1, sb);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float* d; CK(hipMalloc(&d, 4));
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) k<0><<<cus, 512>>>(d, iters, rep);
            if (mode == 1) k<1><<<cus, 512>>>(d, iters, rep);
            if (mode == 2) k<2><<<cus, 512>>>(d, iters, rep);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double n16 = (mode == 0 || mode == 2) ? 32.0 : 0.0, n8 = (mode == 1 || mode == 2) ? 16.0 : 0.0;
        const double flops = (double)cus * 8 * iters * (n16 * 2 * 16 * 16 * 32 + n8 * 2 * 16 * 16 * 128);
        // per SIMD: 2 waves, each n instructions per iteration
        printf("%s: %.2f ms, %.0f TF/s-equivalent, time per iteration and wave-pair %.1f ns (bf16-unit tiles: %.2f per iteration)\n",
               mode == 0 ? "f16  (32 x 16x16x32 per iteration)" : mode == 1 ? "fp8s (16 x 16x16x128 scaled per iteration)" : "mix  (32 f16 + 16 fp8s per iteration)",
               ms, flops / (ms * 1e-3) / 1e12, ms * 1e6 / iters, (n16 / 32.0 + n8 / 16.0));
    }
    return 0;
}
