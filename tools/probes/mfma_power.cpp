// Probe: what the chip SUSTAINS (seconds, under its power cap) on a stream of nothing but MFMAs, by instruction shape:
//   0: v_mfma_f32_16x16x32_bf16 (16 accumulator tiles of 4 registers per wave, the GEMM's C slot)
//   1: v_mfma_f32_32x32x16_bf16 (4 accumulator tiles of 16 registers per wave: half the instructions and operand reads per flop)
//   2: v_mfma_f32_16x16x32_bf16 with ALL-ZERO operands (data-dependent switching power)
// usage: mfma_power MODE SECONDS      -> TF/s over the whole window and over its last third
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int seed) {
    b8 a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = MODE == 2 ? (__bf16)0.f : (__bf16)(0.37f * (((threadIdx.x * 7 + i * 3 + j * 5 + seed) % 13) - 6));
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = MODE == 2 ? (__bf16)0.f : (__bf16)(0.21f * (((threadIdx.x * 3 + i * 11 + j * 7 + seed) % 11) - 5));
    float sink = 0.f;
    if (MODE == 1) {
        v16f acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)        // 16 instructions of 32x32x16 = the flops of 32 of 16x16x32
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[t & 1], a[(t + r) & 3], acc[t], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) sink += acc[i][j];
    } else {
        v4f acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[t & 1], a[(t >> 1) & 3], acc[t], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) sink += acc[i][j];
    }
    if (sink == 12345.678f) out[0] = sink;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    float* d; CK(hipMalloc(&d, 4));
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = 40000;      // ~20 ms per launch
    const double flops = (double)cus * 8 * iters * 32.0 * 2 * 16 * 16 * 32;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    double total = 0;
    for (int rep = 0; total < secs * 1e3; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) k<0><<<cus, 512>>>(d, iters, rep);
        else if (mode == 1) k<1><<<cus, 512>>>(d, iters, rep);
        else k<2><<<cus, 512>>>(d, iters, rep);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t); total += t;
    }
    double last = 0; size_t n3 = ms.size() / 3 ? ms.size() / 3 : 1;
    for (size_t i = ms.size() - n3; i < ms.size(); ++i) last += ms[i];
    printf("%s: %zu launches, first %.0f TF/s, whole window %.0f TF/s, last third %.0f TF/s\n",
           mode == 0 ? "16x16x32 bf16" : mode == 1 ? "32x32x16 bf16" : "16x16x32 bf16, zero operands",
           ms.size(), flops / (ms[0] * 1e-3) / 1e12, flops * ms.size() / (total * 1e-3) / 1e12, flops * n3 / (last * 1e-3) / 1e12);
    return 0;
}
