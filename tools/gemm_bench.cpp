// Standalone timing of tt_gemm_bf16 on the encoder-layer shapes (no torch).
//   ./gemm_bench [M] [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../include/tt_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint32_t hash32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}
__global__ void fill_bf16(uint16_t* p, size_t n, uint64_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = hash32(i * 2654435761ULL + seed);
        float u = ((h & 0xFFFF) + (h >> 16)) * (1.0f / 65536.0f) - 1.0f;   // [-1, 1) triangular
        uint32_t b = __float_as_uint(u * scale);
        b += 0x7FFF + ((b >> 16) & 1);
        p[i] = (uint16_t)(b >> 16);
    }
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 16384;
    int iters = argc > 2 ? atoi(argv[2]) : 20;
    struct Shape { int n, k, epi; const char* name; } shapes[] = {
        {3072, 1024, 0, "qkv(bias)"}, {1024, 1024, 2, "o-proj(+res)"}, {4096, 1024, 1, "ffn-up(gelu)"},
        {1024, 4096, 2, "ffn-down(+res)"}, {1024, 1024, 0, "o-proj shape, bias only"}, {4096, 1024, 0, "ffn-up shape, bias only"},
        {1024, 4096, 0, "ffn-down shape, bias only"}, {1152, 384, 0, "small qkv"}, {1536, 384, 1, "small ffn-up"}};
    uint16_t *a, *w, *c, *r;
    float* bias;
    size_t maxA = (size_t)M * 4096, maxW = (size_t)4096 * 4096, maxC = (size_t)M * 4096;
    CK(hipMalloc(&a, maxA * 2)); CK(hipMalloc(&w, maxW * 2)); CK(hipMalloc(&c, maxC * 2)); CK(hipMalloc(&r, maxC * 2));
    CK(hipMalloc(&bias, 4096 * 4));
    fill_bf16<<<2048, 256>>>(a, maxA, 1, 1.0f);
    fill_bf16<<<2048, 256>>>(w, maxW, 2, 0.05f);
    fill_bf16<<<2048, 256>>>(r, maxC, 3, 1.0f);
    CK(hipMemset(bias, 0, 4096 * 4));
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("gemm_bench: M=%d iters=%d (uniform random operands)\n", M, iters);
    for (auto& s : shapes) {
        for (int i = 0; i < 3; ++i) {
            int rc = tt_gemm_bf16(a, w, bias, s.epi == 2 ? r : nullptr, c, M, s.n, s.k, s.epi, st);
            if (rc) { fprintf(stderr, "rc=%d %s\n", rc, tt_last_error()); return 1; }
        }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) tt_gemm_bf16(a, w, bias, s.epi == 2 ? r : nullptr, c, M, s.n, s.k, s.epi, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        double fl = 2.0 * M * s.n * s.k;
        printf("%-16s M=%d N=%d K=%d  %.3f ms  %.1f TF/s\n", s.name, M, s.n, s.k, ms, fl / (ms * 1e-3) / 1e12);
    }
    // split-bf16 ("bf16x3", the reference precision): the same shapes over hi / lo planes, three MFMA products per product
    {
        uint16_t *ap, *wp, *cp; float *c32, *r32;
        size_t maxAp = (size_t)M * 2 * 4096, maxWp = (size_t)4096 * 2 * 4096;
        CK(hipMalloc(&ap, maxAp * 2)); CK(hipMalloc(&wp, maxWp * 2)); CK(hipMalloc(&cp, (size_t)M * 2 * 4096 * 2));
        CK(hipMalloc(&c32, (size_t)M * 1024 * 4)); CK(hipMalloc(&r32, (size_t)M * 1024 * 4));
        fill_bf16<<<2048, 256>>>(ap, maxAp, 11, 1.0f);
        fill_bf16<<<2048, 256>>>(wp, maxWp, 12, 0.05f);
        fill_bf16<<<2048, 256>>>((uint16_t*)r32, (size_t)M * 1024 * 2, 13, 1.0f);
        CK(hipDeviceSynchronize());
        struct SX { int n, k, epi; const char* name; } sx[] = {{2048, 1024, 0, "x3 q,k proj (planes)"}, {1024, 1024, 2, "x3 o-proj (+res fp32)"},
                                                               {4096, 1024, 1, "x3 ffn-up (erf gelu)"}, {1024, 4096, 2, "x3 ffn-down (+res)"},
                                                               {4096, 1024, 0, "x3 ffn-up shape, bias"}};
        for (auto& s : sx) {
            auto run = [&]() { return tt_gemm_x3(ap, wp, bias, s.epi == 2 ? r32 : nullptr, s.epi == 2 ? nullptr : cp, s.epi == 2 ? c32 : nullptr, M, s.n, s.k, s.epi, st); };
            for (int i = 0; i < 3; ++i) { int rc = run(); if (rc) { fprintf(stderr, "rc=%d %s\n", rc, tt_last_error()); return 1; } }
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) run();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
            double fl = 3.0 * 2.0 * M * s.n * s.k;
            printf("%-24s M=%d N=%d K=%d  %.3f ms  %.1f TF/s of bf16 MFMA work\n", s.name, M, s.n, s.k, ms, fl / (ms * 1e-3) / 1e12);
        }
        CK(hipFree(ap)); CK(hipFree(wp)); CK(hipFree(cp)); CK(hipFree(c32)); CK(hipFree(r32));
    }
    // fp8 (e4m3) forms of the two projections the fp8 mode moves to the fp8 matrix cores
    {
        uint8_t *a8, *w8; float *sa, *sw;
        CK(hipMalloc(&a8, (size_t)M * 1024)); CK(hipMalloc(&w8, (size_t)4096 * 1024));
        CK(hipMalloc(&sa, (size_t)M * 4)); CK(hipMalloc(&sw, 4096 * 4));
        fill_bf16<<<2048, 256>>>((uint16_t*)a8, (size_t)M * 512, 7, 1.0f);    // random bytes (most are finite e4m3 values)
        fill_bf16<<<2048, 256>>>((uint16_t*)w8, (size_t)4096 * 512, 8, 1.0f);
        CK(hipMemset(sa, 0, (size_t)M * 4)); CK(hipMemset(sw, 0, 4096 * 4));
        CK(hipDeviceSynchronize());
        struct S8 { int n, epi; const char* name; } s8[] = {{2048, 0, "fp8 qk(bias)"}, {3072, 0, "fp8 qkv-sized(bias)"}, {4096, 1, "fp8 ffn-up(gelu)"}};
        for (auto& s : s8) {
            for (int i = 0; i < 3; ++i) {
                int rc = tt_gemm_fp8(a8, sa, w8, sw, bias, c, M, s.n, 1024, s.epi, st);
                if (rc) { fprintf(stderr, "rc=%d %s\n", rc, tt_last_error()); return 1; }
            }
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) tt_gemm_fp8(a8, sa, w8, sw, bias, c, M, s.n, 1024, s.epi, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
            double fl = 2.0 * M * s.n * 1024;
            printf("%-22s M=%d N=%d K=1024  %.3f ms  %.1f TF/s\n", s.name, M, s.n, ms, fl / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
