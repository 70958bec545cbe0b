// Experiment: does cutting the token rows into chunks whose FFN intermediate (chunk x 4096 bf16) stays in the 256 MiB
// Infinity Cache pay?  FFN-up(+GELU) then FFN-down(+residual) per chunk, the intermediate buffer REUSED by every chunk,
// against the whole batch at once (intermediate = M x 4096 x 2 B = 3.9 GB at the bench's M, written to and re-read from HBM).
//   ./ffn_chunk_bench [M] [iters]
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
        float u = ((h & 0xFFFF) + (h >> 16)) * (1.0f / 65536.0f) - 1.0f;
        uint32_t b = __float_as_uint(u * scale);
        b += 0x7FFF + ((b >> 16) & 1);
        p[i] = (uint16_t)(b >> 16);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 473600;
    const int iters = argc > 2 ? atoi(argv[2]) : 5;
    const int H = 1024, F = 4096;
    uint16_t *x, *w1, *w2, *hbuf, *y, *qkv, *wq, *wo, *ctx;
    float* bias;
    CK(hipMalloc(&x, (size_t)M * H * 2)); CK(hipMalloc(&y, (size_t)M * H * 2)); CK(hipMalloc(&hbuf, (size_t)M * F * 2));
    CK(hipMalloc(&w1, (size_t)F * H * 2)); CK(hipMalloc(&w2, (size_t)H * F * 2)); CK(hipMalloc(&bias, F * 4));
    CK(hipMalloc(&qkv, (size_t)M * 2 * H * 2)); CK(hipMalloc(&wq, (size_t)2 * H * H * 2)); CK(hipMalloc(&wo, (size_t)H * H * 2));
    CK(hipMalloc(&ctx, (size_t)M * H * 2));
    fill_bf16<<<2048, 256>>>(x, (size_t)M * H, 1, 1.0f);
    fill_bf16<<<2048, 256>>>(w1, (size_t)F * H, 2, 0.05f);
    fill_bf16<<<2048, 256>>>(w2, (size_t)H * F, 3, 0.05f);
    fill_bf16<<<2048, 256>>>(wq, (size_t)2 * H * H, 4, 0.05f);
    fill_bf16<<<2048, 256>>>(wo, (size_t)H * H, 5, 0.05f);
    fill_bf16<<<2048, 256>>>(ctx, (size_t)M * H, 6, 1.0f);
    CK(hipMemset(bias, 0, F * 4));
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("ffn_chunk_bench: M=%d iters=%d  (FFN-up+GELU then FFN-down+residual; intermediate reused per chunk)\n", M, iters);
    const int chunks[] = {0, 131072, 65536, 32768, 16384, 8192};
    for (int mc : chunks) {
        const int step = mc ? mc : M;
        auto run = [&]() {
            for (int lo = 0; lo < M; lo += step) {
                const int m = (M - lo) < step ? (M - lo) : step;
                uint16_t* hb = mc ? hbuf : hbuf + (size_t)lo * F;    // chunked: every chunk writes the same rows
                int rc = tt_gemm_bf16(x + (size_t)lo * H, w1, bias, nullptr, hb, m, F, H, 1, st);
                if (!rc) rc = tt_gemm_bf16(hb, w2, bias, x + (size_t)lo * H, y + (size_t)lo * H, m, H, F, 2, st);
                if (rc) { fprintf(stderr, "rc=%d %s\n", rc, tt_last_error()); exit(1); }
            }
        };
        run(); run();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        const double fl = 2.0 * M * (double)F * H * 2;
        printf("chunk %7d rows (%4d launches, intermediate %6.1f MiB): %.3f ms per FFN  %.1f TF/s\n", step, 2 * ((M + step - 1) / step),
               (double)step * F * 2 / 1048576.0, ms, fl / (ms * 1e-3) / 1e12);
    }
    // the attention-side pair: Q,K projection (N = 2048) written, then the output projection reading another M x 1024 buffer
    for (int mc : chunks) {
        const int step = mc ? mc : M;
        auto run = [&]() {
            for (int lo = 0; lo < M; lo += step) {
                const int m = (M - lo) < step ? (M - lo) : step;
                uint16_t* qb = mc ? qkv : qkv + (size_t)lo * 2 * H;
                int rc = tt_gemm_bf16(x + (size_t)lo * H, wq, bias, nullptr, qb, m, 2 * H, H, 0, st);
                if (!rc) rc = tt_gemm_bf16(ctx + (size_t)lo * H, wo, bias, x + (size_t)lo * H, y + (size_t)lo * H, m, H, H, 2, st);
                if (rc) { fprintf(stderr, "rc=%d %s\n", rc, tt_last_error()); exit(1); }
            }
        };
        run(); run();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        const double fl = 2.0 * M * (double)H * H * 3;
        printf("QK + o-proj, chunk %7d rows: %.3f ms  %.1f TF/s\n", step, ms, fl / (ms * 1e-3) / 1e12);
    }
    return 0;
}
