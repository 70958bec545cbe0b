// Standalone timing of tt_attention_varlen on the rerank shape (800 sequences x 292 tokens, 16 heads x 64).
//   ./att_bench [n_seq] [len] [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

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
    int n_seq = argc > 1 ? atoi(argv[1]) : 800;
    int len = argc > 2 ? atoi(argv[2]) : 292;
    int iters = argc > 3 ? atoi(argv[3]) : 20;
    const int heads = 16, dh = 64, H = heads * dh;
    const int stride = (len + 7) / 8 * 8;
    size_t T = ((size_t)n_seq * stride + 255) / 256 * 256;
    uint16_t *qk, *vt, *out; int32_t *ss, *sl;
    CK(hipMalloc(&qk, T * 2 * H * 2)); CK(hipMalloc(&vt, T * H * 2)); CK(hipMalloc(&out, T * H * 2));
    CK(hipMalloc(&ss, n_seq * 4)); CK(hipMalloc(&sl, n_seq * 4));
    std::vector<int32_t> hs(n_seq), hl(n_seq);
    for (int i = 0; i < n_seq; ++i) { hs[i] = i * stride; hl[i] = len; }
    CK(hipMemcpy(ss, hs.data(), n_seq * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sl, hl.data(), n_seq * 4, hipMemcpyHostToDevice));
    fill_bf16<<<2048, 256>>>(qk, T * 2 * H, 1, 1.0f);
    fill_bf16<<<2048, 256>>>(vt, T * H, 2, 1.0f);
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) {
        int rc = tt_attention_varlen(qk, 2 * H, 0, H, vt, 8 * H, out, H, ss, sl, n_seq, heads, dh, len, st);
        if (rc) { fprintf(stderr, "rc=%d %s\n", rc, tt_last_error()); return 1; }
    }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) tt_attention_varlen(qk, 2 * H, 0, H, vt, 8 * H, out, H, ss, sl, n_seq, heads, dh, len, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    double fl = 4.0 * n_seq * heads * (double)len * len * dh;
    printf("attention n_seq=%d len=%d heads=%d dh=%d: %.3f ms  %.1f TF/s (useful flops)\n", n_seq, len, heads, dh, ms, fl / (ms * 1e-3) / 1e12);
    return 0;
}
