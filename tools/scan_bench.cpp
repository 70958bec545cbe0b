// Standalone timing harness for the scan path of libtt_hip.so (no torch).
//   ./scan_bench [n_rows] [dim] [n_queries] [k] [iters]
// Fills a synthetic unit-norm-ish bf16 corpus on the device, runs tt_scan_topk in
// both corpus-load modes and prints whole-call GB/s (algorithmic bytes N*D*2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../include/tt_hip.h"

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ inline uint32_t hash32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}

__global__ void fill_bf16(uint16_t* p, size_t n, uint64_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        // sum of 4 uniforms ~ gaussian-ish, zero mean
        uint32_t h = hash32(i * 2654435761ULL + seed);
        uint32_t h2 = hash32(i * 40503ULL + seed * 7 + 13);
        float u = ((h & 0xFFFF) + (h >> 16) + (h2 & 0xFFFF) + (h2 >> 16)) * (1.0f / 65536.0f) - 2.0f;
        float v = u * scale;
        uint32_t b = __float_as_uint(v);
        b += 0x7FFF + ((b >> 16) & 1);
        p[i] = (uint16_t)(b >> 16);
    }
}

int main(int argc, char** argv) {
    int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    int d = argc > 2 ? atoi(argv[2]) : 1024;
    int q = argc > 3 ? atoi(argv[3]) : 64;
    int k = argc > 4 ? atoi(argv[4]) : 50;
    int iters = argc > 5 ? atoi(argv[5]) : 20;
    uint16_t *corpus, *queries;
    CK(hipMalloc(&corpus, (size_t)n * d * 2));
    CK(hipMalloc(&queries, (size_t)q * d * 2));
    fill_bf16<<<4096, 256>>>(corpus, (size_t)n * d, 1234, 0.054f);
    fill_bf16<<<64, 256>>>(queries, (size_t)q * d, 4321, 0.054f);
    CK(hipDeviceSynchronize());
    size_t ws_bytes = tt_scan_workspace_bytes(n, d, q, k);
    void* ws;
    CK(hipMalloc(&ws, ws_bytes));
    float* out_s;
    int32_t* out_i;
    int32_t* flag;
    CK(hipMalloc(&out_s, (size_t)q * k * 4));
    CK(hipMalloc(&out_i, (size_t)q * k * 4));
    CK(hipMalloc(&flag, 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("scan_bench: N=%lld D=%d Q=%d K=%d iters=%d ws=%.1f MB CUs=%d\n", (long long)n, d, q, k, iters,
           ws_bytes / 1e6, tt_device_cu_count());
    std::vector<int32_t> ref_idx;
    const char* modes_env = getenv("SCAN_BENCH_MODES");
    std::vector<int> modes;
    if (modes_env) {
        for (const char* c = modes_env; *c;) { modes.push_back(atoi(c)); while (*c && *c != ',') ++c; if (*c == ',') ++c; }
    } else { modes = {1, 0}; }
    for (int mode : modes) {
        char mbuf[16]; snprintf(mbuf, sizeof mbuf, "%d", mode);
        setenv("TT_SCAN_MODE", mbuf, 1);
        for (int w = 0; w < 3; ++w) {
            int rc = tt_scan_topk(corpus, n, d, queries, q, k, 0, out_s, out_i, ws, ws_bytes, flag, st);
            if (rc) { fprintf(stderr, "tt_scan_topk rc=%d: %s\n", rc, tt_last_error()); return 1; }
        }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < iters; ++it)
            tt_scan_topk(corpus, n, d, queries, q, k, 0, out_s, out_i, ws, ws_bytes, flag, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        int32_t hflag = 0;
        CK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
        std::vector<int32_t> idx((size_t)q * k);
        std::vector<float> sc((size_t)q * k);
        CK(hipMemcpy(idx.data(), out_i, idx.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sc.data(), out_s, sc.size() * 4, hipMemcpyDeviceToHost));
        double gb = (double)n * d * 2 / 1e9;
        printf("mode %d: %.3f ms/call  %.1f GB/s (whole call)  %.0f q/s  overflow=%d  top1[q0]=(%d, %.5f) top1[q%d]=(%d, %.5f)\n",
               mode, ms, gb / (ms * 1e-3), q / (ms * 1e-3), hflag, idx[0], sc[0], q - 1, idx[(size_t)(q - 1) * k],
               sc[(size_t)(q - 1) * k]);
        if (ref_idx.empty()) ref_idx = idx;
        else {
            size_t diff = 0;
            for (size_t i = 0; i < idx.size(); ++i) diff += idx[i] != ref_idx[i];
            printf("mode %d vs first mode index mismatches: %zu / %zu\n", mode, diff, idx.size());
        }
    }
    return 0;
}
