// Diagnostic: slot-boundary stamps of the PERSISTENT GEMM kernel around its tile boundaries (TT_GEMM_ABLATE=8 build path):
// where do the cycles of a K = 1024 tile go when the K stream never drains?  Random operands.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
extern "C" int tt_gemm_debug_stamps(const void*, const void*, const float*, void*, int, int, int, void*, void*);
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill(uint16_t* p, size_t n, uint64_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 2654435761ULL + seed; x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
        float u = ((x & 0xFFFF) + ((x >> 16) & 0xFFFF)) * (1.0f / 65536.0f) - 1.0f;
        uint32_t b = __float_as_uint(u * 0.5f); b += 0x7FFF + ((b >> 16) & 1); p[i] = (uint16_t)(b >> 16);
    }
}
int main(int argc, char** argv) {
    int M = 473600, N = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 1024;
    uint16_t *a, *w, *c; float* bias; unsigned long long* st;
    CK(hipMalloc(&a, (size_t)M * K * 2)); CK(hipMalloc(&w, (size_t)N * K * 2)); CK(hipMalloc(&c, (size_t)M * N * 2));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&st, 2048 * 8));
    fill<<<2048, 256>>>(a, (size_t)M * K, 1); fill<<<2048, 256>>>(w, (size_t)N * K, 2); CK(hipMemset(bias, 0, N * 4));
    setenv("TT_GEMM_ABLATE", "8", 1);
    for (int i = 0; i < 3; ++i) { CK(hipMemset(st, 0, 2048 * 8)); tt_gemm_debug_stamps(a, w, bias, c, M, N, K, st, nullptr); CK(hipDeviceSynchronize()); }
    unsigned long long h[2048];
    CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
    const int nk = K / 64;
    printf("persistent bias GEMM M=%d N=%d K=%d, workgroup 0; cycles\n", M, N, K);
    for (int wsel = 0; wsel < 2; ++wsel) {
        printf("wave %d:\n", wsel * 4);
        for (int tile = 2; tile < 7; ++tile) {
            const unsigned long long* b = h + wsel * 1024 + tile * 128;
            const unsigned long long* prev = h + wsel * 1024 + (tile - 1) * 128;
            if (!b[0] || !prev[101]) continue;
            printf("  tile %d: prev epilogue %5lld | epilogue-end -> La(0) start %5lld | main loop %6lld | epilogue %5lld | whole %6lld\n",
                   tile, (long long)(prev[101] - prev[100]), (long long)(b[0] - prev[101]), (long long)(b[100] - b[0]), (long long)(b[101] - b[100]),
                   (long long)(b[101] - prev[101]));
            printf("     K-tile: ");
            for (int t = 0; t < nk; ++t) {
                const long long whole = (long long)((t + 1 < nk ? b[(t + 1) * 4] : b[100]) - b[t * 4]);
                printf("%d:%lld(La %lld Ca %lld Lb %lld Cb %lld) ", t, whole, (long long)(b[t * 4 + 1] - b[t * 4]), (long long)(b[t * 4 + 2] - b[t * 4 + 1]),
                       (long long)(b[t * 4 + 3] - b[t * 4 + 2]), (long long)((t + 1 < nk ? b[(t + 1) * 4] : b[100]) - b[t * 4 + 3]));
                if (t == 3) printf("\n             ");
                if (t >= 5 && t < nk - 3) { t = nk - 4; printf("... "); }
            }
            printf("\n");
        }
    }
    return 0;
}
