// Diagnostic: per-phase s_memtime stamps of the v4 GEMM loop (TT_GEMM_ABLATE=6 build path).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
extern "C" int tt_gemm_debug_stamps(const void*, const void*, const float*, void*, int, int, int, void*, void*);
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    int M = 16384, N = 3072, K = 1024;
    uint16_t *a, *w, *c; float* bias; unsigned long long* st;
    CK(hipMalloc(&a, (size_t)M * K * 2)); CK(hipMalloc(&w, (size_t)N * K * 2)); CK(hipMalloc(&c, (size_t)M * N * 2));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&st, 8 * 32 * 8));
    CK(hipMemset(a, 0x11, (size_t)M * K * 2)); CK(hipMemset(w, 0x22, (size_t)N * K * 2)); CK(hipMemset(bias, 0, N * 4));
    setenv("TT_GEMM_ABLATE", "6", 1);
    for (int i = 0; i < 3; ++i) { CK(hipMemset(st, 0, 8 * 32 * 8)); tt_gemm_debug_stamps(a, w, bias, c, M, N, K, st, nullptr); CK(hipDeviceSynchronize()); }
    unsigned long long h[8 * 32];
    CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
    const char* names[] = {"La start", "glds issued", "reads issued", "reads back", "past barrier", "MFMA issued", "Lb start(past barrier)", "glds issued", "reads issued", "reads back", "past barrier", "MFMA issued", "tile end (past barrier)"};
    for (int wv : {0, 4}) {
        printf("wave %d (deltas in cycles from La start %llu):\n", wv, h[wv * 32]);
        for (int i = 0; i < 13; ++i) printf("  %2d %-26s t=%6lld  d=%5lld\n", i, names[i], (long long)(h[wv * 32 + i] - h[wv * 32]), i ? (long long)(h[wv * 32 + i] - h[wv * 32 + i - 1]) : 0LL);
    }
    printf("coarse (wave 0, block 0): prologue %lld  main loop %lld  epilogue+store drain %lld cycles\n", (long long)(h[21]-h[20]), (long long)(h[22]-h[21]), (long long)(h[23]-h[22]));
    printf("wave4 La start - wave0 La start = %lld\n", (long long)(h[4 * 32] - h[0]));
    return 0;
}
