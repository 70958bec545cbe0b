// Diagnostic: per-phase s_memtime stamps of the v4 GEMM loop (TT_GEMM_ABLATE=6 build path).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
extern "C" int tt_gemm_debug_stamps(const void*, const void*, const float*, void*, int, int, int, void*, void*);
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
    // optional: M N K (a grid of a few workgroups shows the epilogue of a CU that has the memory system to itself)
    int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 1024;
    printf("M=%d N=%d K=%d: %d workgroups\n", M, N, K, (M / 256) * (N / 256));
    const int n_blocks_max = (M / 256) * (N / 256) + 64;
    uint16_t *a, *w, *c; float* bias; unsigned long long* st;
    CK(hipMalloc(&a, (size_t)M * K * 2)); CK(hipMalloc(&w, (size_t)N * K * 2)); CK(hipMalloc(&c, (size_t)M * N * 2));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&st, (256 + (size_t)n_blocks_max * 4) * 8));
    CK(hipMemset(a, 0x11, (size_t)M * K * 2)); CK(hipMemset(w, 0x22, (size_t)N * K * 2)); CK(hipMemset(bias, 0, N * 4));
    setenv("TT_GEMM_ABLATE", "6", 1);
    for (int i = 0; i < 3; ++i) { CK(hipMemset(st, 0, (256 + (size_t)n_blocks_max * 4) * 8)); tt_gemm_debug_stamps(a, w, bias, c, M, N, K, st, nullptr); CK(hipDeviceSynchronize()); }
    unsigned long long h[8 * 32];
    CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
    const char* names[] = {"La start", "glds issued", "reads issued", "reads back", "past barrier", "MFMA issued", "Lb start(past barrier)", "glds issued", "reads issued", "reads back", "past barrier", "MFMA issued", "tile end (past barrier)"};
    for (int wv : {0, 4}) {
        printf("wave %d (deltas in cycles from La start %llu):\n", wv, h[wv * 32]);
        for (int i = 0; i < 13; ++i) printf("  %2d %-26s t=%6lld  d=%5lld\n", i, names[i], (long long)(h[wv * 32 + i] - h[wv * 32]), i ? (long long)(h[wv * 32 + i] - h[wv * 32 + i - 1]) : 0LL);
    }
    printf("coarse (wave 0, block 0): prologue %lld  main loop %lld  epilogue+store drain %lld cycles\n", (long long)(h[21]-h[20]), (long long)(h[22]-h[21]), (long long)(h[23]-h[22]));
    printf("wave4 La start - wave0 La start = %lld\n", (long long)(h[4 * 32] - h[0]));
    // ---- per-CU timeline: gap between a workgroup's exit (stores drained) and the next workgroup's entry on the same CU
    {
        const size_t nb = (size_t)n_blocks_max;
        unsigned long long* r = (unsigned long long*)malloc(nb * 4 * 8);
        CK(hipMemcpy(r, st + 256, nb * 4 * 8, hipMemcpyDeviceToHost));
        struct Rec { unsigned long long cu, s, e; };
        Rec* v = (Rec*)malloc(nb * sizeof(Rec));
        size_t n = 0;
        for (size_t b = 0; b < nb; ++b)
            if (r[b * 4] && r[b * 4 + 1]) v[n++] = Rec{((r[b * 4 + 2] >> 32) << 8) | ((r[b * 4 + 2] >> 8) & 0xFF), r[b * 4], r[b * 4 + 1]};   // (XCC, SE/SH/CU): drop wave / SIMD / pipe ids
        qsort(v, n, sizeof(Rec), [](const void* a, const void* b) {
            const Rec* x = (const Rec*)a; const Rec* y = (const Rec*)b;
            if (x->cu != y->cu) return x->cu < y->cu ? -1 : 1;
            return x->s < y->s ? -1 : (x->s > y->s ? 1 : 0);
        });
        double gap = 0, life = 0; size_t ng = 0, cus = 0; long long gmin = 1LL << 60, gmax = 0;
        for (size_t i = 0; i < n; ++i) {
            life += (double)(v[i].e - v[i].s);
            if (i == 0 || v[i].cu != v[i - 1].cu) { ++cus; continue; }
            const long long g = (long long)(v[i].s - v[i - 1].e);
            gap += (double)g; ++ng; if (g < gmin) gmin = g; if (g > gmax) gmax = g;
        }
        printf("%zu workgroups on %zu distinct CUs: mean in-kernel time %.0f cycles, exit -> next entry on the same CU: "
               "mean %.0f cycles (min %lld, max %lld, %zu gaps)\n", n, cus, life / n, ng ? gap / ng : 0.0, gmin, gmax, ng);
    }
    return 0;
}
