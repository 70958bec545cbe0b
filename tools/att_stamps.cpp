// Diagnostic: s_memtime stamps of one attention workgroup (4 waves) in the middle of the grid: where does a workgroup's
// lifetime go?  800 sequences x 292 tokens x 16 heads x 64 (the rerank shape), random data.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
extern "C" int tt_attention_debug_stamps(const void*, int, int, int, const void*, int, void*, int, const int32_t*, const int32_t*, int, int, int, int, void*, void*);
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill(uint16_t* p, size_t n, uint64_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 2654435761ULL + seed; x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
        float u = ((x & 0xFFFF) + ((x >> 16) & 0xFFFF)) * (1.0f / 65536.0f) - 1.0f;
        uint32_t b = __float_as_uint(u); b += 0x7FFF + ((b >> 16) & 1); p[i] = (uint16_t)(b >> 16);
    }
}
int main(int argc, char** argv) {
    int n_seq = argc > 1 ? atoi(argv[1]) : 1600, len = argc > 2 ? atoi(argv[2]) : 292;
    const int heads = 16, dh = 64, H = heads * dh, stride = (len + 7) / 8 * 8;
    size_t T = ((size_t)n_seq * stride + 255) / 256 * 256;
    uint16_t *qk, *vt, *out; int32_t *ss, *sl; unsigned long long* st;
    CK(hipMalloc(&qk, T * 2 * H * 2)); CK(hipMalloc(&vt, T * H * 2)); CK(hipMalloc(&out, T * H * 2));
    CK(hipMalloc(&ss, n_seq * 4)); CK(hipMalloc(&sl, n_seq * 4)); CK(hipMalloc(&st, 256 * 8));
    std::vector<int32_t> hs(n_seq), hl(n_seq);
    for (int i = 0; i < n_seq; ++i) { hs[i] = i * stride; hl[i] = len; }
    CK(hipMemcpy(ss, hs.data(), n_seq * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sl, hl.data(), n_seq * 4, hipMemcpyHostToDevice));
    fill<<<2048, 256>>>(qk, T * 2 * H, 1); fill<<<2048, 256>>>(vt, T * H, 2);
    CK(hipDeviceSynchronize());
    for (int it = 0; it < 3; ++it) {
        CK(hipMemset(st, 0, 256 * 8));
        int rc = tt_attention_debug_stamps(qk, 2 * H, 0, H, vt, 8 * H, out, H, ss, sl, n_seq, heads, dh, len, st, nullptr);
        if (rc) { fprintf(stderr, "rc=%d\n", rc); return 1; }
        CK(hipDeviceSynchronize());
    }
    unsigned long long h[256];
    CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
    const int n_kt = (len + 63) / 64;
    for (int w = 0; w < 4; ++w) {
        const unsigned long long* b = h + w * 64;
        if (!b[0]) { printf("wave %d: no stamps\n", w); continue; }
        printf("wave %d: entry -> copies issued %lld | ", w, (long long)(b[1] - b[0]));
        for (int kt = 0; kt < n_kt; ++kt) {
            const unsigned long long* t = b + 2 + kt * 5;
            if (!t[3]) { printf("tile %d: wait %lld barrier %lld (idle wave) | ", kt, (long long)(t[1] - t[0]), (long long)(t[2] - t[1])); continue; }
            const unsigned long long nxt = (kt + 1 < n_kt) ? t[5] : b[2 + n_kt * 5];
            printf("tile %d: wait %lld barrier %lld issue+S %lld softmax %lld PV %lld | ", kt, (long long)(t[1] - t[0]), (long long)(t[2] - t[1]),
                   (long long)(t[3] - t[2]), (long long)(t[4] - t[3]), (long long)(nxt - t[4]));
        }
        printf("total to loop end %lld\n", (long long)(b[2 + n_kt * 5] - b[0]));
    }
    return 0;
}
