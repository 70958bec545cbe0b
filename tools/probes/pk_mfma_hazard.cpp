// Probe: which packed-f32 VALU forms (v_pk_mul/add/fma_f32 with their op_sel / neg modifiers) return wrong lanes when ANOTHER
// kernel's MFMA loop shares their SIMDs?  Written after embed_ln_kernel (rowops.hip) was found to return wrong lanes 48-63 of one
// register whenever the streaming scan (32x32x16 MFMA loop) ran on a second stream (profiles/r03_contention_race.log).
// A victim kernel repeats ONE instruction form on lane-dependent inputs and compares every result, bit for bit, with the scalar
// arithmetic the modifiers describe; an aggressor kernel on a second stream spins on MFMAs (or plain FMAs, the control).
// The forms listed are every packed-f32 form the library's disassembly holds, plus the "low result reads the HIGH dword" (op_sel)
// forms that turned out to be the fragile ones.
//   hipcc -O2 -fno-slp-vectorize --offload-arch=gfx950 -o pk_mfma_hazard pk_mfma_hazard.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { MUL = 0, ADD = 1, FMA = 2 };
// one form: operation, op_sel / op_sel_hi / neg bit sets (bit i = source i; neg applies to both halves), and its assembly text
#define FORMS(X)                                                                                                         \
    X(0, MUL, 0, 3, 0, "v_pk_mul_f32 %0, %1, %2")                                                                        \
    X(1, MUL, 0, 2, 0, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")                                                        \
    X(2, MUL, 0, 1, 0, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]")                                                        \
    X(3, ADD, 0, 3, 0, "v_pk_add_f32 %0, %1, %2")                                                                        \
    X(4, ADD, 0, 3, 2, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")                                              \
    X(5, ADD, 0, 1, 2, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")                              \
    X(6, ADD, 0, 2, 0, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]")                                                        \
    X(7, FMA, 0, 7, 0, "v_pk_fma_f32 %0, %1, %2, %3")                                                                    \
    X(8, FMA, 0, 3, 0, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]")                                                  \
    X(9, FMA, 0, 6, 0, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]")                                                  \
    X(10, FMA, 0, 7, 1, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]")                                     \
    X(11, FMA, 0, 1, 1, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]")                  \
    X(12, FMA, 0, 3, 4, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]")                  \
    /* low result reads a HIGH dword: */                                                                                 \
    X(13, FMA, 2, 5, 0, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]")                                  \
    X(14, FMA, 2, 7, 0, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]")                                                    \
    X(15, FMA, 1, 6, 0, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]")                                  \
    X(16, FMA, 4, 3, 0, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]")                                  \
    X(17, MUL, 2, 1, 0, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]")                                          \
    X(18, ADD, 2, 1, 0, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]")                                          \
    X(19, MUL, 1, 3, 0, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]")
constexpr int kForms = 20;

template <int OP, int SEL, int SELHI, int NEG>
__device__ inline v2f expected(v2f s0, v2f s1, v2f s2) {
    v2f e;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int sel = h ? SELHI : SEL;
        float a = s0[(sel >> 0) & 1], b = s1[(sel >> 1) & 1], c = s2[(sel >> 2) & 1];
        if (NEG & 1) a = -a;
        if (NEG & 2) b = -b;
        if (NEG & 4) c = -c;
        e[h] = OP == MUL ? a * b : OP == ADD ? a + b : __builtin_fmaf(a, b, c);
    }
    return e;
}

// counters: [0] wrong results, [1..4] by quarter-wave (lane >> 4), [5] low half wrong, [6] high half wrong, [7] results checked / 1000
template <int V>
__global__ void victim(unsigned long long* cnt, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float a = seed + 0.001f * (float)(threadIdx.x + 1), b = a * 1.37f + 0.5f;
    const float s = 0.75f + 0.01f * (float)(blockIdx.x & 31), m = 0.125f * (float)(1 + (lane & 7));
    unsigned bad = 0, bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < iters; ++it) {
        const v2f s0 = {a, b}, s1 = {s, s + 1.5f}, s2 = {m, m + 2.25f};
        v2f r, e;
#define X(ID, OP, SEL, SELHI, NEG, TXT)                                                                   \
        if constexpr (V == ID) {                                                                          \
            asm volatile(TXT : "=v"(r) : "v"(s0), "v"(s1), "v"(s2));                                      \
            e = expected<OP, SEL, SELHI, NEG>(s0, s1, s2);                                                \
        }
        FORMS(X)
#undef X
        const bool lo = __float_as_uint(r[0]) != __float_as_uint(e[0]), hi = __float_as_uint(r[1]) != __float_as_uint(e[1]);
        bad += lo || hi; bad_lo += lo; bad_hi += hi;
        a = a * 1.0001f + 0.003f; b = b * 0.9999f + 0.007f;
        if (a > 64.f) a -= 63.f;
    }
    if (bad) {
        atomicAdd(&cnt[0], (unsigned long long)bad); atomicAdd(&cnt[1 + (lane >> 4)], (unsigned long long)bad);
        atomicAdd(&cnt[5], (unsigned long long)bad_lo); atomicAdd(&cnt[6], (unsigned long long)bad_hi);
    }
    if (threadIdx.x == 0) atomicAdd(&cnt[7], (unsigned long long)((long long)iters * blockDim.x * 2 / 1000));
}

// The library's only other op_sel user: v_fma_mix_f32 reading an fp16 from the LOW / HIGH half of a dword as src1 (attention_cls_kernel, fp16 build)
template <int HI>
__global__ void victim_mix(unsigned long long* cnt, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float a = seed + 0.001f * (float)(threadIdx.x + 1);
    const float c = 0.125f * (float)(1 + (lane & 7));
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const _Float16 h0 = (_Float16)(0.5f + 0.01f * (float)((it + lane) & 31)), h1 = (_Float16)(1.5f + 0.02f * (float)((it + lane) & 15));
        const unsigned hw = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
        float r;
        if constexpr (HI) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(hw), "v"(c));
        else asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(hw), "v"(c));
        const float e = __builtin_fmaf(a, (float)(HI ? h1 : h0), c);
        bad += __float_as_uint(r) != __float_as_uint(e);
        a = a * 1.0001f + 0.003f;
        if (a > 64.f) a -= 63.f;
    }
    if (bad) { atomicAdd(&cnt[0], (unsigned long long)bad); atomicAdd(&cnt[1 + (lane >> 4)], (unsigned long long)bad); atomicAdd(&cnt[5], (unsigned long long)bad); }
    if (threadIdx.x == 0) atomicAdd(&cnt[7], (unsigned long long)((long long)iters * blockDim.x / 1000));
}

// aggressors: K = 0 plain FMA loop (control), 1 = 16x16x32 bf16 MFMA (the GEMMs'), 2 = 32x32x16 bf16 MFMA (the scan's, attention's)
template <int K>
__global__ void aggressor(float* sink, int iters) {
    const float f = 1.0f + 1e-6f * (float)threadIdx.x;
    if constexpr (K == 0) {
        float x0 = f, x1 = f + 1, x2 = f + 2, x3 = f + 3;
        for (int it = 0; it < iters * 16; ++it) {
            x0 = __builtin_fmaf(x0, 0.9999f, 0.001f); x1 = __builtin_fmaf(x1, 0.9999f, 0.001f);
            x2 = __builtin_fmaf(x2, 0.9999f, 0.001f); x3 = __builtin_fmaf(x3, 0.9999f, 0.001f);
        }
        if (x0 + x1 + x2 + x3 == 12345.f) sink[threadIdx.x] = x0;
    } else if constexpr (K == 1) {
        v8bf a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(f * 0.01f); b[i] = (__bf16)(0.02f); }
        v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[threadIdx.x] = c0[0];
    } else {
        v8bf a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(f * 0.01f); b[i] = (__bf16)(0.02f); }
        v16f c0, c1;
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        }
        if (c0[0] + c1[1] == 12345.f) sink[threadIdx.x] = c0[0];
    }
}

template <int V>   // V < 100: packed form V of FORMS; 100 / 101: v_fma_mix_f32 reading the low / high fp16
static void run_victim(const char* name, hipStream_t vs, hipStream_t as, int agg, int agg_iters, unsigned long long* dcnt, float* sink) {
    unsigned long long h[8];
    CK(hipMemsetAsync(dcnt, 0, sizeof h, vs));
    CK(hipStreamSynchronize(vs));
    if (agg == 0) aggressor<0><<<512, 256, 0, as>>>(sink, agg_iters / 8);
    else if (agg == 1) aggressor<1><<<512, 256, 0, as>>>(sink, agg_iters);
    else if (agg == 2) aggressor<2><<<512, 256, 0, as>>>(sink, agg_iters / 2);
    int launches = 0;
    do {                                    // victim launches for as long as the aggressor is resident (at least 4, at most 64)
        if constexpr (V >= 100) victim_mix<V - 100><<<2048, 256, 0, vs>>>(dcnt, 256, 1.0f + 0.01f * (float)launches);
        else victim<V><<<2048, 256, 0, vs>>>(dcnt, 256, 1.0f + 0.01f * (float)launches);
        CK(hipStreamSynchronize(vs));
        ++launches;
    } while (launches < 4 || (agg >= 0 && hipStreamQuery(as) == hipErrorNotReady && launches < 64));
    CK(hipStreamSynchronize(as));
    CK(hipMemcpy(h, dcnt, sizeof h, hipMemcpyDeviceToHost));
    printf("    %-82s %3d launches %8.1f M results  wrong %10llu  (quarter-waves %llu %llu %llu %llu; low %llu high %llu)\n", name,
           launches, (double)h[7] / 1e3, h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
}

int main(int argc, char** argv) {
    const int agg_iters = argc > 1 ? atoi(argv[1]) : (1 << 20);
    hipStream_t vs, as;
    CK(hipStreamCreate(&vs)); CK(hipStreamCreate(&as));
    unsigned long long* dcnt; float* sink;
    CK(hipMalloc(&dcnt, 8 * sizeof(unsigned long long))); CK(hipMalloc(&sink, 4096));
    const char* an[4] = {"nothing else on the GPU", "a plain v_fma_f32 loop on a second stream", "a 16x16x32 bf16 MFMA loop on a second stream",
                         "a 32x32x16 bf16 MFMA loop on a second stream"};
    for (int agg = -1; agg <= 2; ++agg) {
        printf("== neighbour: %s\n", an[agg + 1]);
#define X(ID, OP, SEL, SELHI, NEG, TXT) run_victim<ID>(TXT, vs, as, agg, agg_iters, dcnt, sink);
        FORMS(X)
#undef X
        run_victim<100>("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]", vs, as, agg, agg_iters, dcnt, sink);
        run_victim<101>("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]", vs, as, agg, agg_iters, dcnt, sink);
        fflush(stdout);
    }
    (void)kForms;
    return 0;
}
