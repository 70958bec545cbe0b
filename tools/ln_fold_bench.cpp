// Round 6: LayerNorm folding MEASURED on one encoder layer's GEMMs, diagnostic library only.
//   today : qkv(bias) + o-proj(+residual) + LayerNorm + ffn-up(GELU) + ffn-down(+residual) + LayerNorm
//   folded: the four GEMMs with the folded epilogues (GemmParams.lnf: consumers scale by rstd and add the rank-1 mean correction, producers
//           rebuild LayerNorm(residual) on the fly and emit per-row partial statistics) + a statistics-finalising launch per LayerNorm
// Also checks the epilogues' arithmetic: identity statistics must reproduce the plain kernels bit for bit, real statistics must match
// the host's LayerNorm algebra on sampled elements, and the emitted partials must be the rows' sums.
//   ./ln_fold_bench [M] [iters]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/tt_hip.h"

extern "C" int tt_gemm_debug_lnfold(const void* a, const void* w, const float* bias, const void* residual, void* c, int m, int n, int k,
                                    int epilogue, int lnf, const float* rows, const float* c0, const float* c1, float* part, void* stream);

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint32_t hash32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}
__global__ void fill_bf16(uint16_t* p, size_t n, uint64_t seed, float scale, float offset) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = hash32(i * 2654435761ULL + seed);
        float u = ((h & 0xFFFF) + (h >> 16)) * (1.0f / 65536.0f) - 1.0f;
        uint32_t b = __float_as_uint(u * scale + offset);
        b += 0x7FFF + ((b >> 16) & 1);
        p[i] = (uint16_t)(b >> 16);
    }
}
__global__ void fill_f32(float* p, size_t n, uint64_t seed, float scale, float offset) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = hash32(i * 2654435761ULL + seed);
        p[i] = (((h & 0xFFFF) + (h >> 16)) * (1.0f / 65536.0f) - 1.0f) * scale + offset;
    }
}
// (sum, sum of squares) partials [M][16][2] -> (rstd, -mu rstd) [M][2]: what a consumer's prologue strip holds.  One float2 per lane (a
// wave reads 512 contiguous bytes = 4 rows), the 16 partials of a row combined across 16 lanes in a fixed order
__global__ void finalize_stats(const float* part, int M, int P, int H, float eps, float* rows) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t m = t >> 4;
    float2 v = make_float2(0.f, 0.f);
    if (m < (size_t)M) v = *reinterpret_cast<const float2*>(part + t * 2);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); }
    if (m < (size_t)M && (t & 15) == 0) {
        const float mu = v.x / H, var = fmaxf(v.y / H - mu * mu, 0.f), rstd = 1.0f / sqrtf(var + eps);
        *reinterpret_cast<float2*>(rows + m * 2) = make_float2(rstd, -mu * rstd);
    }
}
static float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 473600;
    const int iters = argc > 2 ? atoi(argv[2]) : 10;
    const int H = 1024, F = 4096;
    uint16_t *x, *y, *y2, *qkv, *ffn, *w_qkv, *w_o, *w_up, *w_dn, *ctx;
    float *bias, *gamma, *beta, *cs, *rows, *rows_id, *part, *ones, *zeros;
    CK(hipMalloc(&x, (size_t)M * H * 2)); CK(hipMalloc(&y, (size_t)M * H * 2)); CK(hipMalloc(&y2, (size_t)M * H * 2));
    CK(hipMalloc(&ctx, (size_t)M * H * 2)); CK(hipMalloc(&qkv, (size_t)M * 3 * H * 2)); CK(hipMalloc(&ffn, (size_t)M * F * 2));
    CK(hipMalloc(&w_qkv, (size_t)3 * H * H * 2)); CK(hipMalloc(&w_o, (size_t)H * H * 2)); CK(hipMalloc(&w_up, (size_t)F * H * 2)); CK(hipMalloc(&w_dn, (size_t)H * F * 2));
    CK(hipMalloc(&bias, F * 4)); CK(hipMalloc(&gamma, F * 4)); CK(hipMalloc(&beta, F * 4)); CK(hipMalloc(&cs, F * 4));
    CK(hipMalloc(&ones, F * 4)); CK(hipMalloc(&zeros, F * 4));
    CK(hipMalloc(&rows, (size_t)M * 8)); CK(hipMalloc(&rows_id, (size_t)M * 8)); CK(hipMalloc(&part, (size_t)M * (H / 64) * 8));
    fill_bf16<<<2048, 256>>>(x, (size_t)M * H, 1, 1.0f, 0.1f);
    fill_bf16<<<2048, 256>>>(ctx, (size_t)M * H, 2, 1.0f, 0.0f);
    fill_bf16<<<2048, 256>>>(w_qkv, (size_t)3 * H * H, 3, 0.05f, 0.f);
    fill_bf16<<<2048, 256>>>(w_o, (size_t)H * H, 4, 0.05f, 0.f);
    fill_bf16<<<2048, 256>>>(w_up, (size_t)F * H, 5, 0.05f, 0.f);
    fill_bf16<<<2048, 256>>>(w_dn, (size_t)H * F, 6, 0.02f, 0.f);
    fill_f32<<<64, 256>>>(bias, F, 7, 0.1f, 0.f);
    fill_f32<<<64, 256>>>(gamma, F, 8, 0.3f, 1.0f);
    fill_f32<<<64, 256>>>(beta, F, 9, 0.2f, 0.f);
    fill_f32<<<64, 256>>>(cs, F, 10, 0.5f, 0.f);
    fill_f32<<<64, 256>>>(ones, F, 0, 0.f, 1.0f);
    fill_f32<<<64, 256>>>(zeros, F, 0, 0.f, 0.0f);
    fill_f32<<<2048, 256>>>(rows, (size_t)M * 2, 11, 0.2f, 0.7f);
    {   // identity statistics: rstd = 1, -mu rstd = 0
        std::vector<float> h((size_t)M * 2);
        for (int m = 0; m < M; ++m) { h[2 * m] = 1.f; h[2 * m + 1] = 0.f; }
        CK(hipMemcpy(rows_id, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));

    // ---- arithmetic checks (M rows sampled) -----------------------------------------------------------------------------------
    {
        const int Mc = M < 16384 ? M : 16384;      // (>= 128 tiles of 256 x 256: below that the launcher picks the 128 x 128 kernel, which has no folded epilogue)
        std::vector<uint16_t> a((size_t)Mc * H), b((size_t)Mc * H), r((size_t)Mc * H);
        // consumer, identity statistics == plain bias GEMM (N = 1024 slice of the qkv weight)
        if (tt_gemm_bf16(x, w_qkv, bias, nullptr, y, Mc, H, H, 0, st)) { fprintf(stderr, "%s\n", tt_last_error()); return 1; }
        if (tt_gemm_debug_lnfold(x, w_qkv, bias, nullptr, y2, Mc, H, H, 0, 1, rows_id, cs, nullptr, nullptr, st)) { fprintf(stderr, "%s\n", tt_last_error()); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(a.data(), y, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), y2, b.size() * 2, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
        printf("check consumer (identity statistics) vs plain bias GEMM: %zu of %zu elements differ\n", diff, a.size());
        // consumer, real statistics: out = rstd * (plain - bias) + nmr * cs + bias, against the plain output (bf16 rounding apart)
        if (tt_gemm_debug_lnfold(x, w_qkv, zeros, nullptr, y, Mc, H, H, 0, 1, rows_id, zeros, nullptr, nullptr, st)) return 1;      // raw acc (bf16)
        if (tt_gemm_debug_lnfold(x, w_qkv, bias, nullptr, y2, Mc, H, H, 0, 1, rows, cs, nullptr, nullptr, st)) return 1;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(a.data(), y, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), y2, b.size() * 2, hipMemcpyDeviceToHost));
        std::vector<float> hr((size_t)Mc * 2), hcs(H), hb(H), hg(H), hbt(H);
        CK(hipMemcpy(hr.data(), rows, hr.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hcs.data(), cs, H * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), bias, H * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hg.data(), gamma, H * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hbt.data(), beta, H * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int m = 0; m < Mc; m += 37)
            for (int n = 0; n < H; n += 13) {
                const float acc = bf(a[(size_t)m * H + n]);
                const float want = hr[2 * m] * acc + hr[2 * m + 1] * hcs[n] + hb[n];
                worst = fmax(worst, fabs(bf(b[(size_t)m * H + n]) - want) / (fabs(want) + 1.0));
            }
        printf("check consumer (real statistics) vs host algebra on the bf16-rounded accumulators: worst relative deviation %.2e (bf16: 4e-3)\n", worst);
        // producer, identity LayerNorm (gamma 1, beta 0, rstd 1, mu 0) == plain residual GEMM; its partials == the output rows' sums
        if (tt_gemm_bf16(ctx, w_o, bias, x, y, Mc, H, H, 2, st)) return 1;
        if (tt_gemm_debug_lnfold(ctx, w_o, bias, x, y2, Mc, H, H, 2, 2, rows_id, ones, zeros, part, st)) { fprintf(stderr, "%s\n", tt_last_error()); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(a.data(), y, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), y2, b.size() * 2, hipMemcpyDeviceToHost));
        diff = 0;
        for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
        printf("check producer (identity LayerNorm) vs plain residual GEMM: %zu of %zu elements differ\n", diff, a.size());
        std::vector<float> hp((size_t)Mc * (H / 64) * 2);
        CK(hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost));
        double ws = 0, wq = 0;
        for (int m = 0; m < Mc; m += 29) {
            double s = 0, q = 0, ps = 0, pq = 0;
            for (int n = 0; n < H; ++n) { const double v = bf(b[(size_t)m * H + n]); s += v; q += v * v; }
            for (int i = 0; i < H / 64; ++i) { ps += hp[((size_t)m * (H / 64) + i) * 2]; pq += hp[((size_t)m * (H / 64) + i) * 2 + 1]; }
            ws = fmax(ws, fabs(ps - s) / (fabs(s) + 1.0)); wq = fmax(wq, fabs(pq - q) / (fabs(q) + 1.0));
        }
        printf("check producer partials vs the output rows' own sums: worst relative deviation sum %.2e, sum of squares %.2e\n", ws, wq);
        // producer, real LayerNorm of the residual: out = plain(no residual) + LN(x)
        if (tt_gemm_debug_lnfold(ctx, w_o, bias, x, y2, Mc, H, H, 2, 2, rows, gamma, beta, part, st)) return 1;
        if (tt_gemm_bf16(ctx, w_o, bias, nullptr, y, Mc, H, H, 0, st)) return 1;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(a.data(), y, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), y2, b.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r.data(), x, r.size() * 2, hipMemcpyDeviceToHost));
        worst = 0;
        for (int m = 0; m < Mc; m += 37)
            for (int n = 0; n < H; n += 13) {
                const float ln = bf(r[(size_t)m * H + n]) * (hr[2 * m] * hg[n]) + (hr[2 * m + 1] * hg[n] + hbt[n]);
                const float want = bf(a[(size_t)m * H + n]) + ln;
                worst = fmax(worst, fabs(bf(b[(size_t)m * H + n]) - want) / (fabs(want) + 1.0));
            }
        printf("check producer (real LayerNorm of the residual) vs host algebra: worst relative deviation %.2e (two bf16 roundings: 8e-3)\n", worst);
    }

    // ---- timing --------------------------------------------------------------------------------------------------------------------
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto&& fn) {
        for (int i = 0; i < 2; ++i) fn();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) fn();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-58s %8.3f ms\n", name, ms / iters);
        return ms / iters;
    };
    const int P = H / 64;
    printf("ln_fold_bench: M=%d, H=%d, F=%d, iters=%d (diagnostic library)\n", M, H, F, iters);
    float t_today = 0, t_fold = 0, t_ln = 0, t;
    // two passes, today / folded / today / folded: the first pass also warms clocks and buffers (the first timed launch after the host-side
    // checks above runs ~7 % slow); the second pass is the one reported
    for (int pass = 0; pass < 2; ++pass) {
        t_today = t_fold = t_ln = 0;
        printf("pass %d -- today\n", pass + 1);
        t_today += timeit("qkv      bias      N 3072 K 1024", [&] { tt_gemm_bf16(x, w_qkv, bias, nullptr, qkv, M, 3 * H, H, 0, st); });
        t_today += timeit("o-proj   +residual N 1024 K 1024", [&] { tt_gemm_bf16(ctx, w_o, bias, x, y, M, H, H, 2, st); });
        t = timeit("LayerNorm (rows of 1024)", [&] { tt_layernorm_bf16(y, y2, gamma, beta, M, H, 1e-5f, st); });
        t_today += t; t_ln += t;
        t_today += timeit("ffn-up   GELU      N 4096 K 1024", [&] { tt_gemm_bf16(x, w_up, bias, nullptr, ffn, M, F, H, 1, st); });
        t_today += timeit("ffn-down +residual N 1024 K 4096", [&] { tt_gemm_bf16(ffn, w_dn, bias, x, y, M, H, F, 2, st); });
        t = timeit("LayerNorm (rows of 1024)", [&] { tt_layernorm_bf16(y, y2, gamma, beta, M, H, 1e-5f, st); });
        t_today += t; t_ln += t;
        printf("  per layer %.3f ms\npass %d -- folded\n", t_today, pass + 1);
        t_fold += timeit("qkv      bias, consumer epilogue", [&] { tt_gemm_debug_lnfold(x, w_qkv, bias, nullptr, qkv, M, 3 * H, H, 0, 1, rows, cs, nullptr, nullptr, st); });
        t_fold += timeit("o-proj   +LayerNorm(residual), partials out", [&] { tt_gemm_debug_lnfold(ctx, w_o, bias, x, y2, M, H, H, 2, 2, rows, gamma, beta, part, st); });
        t_fold += timeit("statistics finalise (16 partials per row)", [&] { finalize_stats<<<(unsigned)(((size_t)M * 16 + 255) / 256), 256, 0, st>>>(part, M, P, H, 1e-5f, rows); });
        t_fold += timeit("ffn-up   GELU, consumer epilogue", [&] { tt_gemm_debug_lnfold(x, w_up, bias, nullptr, ffn, M, F, H, 1, 1, rows, cs, nullptr, nullptr, st); });
        t_fold += timeit("ffn-down +LayerNorm(residual), partials out", [&] { tt_gemm_debug_lnfold(ffn, w_dn, bias, x, y2, M, H, F, 2, 2, rows, gamma, beta, part, st); });
        t_fold += timeit("statistics finalise (16 partials per row)", [&] { finalize_stats<<<(unsigned)(((size_t)M * 16 + 255) / 256), 256, 0, st>>>(part, M, P, H, 1e-5f, rows); });
        printf("  per layer %.3f ms\n", t_fold);
    }
    // the consumer epilogue does MORE arithmetic than the plain one, yet the qkv launch above is faster with it: alternate the two
    // forms (and the folded form on identity statistics) to see whether that is the epilogue or the order / the data
    printf("alternating, same operands\n");
    for (int rep = 0; rep < 3; ++rep) {
        timeit("qkv plain", [&] { tt_gemm_bf16(x, w_qkv, bias, nullptr, qkv, M, 3 * H, H, 0, st); });
        timeit("qkv consumer epilogue, real statistics", [&] { tt_gemm_debug_lnfold(x, w_qkv, bias, nullptr, qkv, M, 3 * H, H, 0, 1, rows, cs, nullptr, nullptr, st); });
        timeit("qkv consumer epilogue, identity statistics (same values out)", [&] { tt_gemm_debug_lnfold(x, w_qkv, bias, nullptr, qkv, M, 3 * H, H, 0, 1, rows_id, zeros, nullptr, nullptr, st); });
    }
    printf("saved per layer: %.3f ms of the %.3f ms the two LayerNorm launches take today (x 24 layers = %.1f ms per step)\n",
           t_today - t_fold, t_ln, 24 * (t_today - t_fold));
    return 0;
}
